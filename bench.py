"""rays/s of SHERF's volumetric-rendering hot path (ImportanceRenderer.forward) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one 512x512 frame x 64 samples/ray of synthetic input already resident
in HBM (BASELINE.json config 2: single subject novel view).  With N > 1 every rank renders its own target view
of the same subject (BASELINE config 4: views sharded across GPUs, weak scaling) and the step ends with the RCCL
all_gather of the rendered [rays, 5] tiles (asynchronous, awaited one step later; the last one inside the timed region).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_VALID_SAMPLE = 429248       # 214,624 MAC per valid sample (fusion + transformer + decoder)
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def make_inputs(cfg_name, theta, dev):
    from synthdata import fixtures     # seeded synthetic inputs; nothing from oracle/ on the timed path
    c = dict(fixtures.CONFIGS[cfg_name])
    c['theta_tgt'] = theta
    fixtures.CONFIGS['_bench'] = c
    fx = fixtures.renderer_inputs('_bench')
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {k: ({kk: to(vv) for kk, vv in v.items()} if isinstance(v, dict) else to(v)) for k, v in fx['input_data'].items()}
    return fx, d, to


def _device(lrank):
    torch.cuda.set_device(lrank)
    return torch.device('cuda', lrank)


def make_workload(a, theta, dev):
    """The bench frame: seeded synthetic subject / tables / weights on `dev`, the renderer and decoder in the requested modes."""
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder, TriPlaneGenerator
    from sherf_amd.voxel import SparseConvTensor
    from synthdata import fixtures, synth
    smpl = synth.make_synth_smpl(0)
    fx, d, to = make_inputs(a.config, theta, dev)
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=smpl, mlp_precision=a.precision)
    dec = NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.'); fixtures.load_seeded_state(dec, 'decoder.')
    rend.to(dev).train(a.bn_mode == 'train'); dec.to(dev).train(a.bn_mode == 'train')
    # voxelisation glue (triplane.py:129-137) through the product path
    gen = TriPlaneGenerator.__new__(TriPlaneGenerator)
    torch.nn.Module.__init__(gen); gen.renderer = rend
    can = gen.canonical_obs_vertices(d)
    sp_input, _ = gen.prepare_sp_input(d['t_vertices'].float(), can)
    sp = SparseConvTensor(to(fx['vertex_feat']), sp_input['coord'], sp_input['out_sh'], 1)
    opts = dict(fx['options']); opts['mlp_precision'] = a.precision
    return dict(rend=rend, dec=dec, d=d, sp=sp, sp_input=sp_input, planes=to(fx['planes']), obs_feat=to(fx['obs_feat']),
                obs_img=d['obs_img_all'][:, 0], opts=opts)


def tune_child(a, lrank):
    """`bench.py --tune-child`: renders the bench frame once on this rank's GPU, times every launch shape of the MLP kernel on it
    (sherf_amd.tune: each verified bit for bit against the default shape on the device) and prints the report as one line."""
    dev = _device(lrank)
    from sherf_amd import tune
    w = make_workload(a, 0.4, dev)
    d = w['d']
    with torch.no_grad():
        for _ in range(min(2, a.tune_iters)):
            w['rend'](w['planes'], w['obs_img'], w['obs_feat'], w['sp'], None, w['sp_input'], w['dec'], d['ray_o_all'][:, 0],
                      d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, w['opts'])
    torch.cuda.synchronize()
    rend = w['rend']
    kw = dict(iters=2 * a.tune_iters, warmup=min(3, a.tune_iters - 1))
    rep = tune.tune_mlp(rend, w['dec'], **kw)
    rep['gather'] = tune.tune_gather(rend, w['dec'], **kw)
    exact = tune.tune_mlp(rend, w['dec'], exact_capacity=True, **kw)          # the same shapes without their grids of empty workgroups
    rep['shapes_exact_grid'], rep['best_exact_grid'] = exact['shapes'], exact['best']
    bl = {'branch': False, 'branchless': True, 'branchless128': '128'}[rep['gather']['best']]

    def render():
        with torch.no_grad():
            return rend(w['planes'], w['obs_img'], w['obs_feat'], w['sp'], None, w['sp_input'], w['dec'], d['ray_o_all'][:, 0],
                        d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, w['opts'])
    # whole frames: the defaults, the tuned kernels, and both again with launches sized by the frame's own sample count
    cands = [dict(mlp_shape='8x1', gather_branchless=False, exact_grids=False),
             dict(mlp_shape=rep['best'], gather_branchless=bl, exact_grids=False),
             dict(mlp_shape='8x1', gather_branchless=False, exact_grids=True),
             dict(mlp_shape=rep['best_exact_grid'], gather_branchless=bl, exact_grids=True)]
    rep['frame'] = tune.tune_frame(render, rend, cands, iters=a.tune_iters, warmup=min(2, a.tune_iters - 1))
    rep['choice'] = cands[rep['frame']['best']]
    print('TUNE_JSON ' + json.dumps(rep), flush=True)


def pick_mlp_shape(a, lrank, timeout=300):
    """-> (shape name, report | {'error': ...}).  The MLP kernel has several launch shapes with identical results (sherf_amd/tune.py);
    which is fastest is a property of the device, so it is measured -- in a CHILD process, so that a shape that misbehaves on this
    hardware (every one of them is verified against the default on the device before it is eligible) cannot take the benchmark
    down; any failure of the child means the default shape.  Every rank tunes its own GPU; the choice never changes the output."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE',
                                                            'GROUP_RANK', 'ROLE_RANK', 'TORCHELASTIC_RUN_ID')}
    env['LOCAL_RANK'] = str(lrank)
    cmd = [sys.executable, os.path.abspath(__file__), '--tune-child', '--config', a.config, '--precision', a.precision, '--bn-mode', a.bn_mode]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('TUNE_JSON ')]
        if r.returncode != 0 or not line:
            return '8x1', dict(error=f'tune child rc={r.returncode}: {r.stderr.strip()[-300:]}')
        rep = json.loads(line[-1][len('TUNE_JSON '):])
        return rep.get('choice', dict(mlp_shape=rep['best']))['mlp_shape'], rep
    except Exception as ex:                                   # timeout, unparsable output: keep the default
        return '8x1', dict(error=f'{type(ex).__name__}: {str(ex)[:300]}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'bf16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--torch-gpu-baseline', action='store_true', help=argparse.SUPPRESS)      # (now the default at N = 1)
    ap.add_argument('--no-torch-gpu-baseline', action='store_true',
                    help='skip timing the oracle (the reference algorithm as stock ATen ops, brute-force K-NN) ON THE GPU over the whole '
                         'frame: the "reference single-GPU render()" denominator SURVEY.md section 8(d) asks for beside the CPU one '
                         '(N = 1 only; runs in a child process after the measurement, bounded to 4 minutes)')
    ap.add_argument('--torch-gpu-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--save-oracle', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--bn-mode', default='train', choices=['train', 'eval'],
                    help='BatchNorm of the voxel encoder. train (default) = batch statistics: the mode the reference renders in, '
                         'also at test time (eval_*.sh -> train.py --test_flag -> test(G, ...) with G built .train(), '
                         'training_loop.py:193,311-330); eval = running statistics (G_ema.eval(), training_loop.py:196)')
    ap.add_argument('--mlp-shape', default=os.environ.get('SHERF_MLP_SHAPE', 'auto'),
                    help="launch shape of sherf_nerf_mlp (sherf_amd.renderer.MLP_SHAPES); 'auto' (default) = time every shape on this "
                         'GPU in a child process before the run and use the fastest one whose output is bit-identical to the default')
    ap.add_argument('--tune-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--tune-iters', type=int, default=10, help=argparse.SUPPRESS)       # frames per candidate in the tune child
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); lrank = int(os.environ.get('LOCAL_RANK', 0))
    if a.tune_child:
        return tune_child(a, lrank)
    if a.torch_gpu_child:
        print('TORCH_GPU_JSON ' + json.dumps(torch_gpu_baseline(a.config, _device(lrank), a.bn_mode == 'train', save=a.save_oracle)), flush=True)
        return
    tune_report = None
    if a.mlp_shape == 'auto':              # before this process touches the GPU: see pick_mlp_shape
        a.mlp_shape, tune_report = pick_mlp_shape(a, lrank)
    dev = _device(lrank)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('SHERF_DIST_BACKEND', 'nccl')          # 'nccl' = RCCL; 'gloo' only for the CPU dry-run in tests/
        dist.init_process_group(backend, **(dict(device_id=dev) if backend == 'nccl' else {}))
    from sherf_amd import dist as sdist  # noqa: F401

    if os.environ.get('SHERF_DEBUG'):
        from sherf_amd import _lib
        _lib.lib().sherf_set_debug(int(os.environ['SHERF_DEBUG']))     # ablation runs only (tools/gpu_ablate.sh, tools/gpu_sweep.sh)
    w = make_workload(a, 0.4 + rank * 2 * np.pi / max(world, 1), dev)
    rend, dec, d, sp, sp_input, planes, obs_feat, obs_img, opts = (w[k] for k in ('rend', 'dec', 'd', 'sp', 'sp_input', 'planes', 'obs_feat',
                                                                                  'obs_img', 'opts'))
    rend.mlp_shape = a.mlp_shape
    choice = (tune_report or {}).get('choice', {})
    rend.gather_branchless = choice.get('gather_branchless') or rend.gather_branchless
    rend.exact_grids = bool(choice.get('exact_grids')) or rend.exact_grids
    ro, rd, nr, fr = d['ray_o_all'][:, 0], d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0]
    R = ro.shape[1]; S = opts['depth_resolution']
    from sherf_amd import _lib as _abi
    import ctypes as _ct

    if tune_report is not None and 'choice' in tune_report:
        # second guard (the child already verified every candidate on its GPU): this rank's own frame under the tuned switches must
        # equal the frame under the defaults bit for bit, or the defaults are kept
        tuned = dict(mlp_shape=rend.mlp_shape, gather_branchless=rend.gather_branchless, exact_grids=rend.exact_grids)
        if tuned != dict(mlp_shape='8x1', gather_branchless=False, exact_grids=False):
            def frame():
                with torch.no_grad():
                    return [t.clone() for t in rend(planes, obs_img, obs_feat, sp, None, sp_input, dec, ro, rd, nr, fr, d, opts)]
            got = frame()
            rend.mlp_shape, rend.gather_branchless, rend.exact_grids = '8x1', False, False
            ref = frame()
            if all(torch.equal(x, y) for x, y in zip(got, ref)):
                rend.mlp_shape, rend.gather_branchless, rend.exact_grids = tuned['mlp_shape'], tuned['gather_branchless'], tuned['exact_grids']
            else:
                tune_report['reverted_to_defaults'] = True
                a.mlp_shape = '8x1'

    def step():
        with torch.no_grad():
            rgb, depth, acc = rend(planes, obs_img, obs_feat, sp, None, sp_input, dec, ro, rd, nr, fr, d, opts)
            tile = torch.cat([rgb[0], depth[0], acc[0]], 1)
            if world > 1:
                # the gather of this frame's tiles runs on RCCL's own stream (async_op) and is awaited one step later, so that it
                # overlaps the next frame's sampling instead of stalling the render stream for a latency-bound 5 MB exchange
                out = [torch.empty_like(tile) for _ in range(world)]
                inflight.append((torch.distributed.all_gather(out, tile, async_op=True), out, tile))
                while len(inflight) > 1:
                    inflight.pop(0)[0].wait()
        return tile

    inflight = []

    def drain():                      # every gather issued so far is complete (on the render stream's timeline) after this
        while inflight:
            inflight.pop(0)[0].wait()

    for _ in range(a.warmup):
        step()
    drain()
    _abi.call('sherf_profile_frames', 1)       # HIP events around sherf_nerf_mlp etc. on their launch streams (csrc/frame.hip)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                        # the last frame's gather belongs to the timed region
    host_dt = time.perf_counter() - t0             # enqueue time of the K steps (the host runs ahead of the GPU)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    nv = int(rend.last['ws']['counters'][0])
    ms = (_ct.c_float * (64 * 8))(); n_ms = _ct.c_int32(0)
    _abi.call('sherf_profile_frames_read', ms, 64, _ct.byref(n_ms))
    _abi.call('sherf_profile_frames', 0)
    prof = np.array(ms[:n_ms.value * 8], dtype=np.float64).reshape(-1, 8)
    mlp_ms = float(prof[:, 7].mean()) if len(prof) else None
    if rank == 0:
        res = dict(metric='rendered rays/sec at 512x512x64 samples (ImportanceRenderer.forward)', value=world * R * a.steps / dt,
                   unit='rays/s', n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * dt / a.steps,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='bf16x3 MFMA (fp32-grade split bf16), fp32 elsewhere'
                   if a.precision == 'bf16x3' else 'bf16 MFMA, fp32 elsewhere', data='synthetic',
                   config=dict(workload=f'{a.config}: 512x512 rays x 64 samples, synthetic SMPL subject, novel view, all feature branches, '
                                        f'one view per GPU', rays=R, samples_per_ray=S, valid_samples=nv, valid_fraction=nv / (R * S),
                               parallelism=f'views x{world}' if world > 1 else 'single GPU', mlp_precision=a.precision,
                               batchnorm=a.bn_mode, mlp_shape=rend.mlp_shape,
                               gather={False: 'branch', True: 'branchless', '128': 'branchless128'}[rend.gather_branchless],
                               exact_grids=bool(rend.exact_grids)))
        if tune_report is not None:
            res['mlp_tune'] = tune_report
        if mlp_ms:
            ach = nv * FLOP_PER_VALID_SAMPLE / (mlp_ms * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_mlp_bytes_per_launch.json')
            if os.path.exists(pmc):
                traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
            res['roofline'] = dict(kernel='nerf_mlp_kernel', bound='mfma', achieved=ach, peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                                   frac=ach / PEAK_BF16_TFLOPS, traffic=traffic, kernel_ms=mlp_ms,
                                   algorithmic_flop_per_launch=nv * FLOP_PER_VALID_SAMPLE)
        if len(prof):
            names = ('host_enqueue', 'smpl_tables_done', 'encoder_done', 'rays_at_encoder_join', 'gather_done', 'mlp_done', 'frame_done')
            res['frame_timeline_ms'] = {k: round(float(v), 4) for k, v in zip(names, prof[:, :7].mean(0))}
            res['frame_timeline_ms']['host_per_step_python'] = round(1e3 * host_dt / a.steps, 4)
        if not a.no_cpu_baseline and world == 1:            # reported at N = 1 only (rank 0's host cores)
            res['cpu_baseline'] = cpu_baseline(a.config)
        if not a.no_torch_gpu_baseline and world == 1:
            import tempfile
            path = os.path.join(tempfile.mkdtemp(prefix='sherf_bench_'), 'oracle_frame.npz')
            res['torch_gpu_baseline'] = torch_gpu_baseline_child(a, lrank, save=path)
            if res['torch_gpu_baseline'].get('value'):
                res['torch_gpu_baseline']['speedup_vs_it'] = res['value'] / res['torch_gpu_baseline']['value']
            if os.path.exists(path):             # BASELINE's "PSNR vs ref": the timed frame against the oracle's image of the same frame
                res['parity'] = frame_parity(step().detach().float().cpu().numpy(), np.load(path))
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def cpu_baseline(cfg_name):
    """The oracle (CPU port of the reference algorithm, brute-force K-NN included) timed on the host cores, on a
    bounded sample of the same workload: a centred 48x48-ray crop of the 512x512x64 frame."""
    from oracle import fixtures, sherf_oracle as O
    import json as _json
    shapes = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'param_shapes.json')))
    state = {n: torch.from_numpy(fixtures.seeded_param(n, s)) for n, s in shapes.items() if fixtures.seeded_param(n, s) is not None}
    fx = fixtures.renderer_inputs(cfg_name)
    c = fx['cfg']
    H, W, n = c['H'], c['W'], 64
    ys, xs = np.meshgrid(np.arange(H // 2 - n // 2, H // 2 + n // 2), np.arange(W // 2 - n // 2, W // 2 + n // 2), indexing='ij')
    sel = (ys * W + xs).reshape(-1)
    d = {k: (dict(v) if isinstance(v, dict) else v) for k, v in fx['input_data'].items()}
    for k in ('ray_o_all', 'ray_d_all', 'near_all', 'far_all'):
        d[k] = np.ascontiguousarray(d[k][:, :, sel])
    fx = dict(fx); fx['input_data'] = d
    torch.set_num_threads(min(16, os.cpu_count()))     # more threads only add contention at these matrix sizes
    t0 = time.perf_counter()
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True, keep=False)
    dt = time.perf_counter() - t0
    return dict(value=len(sel) / dt, unit='rays/s', cores=torch.get_num_threads(), kind='port',
                sample=f'centred {n}x{n}-ray crop of the {H}x{W}x{c["S"]} frame ({len(sel)} rays, {int(r["mask"].sum())} valid samples), '
                       f'oracle/sherf_oracle.py fp32 torch-CPU, {dt:.1f} s')


def frame_parity(tile, ref):
    """tile [R,5] = (rgb, depth, acc) of the bench frame, ref = the oracle's rgb [R,3] / acc [R] of the same frame (whole 512x512x64
    frame, stock ATen ops on the GPU) -> PSNR on images mapped to [0,1] (test_loop.py:36-37) and the max errors relative to the range."""
    rgb, acc = tile[:, :3].astype(np.float64), tile[:, 4].astype(np.float64)
    r_rgb, r_acc = ref['rgb'].reshape(-1, 3).astype(np.float64), ref['acc'].reshape(-1).astype(np.float64)
    mse = float(np.mean(((rgb / 2 + 0.5) - (r_rgb / 2 + 0.5)) ** 2))
    # per-ray error relative to the range; a sample whose nearest-vertex distance sits within an ulp of the 5 cm shell threshold can fall
    # on either side in two fp32 implementations (the oracle's GEMMs here are rocBLAS's), which moves one ray visibly: the count of rays
    # over the tolerance is reported beside the maximum
    err = np.abs(rgb - r_rgb).max(1) / (np.abs(r_rgb).max() + 1e-12)
    return dict(psnr_vs_oracle_db=float(-10.0 * np.log10(mse)) if mse > 0 else float('inf'), rgb_rel_err=float(err.max()),
                rgb_rel_err_p9999=float(np.quantile(err, 0.9999)), rays_over_tolerance=int((err > 1e-3).sum()), rays=int(err.size),
                acc_rel_err=float(np.abs(acc - r_acc).max() / (np.abs(r_acc).max() + 1e-12)), tolerance=1e-3,
                oracle='oracle/sherf_oracle.py (pinned to the unmodified reference) as stock ATen fp32 ops on the GPU, whole frame')


def torch_gpu_baseline_child(a, lrank, timeout=240, save=None):
    """torch_gpu_baseline in a child process: checker code on stock kernels must not be able to take the bench line down (out of
    memory, a hang) nor to leave its allocator pool in this process."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['LOCAL_RANK'] = str(lrank)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--torch-gpu-child', '--config', a.config, '--bn-mode', a.bn_mode]
                           + (['--save-oracle', save] if save else []),
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('TORCH_GPU_JSON ')]
        if not line:
            return dict(error=f'child rc={r.returncode}: {r.stderr.strip()[-300:]}')
        return json.loads(line[-1][len('TORCH_GPU_JSON '):])
    except Exception as ex:
        return dict(error=f'{type(ex).__name__}: {str(ex)[:300]}')


def torch_gpu_baseline(cfg_name, dev, training, iters=1, save=None):
    """The same oracle as cpu_baseline, run through PyTorch-ROCm's stock kernels on the GPU over the WHOLE frame (checker code
    timed as a baseline, never on the product path).  Its K-NN is the blocked brute force of oracle.nearest_vertex."""
    try:
        from oracle import fixtures, sherf_oracle as O
        import json as _json
        shapes = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'param_shapes.json')))
        state = {n: torch.from_numpy(fixtures.seeded_param(n, s)).to(dev) for n, s in shapes.items() if fixtures.seeded_param(n, s) is not None}
        bench_cfg = dict(fixtures.CONFIGS[cfg_name]); bench_cfg['theta_tgt'] = 0.4          # rank 0's frame of the measurement (make_inputs)
        fixtures.CONFIGS['_bench'] = bench_cfg
        fx = fixtures.renderer_inputs('_bench')
        c = fx['cfg']
        O.NN_CHUNK = 32768                      # 32768 x 6890 distance blocks: large launches, < 4 GB of temporaries
        times = []
        with torch.no_grad():
            for it in range(iters + 1):         # first pass = warm-up (allocator, kernel load)
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                r = O.render_from_fixture(fx, state, training=training, keep=False, device=dev)
                torch.cuda.synchronize(dev); times.append(time.perf_counter() - t0)
        dt = min(times[1:])
        R = c['H'] * c['W']
        if save:                                # the oracle's image of the bench frame: what the parent's `parity` entry compares with
            np.savez(save, rgb=r['rgb'].detach().float().cpu().numpy(), acc=r['acc'].detach().float().cpu().numpy())
        return dict(value=R / dt, unit='rays/s', kind='port', seconds_per_frame=dt,
                    sample=f'whole {c["H"]}x{c["W"]}x{c["S"]} frame ({int(r["mask"].sum())} valid samples), oracle/sherf_oracle.py as stock '
                           f'PyTorch-ROCm fp32 ops on the GPU, best of {iters} after 1 warm-up')
    except Exception as e:                      # a baseline must never take the bench line down with it
        return dict(error=f'{type(e).__name__}: {e}'[:300])


if __name__ == '__main__':
    main()
