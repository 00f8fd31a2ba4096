"""rays/s of SHERF's volumetric-rendering hot path (ImportanceRenderer.forward) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one 512x512 frame x 64 samples/ray of synthetic input already resident
in HBM (BASELINE.json config 2: single subject novel view, framed so that 7.6 % of the samples are valid -- the fraction SURVEY.md
section 8(d) sized the path on).  With N > 1 every rank renders its own target view
of the same subject (BASELINE config 4: views sharded across GPUs, weak scaling) and the step ends with the RCCL
all_gather of the rendered [rays, 5] tiles (asynchronous, awaited one step later; the last one inside the timed region).
Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its N ranks ITSELF (re-executes under
torch.distributed.run on 127.0.0.1, one rank per GPU, as the reference spawns its own: sherf/train.py:98-103); launched by a
torchrun-style launcher it checks --gpus against WORLD_SIZE and refuses a mismatch.
"""
import argparse
import json
import os
import sys
import time

# With `--streams N > 1` frames are issued round-robin on N caller streams, each with its own pair of side streams: more HIP streams
# than the runtime's default of 4 hardware queues, onto which it would multiplex them (two chains sharing a queue run one after the
# other: DESIGN section 7).  Must be set before the HIP runtime initialises; harmless with one stream.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_VALID_SAMPLE = 429248       # 214,624 MAC per valid sample (fusion + transformer + decoder)
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                # MI355X HBM3E (MI355X_MICROARCH.md)
# (the host dry run of this script lowers them)
SECONDARY_ITERS = dict(mlp=20, mlp_warmup=5, frames=20, frames_warmup=10)


def _use_host_build():
    """TEST INFRASTRUCTURE (tests/test_hipcpu_frame.py, tests/test_dist_cpu.py): SHERF_HIPCPU_LIB names a HOST build of the kernel library
    (tests/hipcpu) -- the script then runs its whole plumbing (ranks, process group, gathers, the JSON line) on CPU tensors over gloo.
    The numbers mean nothing there.  Never set on a GPU box."""
    import ctypes
    from sherf_amd import _lib
    import sherf_amd.renderer as AR
    _lib.LIB_PATH, _lib._lib = os.environ['SHERF_HIPCPU_LIB'], None
    _lib.ptr = lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
    _lib.stream = lambda: ctypes.c_void_p(0)
    torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
    torch.cuda.synchronize = lambda dev=None: None
    torch.Tensor.is_cuda = property(lambda self: True)
    AR.ImportanceRenderer._side = lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})()
    AR.ImportanceRenderer.SMPL_NEUTRAL = property(lambda self: self._smpl(torch.device('cpu')))
    globals()['_device'] = lambda lrank: torch.device('cpu')
    os.environ.setdefault('SHERF_DIST_BACKEND', 'gloo')
    if os.environ.get('SHERF_HIPCPU_LIB_BWD'):            # (bench_train.py on the host build: the backward library too)
        from sherf_amd import backward_dense
        _lib.LIB_BWD_PATH, _lib._lib_bwd = os.environ['SHERF_HIPCPU_LIB_BWD'], None
        backward_dense.HipOps._p = staticmethod(lambda m: ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off))


def launch_ranks(n, script=None, argv=None):
    """`--gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 (the reference
    spawns its ranks itself too: sherf/train.py:98-103).  The ranks' output passes through; -> the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(script or __file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.run(cmd, env=env).returncode


def contextlib_null():
    import contextlib
    return contextlib.nullcontext()


def check_world(gpus, world, what='bench.py'):
    if gpus != world:
        print(f'{what}: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a line for the wrong N', file=sys.stderr)
        sys.exit(2)


def make_inputs(cfg_name, theta, dev):
    from synthdata import fixtures     # seeded synthetic inputs; nothing from oracle/ on the timed path
    c = dict(fixtures.CONFIGS[cfg_name])
    c['theta_tgt'] = theta
    fixtures.CONFIGS['_bench'] = c
    fx = fixtures.renderer_inputs('_bench')
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {k: ({kk: to(vv) for kk, vv in v.items()} if isinstance(v, dict) else to(v)) for k, v in fx['input_data'].items()}
    return fx, d, to


def fixtures_variant_note(cfg):
    from synthdata import fixtures
    return ('reference-init: every parameter from the distribution of the reference constructors, alpha_linear.bias + 5 (SURVEY 8(d)); band-limited tables'
            if fixtures.variant_of(cfg) == 'ri' else 'adversarial seeded weights (2.4 x the default scale, density head x 20), white-noise tables')


def _device(lrank):
    # SHERF_LOCAL_DEVICE (testing only): every rank on that device -- a multi-rank dry run of the whole script on a one-GPU box
    # (tools/history/gpu_r4_ab.sh: 2 ranks x 4 caller streams with SHERF_DIST_BACKEND=gloo); never set by the driver
    if os.environ.get('SHERF_LOCAL_DEVICE'):
        lrank = int(os.environ['SHERF_LOCAL_DEVICE'])
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
    variant = fixtures.variant_of(a.config)
    fixtures.load_seeded_state(rend, 'renderer.', variant); fixtures.load_seeded_state(dec, 'decoder.', variant)
    rend.to(dev).train(a.bn_mode == 'train'); dec.to(dev).train(a.bn_mode == 'train')
    # voxelisation glue (triplane.py:129-137) through the product path
    gen = TriPlaneGenerator.__new__(TriPlaneGenerator)
    torch.nn.Module.__init__(gen); gen.renderer = rend
    can = gen.canonical_obs_vertices(d)
    sp_input, _ = gen.prepare_sp_input(d['t_vertices'].float(), can)
    sp = SparseConvTensor(to(fx['vertex_feat']), sp_input['coord'], sp_input['out_sh'], 1)
    opts = dict(fx['options']); opts['mlp_precision'] = a.precision
    if getattr(a, 'table_precision', None):
        opts['table_precision'] = a.table_precision
    if getattr(a, 'encoder_precision', None):
        opts['encoder_precision'] = a.encoder_precision
    return dict(rend=rend, dec=dec, d=d, sp=sp, sp_input=sp_input, planes=to(fx['planes']), obs_feat=to(fx['obs_feat']),
                obs_img=d['obs_img_all'][:, 0], opts=opts)


def fresh_copy(d):
    """The frame's inputs with the tensors the valid-sample count depends on CLONED (rays, depth range, posed vertices, the SMPL frame's global
    rotation / translation): what a caller rendering a sequence hands over -- new tensors every frame."""
    d2 = dict(d)
    for k in ('ray_o_all', 'ray_d_all', 'near_all', 'far_all', 'vertices'):
        d2[k] = d[k].clone()
    d2['params'] = dict(d['params'])
    for k in ('R', 'Th'):
        d2['params'][k] = d['params'][k].clone()
    return d2


def render_frame(w):
    d = fresh_copy(w['d']) if w.get('fresh') else w['d']
    with torch.no_grad():
        return w['rend'](w['planes'], w['obs_img'], w['obs_feat'], w['sp'], None, w['sp_input'], w['dec'], d['ray_o_all'][:, 0],
                         d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, w['opts'])


def time_frames(w, steps, warmup, dev, streams=None):
    """ms per frame of `steps` back-to-back frames after `warmup` (device synchronised on both sides); `streams`: a list of caller streams
    the frames are issued on round-robin (None: the current stream, one frame in flight)."""
    def frame(i):
        if not streams:
            return render_frame(w)
        with torch.cuda.stream(streams[i % len(streams)]):
            return render_frame(w)
    for i in range(warmup):
        frame(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        frame(i)
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / steps


MLP_FORM_ENTRY = dict(one='sherf_nerf_mlp', pipelined='sherf_nerf_mlp3', two_tiles='sherf_nerf_mlp2')
MLP_FORM_KERNEL = dict(one='nerf_mlp_kernel', pipelined='nerf_mlp3_kernel', two_tiles='nerf_mlp2_kernel')


def mlp_kernel_alone(w, precision, dev, iters=None, warmup=None, exact_grid=False, split=False, form='one'):
    """`sherf_nerf_mlp` (split: `sherf_nerf_mlp_split`, the two-launch form) alone on the tokens of the frame `w` rendered last, in
    `precision`: (ms per call from HIP events on the launch stream, its [nv, 4] output)."""
    import ctypes as ct
    from sherf_amd import _lib
    from sherf_amd.renderer import MLP_PRECISIONS
    iters = SECONDARY_ITERS['mlp'] if iters is None else iters
    warmup = SECONDARY_ITERS['mlp_warmup'] if warmup is None else warmup
    rend = w['rend']
    ws, cap = rend.last['ws'], int(rend.last['cap'])
    wc = rend._weights(w['dec'], dev, precision)
    nv = int(ws['counters'][0])
    if exact_grid:                                  # launch for the frame's own sample count instead of the buffers' capacity (R * S): no workgroup
        cap = max((nv + 255) // 256 * 256, 256)     # that only reads the count and exits
    out = torch.empty(max((nv + 31) // 32 * 32, 32), 4, device=dev)
    A = _lib.addr
    st = torch.cuda.current_stream(dev)
    stream = ct.c_void_p(st.cuda_stream)
    if split:
        zfrag = rend._workspace(dev).zfrag(cap, precision, dev)
        launch = lambda: _lib.call('sherf_nerf_mlp_split', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']),
                                   MLP_PRECISIONS[precision], cap, A(zfrag), A(out), stream)
    else:
        launch = lambda: _lib.call(MLP_FORM_ENTRY[form], A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']),
                                   MLP_PRECISIONS[precision], cap, A(out), stream)
    for _ in range(warmup):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        launch()
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters, out[:nv].clone()


def secondary_measurements(a, w, dev, nv, R):
    """Context beside the headline (N = 1 only, after the timed region): the MLP kernel alone in every precision on the headline
    frame's tokens, each with its measured error against the fp32-grade f16x3 output (true relative error, floors of oracle/parity.py),
    and the frame time of more workloads: BASELINE config 3 (novel pose), cfg2 framed so that the valid-sample fraction matches
    SURVEY 8(d)'s probe value (0.076 instead of 0.041; also promoted to `value_dense`), and the ADVERSARIAL seeded weights / white-noise
    tables of rounds 1-2 (which `auto` keeps on f16x3)."""
    out = {}
    try:
        ms3, ref = mlp_kernel_alone(w, 'f16x3', dev)
        sig = ref[:, 3].clamp(min=0)
        fl = lambda ms: nv * FLOP_PER_VALID_SAMPLE / (ms * 1e-3) / 1e12
        rows = dict(f16x3=dict(kernel_ms=ms3, achieved_tflops=fl(ms3), frac=fl(ms3) / PEAK_BF16_TFLOPS, mfma_per_product=3))
        rows['f16_exact_grid'] = dict(kernel_ms=mlp_kernel_alone(w, 'f16', dev, exact_grid=True)[0],
                                      note='the f16 launch sized for the valid samples instead of the capacity R * S: what the empty workgroups of the in-frame launch cost')
        for name in ('f16', 'bf16'):
            ms1, got = mlp_kernel_alone(w, name, dev)
            rows[name] = dict(kernel_ms=ms1, achieved_tflops=fl(ms1), frac=fl(ms1) / PEAK_BF16_TFLOPS, mfma_per_product=1, form='one launch (nerf_mlp_kernel)',
                              sigma_rel_err_max_vs_f16x3=float(((got[:, 3].clamp(min=0) - sig).abs() / sig.clamp(min=1.0)).max()),
                              rgb_rel_err_max_vs_f16x3=float(((got[:, :3] - ref[:, :3]).abs() / ref[:, :3].abs().clamp(min=0.1)).max()))
            ms2, got2 = mlp_kernel_alone(w, name, dev, split=True)
            for form in ('pipelined', 'two_tiles'):                  # round 5's launch forms of the single-product network (the product default: pipelined)
                msf, gotf = mlp_kernel_alone(w, name, dev, form=form)
                rows[f'{name}_{form}'] = dict(kernel_ms=msf, achieved_tflops=fl(msf), frac=fl(msf) / PEAK_BF16_TFLOPS, mfma_per_product=1,
                                              form=f'one launch ({MLP_FORM_KERNEL[form]})', bit_identical_to_one_tile_kernel=bool(torch.equal(got, gotf)))
            rows[name + '_two_launches'] = dict(kernel_ms=ms2, achieved_tflops=fl(ms2), frac=fl(ms2) / PEAK_BF16_TFLOPS, mfma_per_product=1,
                                                form='nerf_tokens_kernel + nerf_decoder_kernel (sherf_nerf_mlp_split)',
                                                bit_identical_to_one_launch=bool(torch.equal(got, got2)))
        ms3b, got3 = mlp_kernel_alone(w, 'f16x3', dev, split=True)
        rows['f16x3_two_launches'] = dict(kernel_ms=ms3b, frac=fl(ms3b) / PEAK_BF16_TFLOPS, mfma_per_product=3, bit_identical_to_one_launch=bool(torch.equal(ref, got3)))
        out['mlp_kernel_alone'] = rows
    except Exception as ex:
        out['mlp_kernel_alone'] = dict(error=f'{type(ex).__name__}: {str(ex)[:200]}')
    ri = '_ri' if a.config.endswith('_ri') else ''
    for cfg in ('cfg2' + ri, 'cfg3' + ri, 'cfg2_dense' + ri, 'cfg2'):
        if cfg == a.config or cfg in out:
            continue
        try:
            b = argparse.Namespace(**vars(a)); b.config = cfg
            w2 = make_workload(b, 0.4, dev)
            w2['rend'].exact_grids = w['rend'].exact_grids
            import ctypes as _ct
            from sherf_amd import _lib as _abi
            time_frames(w2, 1, SECONDARY_ITERS['frames_warmup'], dev)
            _abi.call('sherf_profile_frames', 1)
            ms = time_frames(w2, SECONDARY_ITERS['frames'], 0, dev)
            buf = (_ct.c_float * (64 * 8))(); n_ms = _ct.c_int32(0)
            _abi.call('sherf_profile_frames_read', buf, 64, _ct.byref(n_ms))
            _abi.call('sherf_profile_frames', 0)
            prof = np.array(buf[:n_ms.value * 8], dtype=np.float64).reshape(-1, 8)
            nv2 = int(w2['rend'].last['ws']['counters'][0])
            S = w2['opts']['depth_resolution']
            names = ('host_enqueue', 'smpl_tables_done', 'encoder_done', 'rays_at_encoder_join', 'gather_done', 'mlp_done', 'frame_done', 'mlp_kernel')
            out[cfg] = dict(ms_per_frame=ms, rays_per_s=R / (ms * 1e-3), valid_samples=nv2, valid_fraction=nv2 / (R * S),
                            mlp_precision=w2['rend'].last.get('mlp_precision'), frames_in_flight=1,
                            frame_timeline_ms={k: round(float(v), 4) for k, v in zip(names, prof.mean(0))} if len(prof) else None)

            del w2
        except Exception as ex:
            out[cfg] = dict(error=f'{type(ex).__name__}: {str(ex)[:200]}')
    return out


def pmc_traffic(a, lrank, timeout=150, child=None, keys=None):
    """HBM bytes per launch of nerf_mlp_kernel from rocprofv3's counters, measured on THIS box right after the run: two `--pmc` passes
    (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md) of a 3-frame child, averaged over the kernel's dispatches;
    FETCH_SIZE doubled as that guide prescribes for gfx950 (128-byte requests tallied at 64).  -> dict or {'error': ...}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return dict(error='rocprofv3 not found')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(LOCAL_RANK=str(lrank), TMPDIR='/tmp')
    vals = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        outdir = tempfile.mkdtemp(prefix='sherf_pmc_', dir='/tmp')
        # (`child` / `keys`: another command and other kernel names under the same two counter passes -- bench_train.py's tap scatter)
        cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', outdir, '--'] + (list(child) if child else
              [sys.executable, os.path.abspath(__file__), '--pmc-child', '--config', a.config, '--precision', getattr(a, 'precision_used', a.precision), '--bn-mode', a.bn_mode])
        try:
            r = subprocess.run(cmd, env=env, cwd='/tmp', stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
            rows = []
            for f in glob.glob(os.path.join(outdir, '**', '*counter_collection.csv'), recursive=True):
                rows += list(csv.DictReader(open(f)))
            per = {}                                  # the network's kernels (one launch, or the two of sherf_nerf_mlp_split): mean per dispatch, summed
            for x in rows:
                kn = x.get('Kernel_Name', '')
                for key in (keys or ('nerf_mlp_kernel', 'nerf_mlp2_kernel', 'nerf_mlp3_kernel', 'nerf_tokens_kernel', 'nerf_decoder_kernel')):
                    if key in kn and x.get('Counter_Name') == counter:
                        per.setdefault(key, []).append(float(x['Counter_Value']))
            if not per:
                return dict(error=f'{counter}: no nerf_*_kernel rows (rc={r.returncode}): {r.stderr.strip()[-200:]}')
            vals[counter] = (sum(sum(v) / len(v) for v in per.values()), max(len(v) for v in per.values()), {k: sum(v) / len(v) for k, v in per.items()})
        except Exception as ex:
            return dict(error=f'{counter}: {type(ex).__name__}: {str(ex)[:200]}')
        finally:
            shutil.rmtree(outdir, ignore_errors=True)
    fetch_kb, write_kb = vals['FETCH_SIZE'][0], vals['WRITE_SIZE'][0]
    return dict(hbm_bytes_per_launch=int(2 * fetch_kb * 1024 + write_kb * 1024), fetch_size_kb_raw=fetch_kb, write_size_kb_raw=write_kb,
                dispatches=vals['FETCH_SIZE'][1], fetch_kb_by_kernel=vals['FETCH_SIZE'][2], write_kb_by_kernel=vals['WRITE_SIZE'][2],
                note='FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE as reported; the network\'s kernels summed per frame')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='cfg2_dense_ri',
                    help='cfg2_dense_ri (default) = BASELINE config 2 framed at the valid-sample fraction SURVEY 8(d) sized the path on (7.6 %%), with '
                         'the network SURVEY 8(d) specifies (the reference constructors\' initialisation, alpha bias + 5) and band-limited tables; '
                         'cfg2_ri = the same subject framed wider (4.1 %% valid: round 3\'s headline, now in `secondary`); cfg2 = the adversarial '
                         'seeded weights of rounds 1-2')
    ap.add_argument('--precision', default='auto', choices=['auto', 'f16x3', 'f16', 'bf16'],
                    help='MLP operand precision; auto (the product default) = calibrated per set of weights on the first frame')
    ap.add_argument('--table-precision', default=None, choices=['f32', 'f16'], help='override the folded tables\' format (default: follows the MLP precision)')
    ap.add_argument('--encoder-precision', default=None, choices=['f16x3', 'f16'], help='override the sparse convolutions\' operand precision (default: follows the tables)')
    ap.add_argument('--streams', type=int, default=1,
                    help='caller streams the frames are issued on, round-robin (each frame is one ImportanceRenderer.forward on its stream; every '
                         'stream has its own workspace).  1 (default since round 5) = ONE frame in flight: the metric SURVEY 8(d) defines (R / wall '
                         'time of one ImportanceRenderer.forward) and what the reference\'s strictly sequential evaluation loop does -- `value`, the '
                         'roofline and the frame timeline all come from this one process.  The throughput with four frames in flight (round 4\'s '
                         'headline: the low-occupancy first phase of three frames under the chip-filling kernels of a fourth, 1.70 -> 1.49 ms per '
                         'frame at 3.4 x the latency) is measured by a child run and reported beside it as `value_frames_overlapped`')
    ap.add_argument('--overlap-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--fresh-inputs', action='store_true',
                    help='every frame gets FRESH input tensors (rays, depth range, vertices, SMPL parameters cloned per frame): the identity caches of '
                         'the renderer miss as they do on a real sequence, so the token-capacity check behind the sampler runs (one host wait per '
                         'frame).  Default off: the same tensors every frame, like a renderer turned around a fixed subject with unchanged rays; '
                         'the N = 1 line reports both (`secondary.fresh_inputs`)')
    ap.add_argument('--partition', default='views', choices=['views', 'rays'],
                    help='N > 1: views (default, BASELINE config 4: every rank renders its own target view, weak scaling) or rays (ONE frame '
                         'cut into interleaved 1024-ray tiles over the ranks, sherf_amd.dist.ray_tiles: strong scaling, value = the frame\'s '
                         'rays / time); either way one RCCL all_gather of the rendered tiles per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-torch-gpu-baseline', action='store_true',
                    help='skip timing the oracle (the reference algorithm as stock ATen ops, brute-force K-NN) ON THE GPU over the whole '
                         'frame: the "reference single-GPU render()" denominator SURVEY.md section 8(d) asks for beside the CPU one, and '
                         'the source of the `parity` entry (N = 1 only; runs in a child process after the measurement)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the bf16 kernel line and the cfg3 / cfg2_dense frame timings (N = 1 only)')
    ap.add_argument('--no-train', action='store_true', help='skip the `train` entry: bench_train.py (BASELINE config 5: forward + backward through the HIP '
                                                            'kernels + flat-gradient exchange + Adam) run for a few steps in a child process (N = 1 only)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc child passes that measure roofline.traffic (N = 1 only)')
    ap.add_argument('--exact-grids', action='store_true',
                    help='size the launches behind the compaction by the frame\'s own sample count (one host wait per frame: sherf_hip.h)')
    ap.add_argument('--torch-gpu-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--save-oracle', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--bn-mode', default='train', choices=['train', 'eval'],
                    help='BatchNorm of the voxel encoder. train (default) = batch statistics: the mode the reference renders in, '
                         'also at test time (eval_*.sh -> train.py --test_flag -> test(G, ...) with G built .train(), '
                         'training_loop.py:193,311-330); eval = running statistics (G_ema.eval(), training_loop.py:196)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ and not (a.torch_gpu_child or a.pmc_child):
        sys.exit(launch_ranks(a.gpus))
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); lrank = int(os.environ.get('LOCAL_RANK', 0))
    if not (a.torch_gpu_child or a.pmc_child):
        check_world(a.gpus, world)
    if os.environ.get('SHERF_HIPCPU_LIB'):
        _use_host_build()
    if a.torch_gpu_child:
        print('TORCH_GPU_JSON ' + json.dumps(torch_gpu_baseline(a.config, _device(lrank), a.bn_mode == 'train', save=a.save_oracle)), flush=True)
        return
    dev = _device(lrank)
    if a.pmc_child:                          # three frames under rocprofv3 --pmc (pmc_traffic): nothing else
        w = make_workload(a, 0.4, dev)
        if os.environ.get('SHERF_BENCH_MLP_FORM'):         # the form `auto` chose in the parent: no timing launches of the other forms under the counters
            w['opts']['mlp_form'] = os.environ['SHERF_BENCH_MLP_FORM']
        time_frames(w, 3, 1, dev)
        return
    if world > 1:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get('SHERF_DIST_BACKEND', 'nccl')          # 'nccl' = RCCL; 'gloo' only for the CPU dry-run in tests/
        # a rank that dies must take the job down instead of leaving the others in the all_gather: bounded collective timeout here,
        # and every exception below leaves through os._exit (the launcher then stops the remaining ranks)
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get('SHERF_DIST_TIMEOUT', '300'))),
                                **(dict(device_id=dev) if backend == 'nccl' else {}))
    from sherf_amd import dist as sdist  # noqa: F401

    if os.environ.get('SHERF_DEBUG'):
        from sherf_amd import _lib as _dbg
        _dbg.lib().sherf_set_debug(int(os.environ['SHERF_DEBUG'], 0))     # ablation runs only (sampler: 1 = no candidates, 2 = every sample)
    rays_mode = world > 1 and a.partition == 'rays'
    w = make_workload(a, 0.4 if rays_mode else 0.4 + rank * 2 * np.pi / max(world, 1), dev)
    w['fresh'] = bool(a.fresh_inputs)
    rend, opts = w['rend'], w['opts']
    R_frame = w['d']['ray_o_all'].shape[2]
    if rays_mode:
        # this rank's interleaved tiles of THE frame (sherf_amd.dist: balances the body's footprint); the depth image is clamped with
        # the whole frame's depth range (ray_marcher.py:57), computed from the full near / far every rank holds -- no collective
        tile_rays = 1024 if R_frame >= 16 * 1024 else 128              # (small test frames: still several tiles per rank)
        idx = sdist.ray_tiles(R_frame, rank, world, tile=tile_rays).to(dev)
        opts['depth_range'] = sdist.depth_range(w['d']['near_all'][:, 0], w['d']['far_all'][:, 0])
        for k in ('ray_o_all', 'ray_d_all', 'near_all', 'far_all'):
            w['d'][k] = w['d'][k][:, :, idx].contiguous()
        n_pad = sdist.padded_shard_size(R_frame, world, tile_rays)
    rend.exact_grids = bool(a.exact_grids) or rend.exact_grids
    R = w['d']['ray_o_all'].shape[2]; S = opts['depth_resolution']
    from sherf_amd import _lib as _abi
    import ctypes as _ct

    n_streams = max(1, int(a.streams))
    if dev.type == 'cuda':
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream(dev)]
    else:
        # the host build of the tests has no streams: `--streams N` there still walks the round-robin bookkeeping (one gather queue per
        # caller stream, below) with every "stream" the host itself -- what tests/test_hipcpu_frame.py runs with two gloo ranks
        import contextlib
        streams = [None] * n_streams
        torch.cuda.stream = lambda st: contextlib.nullcontext()
    torch.cuda.synchronize(dev)                  # the workload's setup (default stream) is complete before any frame stream starts
    counter = [0]

    def step():
        if n_streams == 1:
            return frame_on_current_stream(0)
        k = counter[0] % n_streams
        counter[0] += 1
        with torch.cuda.stream(streams[k]):
            return frame_on_current_stream(k)

    def frame_on_current_stream(k):
        render_frame(w)
        tile = rend.last['out']          # the frame's output buffer as the kernel wrote it: planar [rgb (3R) | depth (R) | acc (R)], no repacking
        if rays_mode and tile.numel() != 5 * n_pad:           # ranks own 1024-ray tiles: equal shard sizes for the gather
            tile = torch.nn.functional.pad(tile, (0, 5 * n_pad - tile.numel()))
        if world > 1:
            # the gather of this frame's tiles runs on RCCL's own stream (async_op) and is awaited one step later, so that it
            # overlaps the next frame's sampling instead of stalling the render stream for a latency-bound 5 MB exchange
            # One queue PER CALLER STREAM (round 5): a frame waits for the previous gather issued on ITS stream only -- with one shared
            # queue (round 4) every frame's stream was made to wait for a gather that belongs to another stream's frame, which serialises
            # the streams the overlap is built from.  RCCL orders the collectives themselves (same issue order on every rank: the
            # round-robin is deterministic).
            out = [torch.empty_like(tile) for _ in range(world)]
            q = inflight[k]
            q.append((torch.distributed.all_gather(out, tile, async_op=True), out, tile))
            gather_log.append((k, counter[0]))
            while len(q) > 1:
                q.pop(0)[0].wait()
        return tile

    inflight = [[] for _ in range(n_streams)]
    gather_log = []

    def drain():                      # every gather issued so far is complete (on its render stream's timeline) after this
        for k, q in enumerate(inflight):
            with torch.cuda.stream(streams[k]) if n_streams > 1 else contextlib_null():
                while q:
                    q.pop(0)[0].wait()

    for _ in range(n_streams if n_streams > 1 else 0):     # every stream's workspace exists (and `auto` is calibrated) before the warm-up proper
        step()
    for _ in range(a.warmup):
        step()
    drain()
    # Round 6: frames CAN replay as hipGraphs (csrc/frame.hip; SHERF_FRAME_GRAPH=1 and rendering option aux_stream=False -- opt-in: measured on the
    # MI355X, a replayed frame takes the GPU as long as an enqueued one).  A replayed graph carries no per-frame timing events, so with graphs on the
    # timed region runs WITHOUT the driver's HIP events and a second pass of frames right behind it (same process, frames enqueued launch by launch
    # with the events around every stage on its launch stream) measures the kernel's duration and the timeline.  Default: graphs off, events inside
    # the timed region as in rounds 2-5.
    graphs_on = os.environ.get('SHERF_FRAME_GRAPH', '0') == '1' and dev.type == 'cuda'
    _abi.call('sherf_profile_frames', 0 if graphs_on else 1)       # HIP events around sherf_nerf_mlp etc. on their launch streams (csrc/frame.hip)
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
    rank_devices = [str(dev)]
    if world > 1:
        rank_devices = [None] * world
        torch.distributed.all_gather_object(rank_devices, f'rank {rank}: {dev}' + (f' ({torch.cuda.get_device_name(dev)})' if dev.type == 'cuda' else ''))
    parity_failed = False
    gstats = (_ct.c_int64 * 4)()
    if hasattr(_abi.lib(), 'sherf_frame_graph_stats'):
        _abi.call('sherf_frame_graph_stats', gstats, 4)
    if graphs_on:                                  # the events' pass: see above
        _abi.call('sherf_profile_frames', 1)
        for _ in range(max(8, min(a.steps, 32))):
            step()
        drain()
        torch.cuda.synchronize()
    ms = (_ct.c_float * (64 * 8))(); n_ms = _ct.c_int32(0)
    _abi.call('sherf_profile_frames_read', ms, 64, _ct.byref(n_ms))
    _abi.call('sherf_profile_frames', 0)
    # the HOST's own cost of a step (Python + the native enqueue), measured on an EMPTY queue: in the timed loop above the host runs
    # ahead of the GPU until the queue pushes back, so its wall time there is the GPU's (what `host_per_step_python` used to report)
    t1 = time.perf_counter()
    for _ in range(3):
        step()
    host_alone = (time.perf_counter() - t1) / 3
    drain()
    torch.cuda.synchronize()
    prof = np.array(ms[:n_ms.value * 8], dtype=np.float64).reshape(-1, 8)
    # With frames of several caller streams in flight the HIP events around a launch measure its WALL time, which includes what it cedes to
    # the other frames' kernels -- not the kernel's own cost, and the per-frame timeline is a latency, not a schedule.  The roofline, the
    # timeline and the secondary lines therefore come from a CHILD process that renders one frame at a time (`--streams 1`: this file, same
    # workload, same box, right after the timed region); in this process they would also be distorted by HIP's mapping of a dozen streams
    # onto four hardware queues (measured: 2.9 instead of 1.7 ms per frame for one frame in flight after a four-stream run).
    prof_overlap, one_frame = None, None
    if n_streams > 1 and world == 1 and rank == 0 and not a.overlap_child:
        prof_overlap = prof
        one_frame = bench_child(a, lrank, ['--streams', '1', '--steps', str(max(8, min(a.steps, 24))), '--warmup', '6']
                                + (['--no-secondary'] if a.no_secondary else []))
    mlp_ms = float(prof[:, 7].mean()) if len(prof) else None
    if rank == 0:
        used = rend.last.get('mlp_precision', a.precision)                  # what `auto` resolved to for these weights
        dtype = {'f16x3': 'f16x3 MFMA (fp32-grade: operands split hi + lo in fp16, three products, fp32 accumulate), fp32 elsewhere',
                 'f16': 'f16 MFMA (fp16 operands rounded to nearest, one product, fp32 accumulate), fp32 elsewhere',
                 'bf16': 'bf16 MFMA (one product, fp32 accumulate), fp32 elsewhere'}[used]
        res = dict(metric='rendered rays/sec at 512x512x64 samples (ImportanceRenderer.forward)', value=(R_frame if rays_mode else world * R) * a.steps / dt,
                   unit='rays/s', n_gpus=world, rccl_ranks=torch.distributed.get_world_size() if world > 1 else 1, rank_devices=rank_devices,
                   steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * dt / a.steps,
                   higher_is_better=True, scaling='strong' if rays_mode else 'weak', vs_baseline=None, dtype=dtype, data='synthetic',
                   config=dict(workload=f'{a.config}: 512x512 rays x 64 samples, synthetic SMPL subject, novel view, all feature branches, '
                                        f'one view per GPU', rays=R_frame if rays_mode else R, rays_per_rank=R, samples_per_ray=S, valid_samples=nv, valid_fraction=nv / (R * S),
                               parallelism=(f'ray tiles x{world} (one frame)' if rays_mode else f'views x{world}') if world > 1 else 'single GPU', mlp_precision=used, mlp_precision_requested=a.precision,
                               mlp_precision_auto=getattr(rend, 'auto_report', None), network=fixtures_variant_note(a.config),
                               batchnorm=a.bn_mode, exact_grids=bool(rend.exact_grids), caller_streams=n_streams,
                               table_precision=rend.last.get('table_precision'), encoder_precision=rend.last.get('encoder_precision'), pe_in_gather=bool(rend.last.get('pe_in_gather')), mlp_form=rend.last.get('mlp_form'), mlp_form_auto=getattr(rend, 'form_report', None),
                               frame_graphs=dict(on=bool(graphs_on), captured=int(gstats[0]), replayed_frames=int(gstats[1]), enqueued_frames=int(gstats[2]), failed_captures=int(gstats[3])),
                               # workspace_bytes: ONE caller stream's (the largest); sampler side at R * S, token side (484 B / sample) at 1.5 x the frame's valid samples
                               workspace_bytes=max([w_.nbytes() for w_ in (rend._ws or {}).values()] or [0]), workspaces=len(rend._ws or {}), token_capacity=int(rend.last.get('cap', 0)),
                               sampler_capacity=int(rend.last.get('sampler_cap', 0))))
        if one_frame is not None and isinstance(one_frame.get('roofline'), dict):
            res['roofline'] = dict(one_frame['roofline'])
            res['roofline']['timing'] = ('HIP events recorded by the native frame driver around the network\'s launch on its launch stream, mean over the frames of a child run '
                                         'of this file with ONE frame in flight (--streams 1; same workload, same box, right after the timed region): the kernel\'s own '
                                         'duration.  In the timed region frames of several streams share the chip: `kernel_ms_with_frames_overlapping` is a launch\'s wall time there')
            if len(prof_overlap):
                res['roofline']['kernel_ms_with_frames_overlapping'] = float(prof_overlap[:, 7].mean())
        elif mlp_ms:
            ach = nv * FLOP_PER_VALID_SAMPLE / (mlp_ms * 1e-3) / 1e12
            two = bool(rend.last.get('mlp_split'))
            tiles = (nv + 31) // 32
            res['roofline'] = dict(kernel='nerf_tokens_kernel + nerf_decoder_kernel (sherf_nerf_mlp_split: the network as two launches, timed together)' if two
                                   else MLP_FORM_KERNEL.get(rend.last.get('mlp_form', 'one'), 'nerf_mlp_kernel'), bound='mfma', achieved=ach, peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                                   frac=ach / PEAK_BF16_TFLOPS, traffic=None, kernel_ms=mlp_ms,
                                   algorithmic_flop_per_launch=nv * FLOP_PER_VALID_SAMPLE,
                                   # what the matrix pipe EXECUTES: 374 v_mfma_f32_32x32x16 per 32-sample tile and product (the two 1x1 projections of
                                   # SURVEY 8(d)'s count are folded into the tables by other kernels, the transformer skips the token nobody reads)
                                   executed_mfma_flop=tiles * 374 * 32768 * (3 if used == 'f16x3' else 1),
                                   frac_executed=tiles * 374 * 32768 / (mlp_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                   timing=('HIP events recorded by the native frame driver around the network\'s launch(es) on their launch stream, mean over '
                                           + ('a second pass of frames in this process right behind the timed region (the timed frames replay hipGraphs, which carry no per-frame events)' if graphs_on else 'the timed frames'))
                                          + (' (frames of several caller streams overlap: a launch\'s wall time, not the kernel\'s own duration)' if n_streams > 1 else ''))
        if one_frame is not None and isinstance(one_frame.get('frame_timeline_ms'), dict):
            res['frame_timeline_ms'] = dict(one_frame['frame_timeline_ms'], note='one frame in flight (child run with --streams 1)')
            res['frame_timeline_ms']['host_per_step_python'] = round(1e3 * host_alone, 4)
            res['frame_timeline_ms']['host_wall_per_step_in_timed_loop'] = round(1e3 * host_dt / a.steps, 4)
            if len(prof_overlap):
                res['latency_ms_per_frame_with_frames_overlapping'] = round(float(prof_overlap[:, 6].mean()), 4)
            res['value_one_frame_in_flight'] = one_frame.get('value'); res['ms_per_step_one_frame_in_flight'] = one_frame.get('ms_per_step')
        elif len(prof):
            names = ('host_enqueue', 'smpl_tables_done', 'encoder_done', 'rays_at_encoder_join', 'gather_done', 'mlp_done', 'frame_done')
            res['frame_timeline_ms'] = {k: round(float(v), 4) for k, v in zip(names, prof[:, :7].mean(0))}
            res['frame_timeline_ms']['host_per_step_python'] = round(1e3 * host_alone, 4)          # Python + native enqueue, empty queue
            res['frame_timeline_ms']['host_wall_per_step_in_timed_loop'] = round(1e3 * host_dt / a.steps, 4)   # (includes queue back-pressure)
        if a.overlap_child and len(prof):                   # (child of a one-stream run: throughput + latency with frames overlapping, nothing else)
            res['latency_ms_per_frame'] = round(float(prof[:, 6].mean()), 4)
            res['kernel_ms_with_frames_overlapping'] = float(prof[:, 7].mean())
        if n_streams == 1 and world == 1 and not a.overlap_child and len(prof):
            res['latency_ms_per_frame'] = round(float(prof[:, 6].mean()), 4)          # one frame in flight: the frame's own duration on the GPU
        if n_streams == 1 and world == 1 and not a.overlap_child and not a.no_secondary and dev.type == 'cuda':
            # round 4's headline beside the metric proper: the same frames issued round-robin on FOUR caller streams (own workspaces and side
            # streams each) -- whole-job throughput with the first phase of three frames hidden under the chip-filling kernels of a fourth
            ov = bench_child(a, lrank, ['--streams', '4', '--overlap-child', '--steps', str(max(a.steps, 100)), '--warmup', '16', '--no-secondary'])
            res['value_frames_overlapped'] = ov.get('value'); res['ms_per_step_frames_overlapped'] = ov.get('ms_per_step')
            res['latency_ms_per_frame_with_frames_overlapping'] = ov.get('latency_ms_per_frame')
            res['frames_overlapped'] = dict(caller_streams=4, steps=ov.get('steps'), kernel_ms_with_frames_overlapping=ov.get('kernel_ms_with_frames_overlapping'),
                                            workspaces=(ov.get('config') or {}).get('workspaces'), error=ov.get('error'),
                                            note='child run of this file with --streams 4: every frame does all of its work (no cross-frame caching), four frames are in '
                                                 'flight, each on its own 1.6 GB workspace; a frame then takes `latency_ms_per_frame_with_frames_overlapping`')
        if one_frame is not None and one_frame.get('error'):
            res['one_frame_in_flight_child'] = one_frame
        if world == 1 and not a.no_secondary:
            if one_frame is not None:                  # (one frame in flight: measured by the child; see above)
                res['secondary'] = one_frame.get('secondary') or dict(error='no secondary in the child line')
                ri_ = '_ri' if a.config.endswith('_ri') else ''
                if a.config.startswith('cfg2_dense'):  # round 3's headline framing issued the way the headline is: N caller streams, its own child
                    wide_n = bench_child(a, lrank, ['--streams', str(n_streams), '--config', 'cfg2' + ri_, '--steps', str(a.steps), '--warmup', str(max(a.warmup, 3 * n_streams)), '--no-secondary'])
                    if isinstance(res['secondary'].get('cfg2' + ri_), dict) and wide_n.get('value'):
                        res['secondary']['cfg2' + ri_].update({f'rays_per_s_{n_streams}_streams': wide_n['value'], f'ms_per_frame_{n_streams}_streams': wide_n['ms_per_step']})
            else:
                res['secondary'] = secondary_measurements(a, w, dev, nv, R)
            dense = res['secondary'].get('cfg2_dense' + ('_ri' if a.config.endswith('_ri') else ''), {})
            if a.config.startswith('cfg2_dense'):        # the headline IS the framing SURVEY 8(d) sized the path on (valid fraction 0.076)
                dense = dict(rays_per_s=res['value'], ms_per_frame=res['ms_per_step'], valid_fraction=nv / (R * S))
                wide = res['secondary'].get('cfg2' + ('_ri' if a.config.endswith('_ri') else ''), {})
                if wide.get('rays_per_s'):               # round 3's headline framing (valid fraction 0.041), for comparison across rounds
                    k = f'rays_per_s_{n_streams}_streams' if n_streams > 1 and f'rays_per_s_{n_streams}_streams' in wide else 'rays_per_s'
                    res['value_cfg2_wide_framing'] = wide[k]; res['ms_per_step_cfg2_wide_framing'] = R / wide[k] * 1e3
                    res['value_cfg2_wide_framing_one_frame_in_flight'] = wide['rays_per_s']
            if dense.get('rays_per_s'):                  # the valid-sample fraction SURVEY 8(d) sized the path on (0.076): first class
                res['value_dense'] = dense['rays_per_s']; res['ms_per_step_dense'] = dense['ms_per_frame']
                res['valid_fraction_dense'] = dense['valid_fraction']
        # what the multi-GPU steps exchange, so that the first real SCALE run can be held against a prediction (no curve has been measured by
        # us: one-GPU boxes).  Views / ray tiles: ONE all_gather of the frame's planar [5 R] fp32 output per step, asynchronous, awaited one frame
        # later on the same caller stream -> off the critical path unless it takes longer than a frame; bandwidth model: every rank receives
        # (N - 1) x 5 R x 4 bytes over its 7 xGMI links (~153 GB/s each, MI355X_MICROARCH.md) + ~20 us of launch / latency.
        gather_bytes = 5 * (n_pad if rays_mode else R) * 4
        res['config']['exchange'] = dict(collective='all_gather (RCCL), one per step, async', bytes_per_rank=gather_bytes, bytes_received_per_rank_at_8_gpus=7 * gather_bytes,
                                         predicted_us_at_8_gpus=round(20 + 7 * gather_bytes / (7 * 153e9) * 1e6, 1), on_critical_path=False,
                                         gathers_issued=len(gather_log), training_step_all_reduce_bytes='<= 219 MB: the flat gradient of the generator (28.7 M + 23.4 M + 0.3 M parameters x 4 B), training_loop.py:374-383',
                                         measured_scaling='none by us (one-GPU boxes): the driver\'s SCALE run is the measurement')
        res['config']['inputs'] = ('FRESH tensors every frame (--fresh-inputs)' if a.fresh_inputs else
                                   'the same input tensors every frame (the renderer\'s identity caches hit: no host wait per frame); `secondary.fresh_inputs` = new tensors per frame')
        if world == 1 and not a.no_secondary and not a.overlap_child and n_streams == 1 and not a.fresh_inputs and isinstance(res.get('secondary'), dict):
            try:                                            # the sequence case -- rays / vertices / pose in new tensors every frame
                w['fresh'] = True
                st0 = dict((rend.__dict__.get('_flags') or {}))
                ms_f = time_frames(w, max(20, min(a.steps, 100)), 5, dev)
                st1 = rend.__dict__.get('_flags') or {}
                res['secondary']['fresh_inputs'] = dict(ms_per_frame=ms_f, rays_per_s=R / (ms_f * 1e-3), frames_in_flight=1,
                                                        token_rerenders=int(st1.get('token_rerenders', 0)) - int(st0.get('token_rerenders', 0)),
                                                        note='rays, depth range, posed vertices and the SMPL frame cloned into NEW tensors every frame: the valid-sample count is '
                                                             'read back right behind the sampler (one host wait per frame, the rest of the frame stays in flight) and checked '
                                                             'against the token workspace; the clones themselves (8.5 MB) are inside the timed region')
            except Exception as ex:
                res['secondary']['fresh_inputs'] = dict(error=f'{type(ex).__name__}: {str(ex)[:200]}')
            finally:
                w['fresh'] = False
        ours = None
        if world == 1 and not a.no_torch_gpu_baseline:      # our own samples of the frame, for the margin protocol below
            flat = step().detach().float().cpu()
            tile = torch.cat([flat[:3 * R].view(R, 3), flat[3 * R:4 * R].view(R, 1), flat[4 * R:].view(R, 1)], 1).numpy()   # [R,5] = (rgb, depth, acc)
            lw = rend.last['ws']
            ours = dict(tile=tile, cs_idx=lw['cs_idx'][:nv].cpu(), cs_vid=lw['cs_vid'][:nv].cpu(), cs_tvid=lw['cs_tvid'][:nv].cpu(),
                        sample_out=lw['sample_out'][:nv].cpu())
        if world == 1 and not a.no_train and not a.no_secondary and not os.environ.get('SHERF_HIPCPU_LIB'):
            res['train'] = train_step_child(lrank)
            # the same step on the headline framing (1.27 M valid samples instead of cfg2's 0.69 M), the reference-init network
            if a.config.startswith('cfg2_dense'):
                res['train_dense'] = train_step_child(lrank, extra=['--config', a.config, '--no-pmc'])
        if world == 1 and not a.no_secondary and not a.overlap_child and not os.environ.get('SHERF_HIPCPU_LIB') and isinstance(res.get('secondary'), dict):
            # the drop-in's real entry point at real size: TriPlaneGenerator.forward with the full-size StyleGAN2 backbone and both
            # ResNet-18 passes around this renderer, per-stage HIP-event breakdown (bench_generator.py, a child process)
            gf = generator_forward_child(a, lrank)
            if gf.get('value'):
                gf['rays_per_s_vs_renderer_alone'] = gf['value'] / res['value']
            res['secondary']['generator_forward'] = gf
        if not a.no_cpu_baseline and world == 1:            # reported at N = 1 only (rank 0's host cores)
            res['cpu_baseline'] = cpu_baseline(a.config)
        if world == 1 and not a.no_pmc and 'roofline' in res:
            a.precision_used = used
            del w
            torch.cuda.empty_cache()
            form_used = res['config'].get('mlp_form')
            if form_used in MLP_FORM_KERNEL and not res['config'].get('mlp_split'):
                os.environ['SHERF_BENCH_MLP_FORM'] = form_used                   # (inherited by the counter passes' child)
            pm = pmc_traffic(a, lrank, keys=(MLP_FORM_KERNEL[form_used] + '<',) if form_used in MLP_FORM_KERNEL and used != 'f16x3' else None)
            res['roofline']['traffic'] = pm.get('hbm_bytes_per_launch')
            res['roofline']['traffic_detail'] = pm
        if ours is not None:
            import tempfile
            path = os.path.join(tempfile.mkdtemp(prefix='sherf_bench_'), 'oracle_frame.npz')
            res['torch_gpu_baseline'] = torch_gpu_baseline_child(a, lrank, save=path)
            if res['torch_gpu_baseline'].get('value'):
                res['torch_gpu_baseline']['speedup_vs_it'] = res['value'] / res['torch_gpu_baseline']['value']
            if os.path.exists(path):             # BASELINE's "PSNR vs ref": the timed frame against the oracle's whole frame
                try:
                    from synthdata import fixtures as _fx
                    res['parity'] = frame_parity(ours, np.load(path), S, plain=_fx.variant_of(a.config) == 'ri')
                except Exception as ex:
                    res['parity'] = dict(error=f'{type(ex).__name__}: {str(ex)[:300]}', ok=False)
            else:
                res['parity'] = dict(error='the oracle child wrote no frame: ' + str(res['torch_gpu_baseline'].get('error')), ok=False)
            res['parity_ok'] = bool(res['parity'].get('ok'))
        # the keys a reader compares across rounds first (the driver's tail cuts long lines)
        first = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'latency_ms_per_frame', 'value_frames_overlapped', 'ms_per_step_frames_overlapped',
                 'latency_ms_per_frame_with_frames_overlapping', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'parity_ok', 'roofline', 'cpu_baseline')
        res = {**{k: res[k] for k in first if k in res}, **{k: v for k, v in res.items() if k not in first}}
        print(json.dumps(res))
        if res.get('parity_ok') is False:
            sys.stdout.flush()
            print('bench.py: PARITY FAILED (see "parity" in the JSON line)', file=sys.stderr)
            parity_failed = True
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if parity_failed:
        sys.exit(3)


def _oracle_state(cfg_name, dev=None):
    """name -> tensor of every renderer / decoder parameter of the configuration's variant (checker side: the oracle's weights)."""
    from oracle import fixtures
    import json as _json
    shapes = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'param_shapes.json')))
    variant = fixtures.variant_of(cfg_name)
    vals = {n: fixtures.param_value(variant, n, s, shapes) for n, s in shapes.items()}
    return {n: (torch.from_numpy(v) if dev is None else torch.from_numpy(v).to(dev)) for n, v in vals.items() if v is not None}


def cpu_baseline(cfg_name, stride=61):
    """The oracle (CPU port of the reference algorithm, pinned to the unmodified reference by the golden vectors; brute-force K-NN
    included) timed on the host cores, on a bounded UNBIASED sample of the same workload: every `stride`-th ray of the whole frame
    (61 is coprime to the image width, so the subset covers every column and row band: the valid-sample fraction of the sample is
    the frame's, unlike round 3's centred crop which sat on the body)."""
    from oracle import fixtures, sherf_oracle as O
    state = _oracle_state(cfg_name)
    fx = fixtures.renderer_inputs(cfg_name)
    c = fx['cfg']
    H, W = c['H'], c['W']
    sel = np.arange(0, H * W, stride if H * W >= 64 * stride else 1)
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
                kind_note='oracle/sherf_oracle.py: the CPU restatement of the reference, pinned to the unmodified reference by tests/golden (tests/test_oracle_golden.py)',
                sample=f'every {stride}th ray of the whole {H}x{W}x{c["S"]} frame ({len(sel)} rays, {int(r["mask"].sum())} valid samples = '
                       f'{int(r["mask"].sum()) / (len(sel) * c["S"]):.1%} of the sample: the frame\'s own fraction), fp32 torch-CPU, {dt:.1f} s')


def frame_parity(ours, ref, S, plain=False):
    """The timed frame against the oracle's whole 512x512x64 frame (stock ATen ops on the GPU: fp32 = the reference, float64 = the
    truth on the reference's branches), by oracle/parity.py (SURVEY section 7 hard part 1): every sample the two take a different
    branch on is listed with the ORACLE's decision margin; per-sample errors are TRUE relative errors (|d| / max(|ref|, floor)); a ray
    over the image tolerance must contain a flipped / in-margin sample.  `ok` = the verdict:
      every workload: quantile_p(|ours - fp64|) <= quantile_p(|ref32 - fp64|) + 1e-3 at p50 / p99 / p99.9 / max, sigma+ and rgb;
      plain=True (the reference-init network): in addition every sample off the margins within 1e-3 of the fp32 reference outright.
    ours: tile [R,5] = (rgb, depth, acc) + the compact samples."""
    from oracle import parity
    o = {k: torch.from_numpy(ref[k]) for k in ref.files}
    truth = dict(sample_sigma=o.pop('truth_sigma'), sample_rgb=o.pop('truth_rgb'))
    rep, touched = parity.truth_protocol(o, truth, ours['cs_idx'], ours['cs_vid'], ours['cs_tvid'], ours['sample_out'], S)
    img = parity.image_protocol(ours['tile'][:, :3], ours['tile'][:, 4], o['rgb'], o['acc'], touched)
    flips_ok = max(rep['mask_flip_max_margin'], rep['vertex_flip_max_gap'], rep['t_vertex_flip_max_gap']) < parity.EPS
    img_ok = (img['rays_over_tolerance_unexplained'] == 0 and img['rgb_err_max_clean'] < 1e-3 and img['acc_err_max_clean'] < 1e-3
              and img['dpsnr_vs_target_db'] <= 0.05)
    out = dict(protocol='oracle/parity.py: flips listed with the oracle\'s margin; per-sample true relative error of ours and of the fp32 reference '
                        'against the float64 truth on the same branches, quantile by quantile; rays over tolerance must be explained',
               truth=rep, image=img, tolerance=1e-3, flips_ok=bool(flips_ok), image_ok=bool(img_ok), truth_ok=bool(rep['ok']),
               table=parity.format_truth_table(rep),
               oracle='oracle/sherf_oracle.py (pinned to the unmodified reference) as stock ATen ops on the GPU, whole frame: fp32 + float64 truth')
    ok = flips_ok and img_ok and rep['ok']
    # the plain numbers against the fp32 reference are ALWAYS in the report (ours vs the reference, maximum over every sample off the
    # margins); they decide the verdict on the reference-init workloads, where 1e-3 is required outright
    srep, _ = parity.sample_protocol(o, ours['cs_idx'], ours['cs_vid'], ours['cs_tvid'], ours['sample_out'], S)
    out['samples'] = srep
    out['plain_within_1e-3_of_fp32_reference'] = bool(srep['sigma_rel_max'] <= 1e-3 and srep['rgb_rel_max'] <= 1e-3)
    if plain:
        out['plain_ok'] = out['plain_within_1e-3_of_fp32_reference']
        ok = ok and out['plain_ok']
    out['ok'] = bool(ok)
    return out


def bench_child(a, lrank, extra, timeout=400):
    """This file again in a child process with other flags (same box, same environment) -> its JSON line.  No baselines, no PMC passes, no
    training step in the child."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['LOCAL_RANK'] = str(lrank)
    base = ['--config', a.config, '--precision', a.precision, '--bn-mode', a.bn_mode, '--no-cpu-baseline', '--no-torch-gpu-baseline', '--no-pmc', '--no-train']
    for flag, val in (('--table-precision', a.table_precision), ('--encoder-precision', a.encoder_precision)):
        if val:
            base += [flag, val]
    args, seen = [], set()
    for tok in extra:                                   # flags in `extra` override the ones inherited from this run
        args.append(tok)
        if tok.startswith('--'):
            seen.add(tok)
    i, inherited = 0, []
    while i < len(base):
        takes_value = i + 1 < len(base) and not base[i + 1].startswith('--')
        if base[i] not in seen:
            inherited += base[i:i + (2 if takes_value else 1)]
        i += 2 if takes_value else 1
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + inherited + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=timeout, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if not line:
            return dict(error=f'child rc={r.returncode}: {r.stderr.strip()[-300:]}')
        return json.loads(line[-1])
    except Exception as ex:
        return dict(error=f'{type(ex).__name__}: {str(ex)[:300]}')


def train_step_child(lrank, timeout=480, extra=()):
    """BASELINE config 5 beside the render metric: bench_train.py (one view, forward + backward through the HIP kernels + the reference's
    flat-gradient exchange + Adam, training_loop.py:354-386) for a few steps in a child process -> its JSON line (ms per step, phases,
    the roofline of its dominant kernel), so that the driver's bench run times the training step too."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['LOCAL_RANK'] = str(lrank)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench_train.py'), '--steps', '4', '--warmup', '2'] + list(extra), env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=timeout, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if not line:
            return dict(error=f'child rc={r.returncode}: {r.stderr.strip()[-300:]}')
        d = json.loads(line[-1])
        return {k: d.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'phases_ms', 'host_ms', 'roofline', 'dtype', 'config', 'final_loss')}
    except Exception as ex:
        return dict(error=f'{type(ex).__name__}: {str(ex)[:300]}')


def generator_forward_child(a, lrank, timeout=300):
    """bench_generator.py in a child process -> its JSON line (TriPlaneGenerator.forward at full size: rays/s, stage table, cached-backbone variant)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['LOCAL_RANK'] = str(lrank)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench_generator.py'), '--config', a.config, '--precision', a.precision, '--steps', '20', '--warmup', '5'],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if not line:
            return dict(error=f'child rc={r.returncode}: {r.stderr.strip()[-300:]}')
        return json.loads(line[-1])
    except Exception as ex:
        return dict(error=f'{type(ex).__name__}: {str(ex)[:300]}')


def torch_gpu_baseline_child(a, lrank, timeout=300, save=None):
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


def torch_gpu_baseline(cfg_name, dev, training, iters=3, save=None):
    """The same oracle as cpu_baseline, run through PyTorch-ROCm's stock kernels on the GPU over the WHOLE frame (checker code
    timed as a baseline, never on the product path).  Its K-NN is the blocked brute force of oracle.nearest_vertex."""
    try:
        from oracle import fixtures, sherf_oracle as O
        state = _oracle_state(cfg_name, dev)
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
        if save:                                # the oracle's image AND samples of the bench frame + its decision margins (one more,
            fx['options'] = dict(fx['options'], margins=True)          # untimed pass): what the parent's `parity` protocol compares with
            with torch.no_grad():
                r = O.render_from_fixture(fx, state, training=training, keep=False, device=dev)
                t64 = O.truth64_from_fixture(fx, state, r, training=training, device=dev)      # float64, same branches
            g = lambda k: r[k].detach().cpu().numpy()
            np.savez(save, rgb=g('rgb'), acc=g('acc'), mask=g('mask'), valid=g('valid'), d2_all=g('d2_all'), vert_id=g('vert_id'),
                     t_vert_id=g('t_vert_id'), vert_gap=g('vert_gap'), t_vert_gap=g('t_vert_gap'), sample_rgb=g('sample_rgb'),
                     sample_sigma=g('sample_sigma'), truth_sigma=t64['sample_sigma'].detach().cpu().numpy(),
                     truth_rgb=t64['sample_rgb'].detach().cpu().numpy())
        return dict(value=R / dt, unit='rays/s', kind='port', seconds_per_frame=dt,
                    sample=f'whole {c["H"]}x{c["W"]}x{c["S"]} frame ({int(r["mask"].sum())} valid samples), oracle/sherf_oracle.py as stock '
                           f'PyTorch-ROCm fp32 ops on the GPU, best of {iters} after 1 warm-up')
    except Exception as e:                      # a baseline must never take the bench line down with it
        return dict(error=f'{type(e).__name__}: {e}'[:300])


if __name__ == '__main__':
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        if int(os.environ.get('WORLD_SIZE', 1)) > 1:        # do not linger in (or before) a collective the other ranks are waiting in
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)
        raise
