"""In-kernel timeline of `nerf_mlp_kernel` (profiling aid, GPU box only):

    python tools/mlp_trace.py [--lib sherf_amd/libsherf_hip_trace.so] [--shapes 4x1,8x1] [--out gpurun_out/mlp_trace.json]

Renders the bench frame (cfg2) once with the product library, then launches `sherf_nerf_mlp` from a build with
-DSHERF_MLP_TRACE=1 (tools/build_variants.sh) on the frame's own tokens: every wave stamps s_memtime at the end of each step's
MFMA stream (0), after the weight-DMA wait (1) and after the workgroup barrier (2).  Prints, per step class, the mean cycles a wave
spends computing, parked on vmcnt and parked at the barrier -- the split SQ_WAIT_ANY cannot give.
Also times alternative builds of the same ABI given with --time-libs (same buffers, one process)."""
import argparse
import ctypes as ct
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=os.path.join(ROOT, 'sherf_amd', 'libsherf_hip_trace.so'))
    ap.add_argument('--shapes', default='4x1,8x1')
    ap.add_argument('--every', type=int, default=61)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'mlp_trace.json'))
    ap.add_argument('--time-libs', default='', help='comma list of tag=path of alternative libraries to time (shape list applies)')
    a = ap.parse_args()
    import bench
    from sherf_amd import _lib
    from sherf_amd.renderer import MLP_SHAPES
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    ns = argparse.Namespace(config='cfg2', precision='bf16x3', bn_mode='train')
    w = bench.make_workload(ns, 0.4, dev)
    d, rend, dec = w['d'], w['rend'], w['dec']
    with torch.no_grad():
        for _ in range(2):
            rend(w['planes'], w['obs_img'], w['obs_feat'], w['sp'], None, w['sp_input'], dec, d['ray_o_all'][:, 0], d['ray_d_all'][:, 0],
                 d['near_all'][:, 0], d['far_all'][:, 0], d, w['opts'])
    torch.cuda.synchronize()
    ws, cap = rend.last['ws'], int(rend.last['cap'])
    wc = rend._weights(dec, dev)
    nv = int(ws['counters'][0])
    tiles = (nv + 31) // 32
    A = _lib.addr
    out = torch.empty(tiles * 32, 4, device=dev)
    ref = torch.empty(tiles * 32, 4, device=dev)
    stream = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.call('sherf_nerf_mlp', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 1, 0, cap, A(ref), stream)
    torch.cuda.synchronize()
    capx = (nv + 255) // 256 * 256

    def bind(path):
        lib = ct.CDLL(path)
        f = lib.sherf_nerf_mlp
        f.restype = ct.c_int
        f.argtypes = [ct.c_void_p] * 5 + [ct.c_int, ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p]
        return lib, f

    def timed(f, sid, iters=20):
        for _ in range(3):
            f(A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 1, sid, capx, A(out), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            rc = f(A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 1, sid, capx, A(out), stream)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters, rc

    report = dict(valid_samples=nv, tiles=tiles, shapes={}, libs={})
    shapes = a.shapes.split(',')
    if os.path.exists(a.lib):
        lib, f = bind(a.lib)
        for name in shapes:
            sid = MLP_SHAPES[name]
            nw = 4 if name.startswith('4x') else 8
            groups = (tiles + nw - 1) // nw
            every = a.every
            nslot = (groups + every - 1) // every
            buf = torch.zeros(nslot * 8 * 256, dtype=torch.int32, device=dev)
            lib.sherf_mlp_set_trace.argtypes = [ct.c_void_p, ct.c_int]
            assert lib.sherf_mlp_set_trace(buf.data_ptr(), every) == 0
            ms, rc = timed(f, sid, iters=5)
            assert rc == 0
            err = float((out[:nv] - ref[:nv]).abs().max())
            t = buf.cpu().numpy().astype(np.int64).reshape(nslot, 8, 64, 4)[:, :nw] & 0xffffffff
            nsteps = 49
            st = t[:, :, :nsteps, :3]                                   # [slot, wave, step, (compute end, dma landed, barrier left)]
            start, end = t[:, :, 63, 0], t[:, :, 63, 2]
            prev = np.concatenate([start[:, :, None], st[:, :, :-1, 2]], 2)
            comp = (st[..., 0] - prev) & 0xffffffff
            vmw = (st[..., 1] - st[..., 0]) & 0xffffffff
            barw = (st[..., 2] - st[..., 1]) & 0xffffffff
            total = (end - start) & 0xffffffff
            ok = total.reshape(-1) > 0
            cls = dict(prologue=slice(0, 9), first=slice(9, 13), trunk=slice(13, 29), skip=slice(29, 33), trunk2=slice(33, 41),
                       heads=slice(41, 46), views=slice(46, 48), rgb=slice(48, 49))
            rep = dict(ms=ms, max_abs_diff_vs_product=err, traced_workgroups=int(nslot), cycles_per_tile=float(total.reshape(-1)[ok].mean()),
                       compute_sum=float(comp.sum(2).mean()), vmcnt_wait_sum=float(vmw.sum(2).mean()), barrier_wait_sum=float(barw.sum(2).mean()),
                       classes={k: dict(steps=int(v.stop - v.start), compute=float(comp[:, :, v].mean()), vmcnt_wait=float(vmw[:, :, v].mean()),
                                        barrier_wait=float(barw[:, :, v].mean())) for k, v in cls.items()},
                       per_wave_compute=[float(comp[:, wv].sum(1).mean()) for wv in range(nw)],
                       per_wave_barrier=[float(barw[:, wv].sum(1).mean()) for wv in range(nw)],
                       hw_id_sample=[int(x) for x in t[0, :, 63, 1]])
            report['shapes'][name] = rep
            print(f'[trace] {name}: {ms:.3f} ms  cycles/tile {rep["cycles_per_tile"]:.0f} = compute {rep["compute_sum"]:.0f} + vmcnt {rep["vmcnt_wait_sum"]:.0f} '
                  f'+ barrier {rep["barrier_wait_sum"]:.0f}   (|diff| vs product {err:.1e})')
            for k, v in rep['classes'].items():
                print(f'        {k:9s} x{v["steps"]:2d}: compute {v["compute"]:7.0f}  vmcnt {v["vmcnt_wait"]:6.0f}  barrier {v["barrier_wait"]:6.0f}')
        assert lib.sherf_mlp_set_trace(None, 0) == 0
    for item in [x for x in a.time_libs.split(',') if x]:
        tag, path = item.split('=')
        if not os.path.exists(path):
            continue
        lib2, f2 = bind(path)
        report['libs'][tag] = {}
        for name in shapes:
            out.fill_(float('nan'))
            ms, rc = timed(f2, MLP_SHAPES[name])
            err = float((out[:nv] - ref[:nv]).abs().max()) if rc == 0 else float('nan')
            sig = ref[:nv, 3].clamp(min=0)
            e_sig = float(((out[:nv, 3].clamp(min=0) - sig).abs().max() / sig.max())) if rc == 0 else float('nan')
            report['libs'][tag][name] = dict(ms=ms, rc=rc, max_abs_diff_vs_product=err, sigma_rel_to_max=e_sig)
            print(f'[lib] {tag:10s} {name:10s} {ms:.3f} ms  rc={rc}  |diff| vs product {err:.2e}  sigma+ rel-to-max {e_sig:.2e}')
    # ablations of the product library (results are garbage, timing only): 64 = no workgroup barriers, 32 = no weight DMA
    plib = _lib.lib()
    fprod = plib.sherf_nerf_mlp
    report['ablate'] = {}
    for tag, dbg in (('none', 0), ('no_barrier', 64), ('no_dma', 32), ('no_dma_no_barrier', 96)):
        plib.sherf_set_debug(dbg)
        for name in shapes:
            sid = MLP_SHAPES[name]
            for _ in range(3):
                _lib.call('sherf_nerf_mlp', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 1, sid, capx, A(out), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _lib.call('sherf_nerf_mlp', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 1, sid, capx, A(out), stream)
            e1.record(); torch.cuda.synchronize()
            report['ablate'].setdefault(tag, {})[name] = e0.elapsed_time(e1) / 20
            print(f'[ablate] {tag:18s} {name:10s} {e0.elapsed_time(e1) / 20:.3f} ms')
    plib.sherf_set_debug(0)
    # plain bf16 (one product) in the default shape: the north_star's nominal precision, for the secondary line
    for _ in range(3):
        _lib.call('sherf_nerf_mlp', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 0, 0, capx, A(out), stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.call('sherf_nerf_mlp', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), 0, 0, capx, A(out), stream)
    e1.record(); torch.cuda.synchronize()
    report['bf16x1_8x1_ms'] = e0.elapsed_time(e1) / 20
    print(f'[bf16 x1, 8x1] {report["bf16x1_8x1_ms"]:.3f} ms')
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
