"""In-kernel timeline + variant timings of `nerf_mlp_kernel` (profiling aid, GPU box only):

    bash tools/build_variants.sh && gpurun -- 'python tools/mlp_trace.py > gpurun_out/mlp_trace.log'

Renders the bench frame (cfg2) once with the product library, then on the frame's own tokens
  * launches `sherf_nerf_mlp` from the -DSHERF_MLP_TRACE=1 build: every wave stamps s_memtime at the end of each step's MFMA stream
    (0), after the weight-DMA wait (1) and after the workgroup barrier (2) -> per step class, the mean cycles a wave spends
    computing, parked on vmcnt and parked at the barrier -- the split SQ_WAIT_ANY cannot give;
  * times every libsherf_hip_<tag>.so variant found beside the product library, interleaved over several rounds after a long
    warm-up (the first launches after an idle gap run at a lower clock: round 2 call B), same buffers, one process."""
import argparse
import ctypes as ct
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = dict(transformer=(0, 2), first=(2, 4), trunk1_4=(4, 20), skip=(20, 26), trunk6_7=(26, 34), heads=(34, 39), views=(39, 42))
N_STEPS = 43


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--every', type=int, default=61)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--config', default='cfg2_ri')
    ap.add_argument('--precision', default='f16', choices=['f16x3', 'f16', 'bf16'], help='the precision the variants / the trace run in')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'mlp_trace.json'))
    a = ap.parse_args()
    import bench
    from sherf_amd import _lib
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    ns = argparse.Namespace(config=a.config, precision='f16x3', bn_mode='train')
    from sherf_amd.renderer import MLP_PRECISIONS
    P = MLP_PRECISIONS[a.precision]
    w = bench.make_workload(ns, 0.4, dev)
    for _ in range(2):
        bench.render_frame(w)
    torch.cuda.synchronize()
    rend, dec = w['rend'], w['dec']
    ws = rend.last['ws']
    nv = int(ws['counters'][0])
    tiles = (nv + 31) // 32
    A = _lib.addr
    capx = (nv + 255) // 256 * 256
    stream = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    wcs = {p: {k: v for k, v in rend._weights(dec, dev, p).items() if k in ('stream', 'wbias')} for p in ('f16x3', 'bf16', 'f16')}
    out = torch.empty(tiles * 32, 4, device=dev)

    def bind(path):
        lib = ct.CDLL(path)
        f = lib.sherf_nerf_mlp
        f.restype = ct.c_int
        f.argtypes = [ct.c_void_p] * 5 + [ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p]
        return lib, f

    def launch(f, prec=None):
        prec = P if prec is None else prec
        wc = wcs[{0: 'bf16', 1: 'f16x3', 2: 'f16'}[prec]]
        return f(A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), prec, capx, A(out), stream)

    def timed(f, prec=None, iters=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            rc = launch(f, prec)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        return e0.elapsed_time(e1) / iters

    libs = {'product': os.path.join(ROOT, 'sherf_amd', 'libsherf_hip.so')}
    for path in sorted(glob.glob(os.path.join(ROOT, 'sherf_amd', 'libsherf_hip_*.so'))):
        tag = os.path.basename(path)[len('libsherf_hip_'):-3]
        if tag not in ('bwd', 'ops') and not tag.startswith('nn'):        # (nn*: sampler variants, same MLP)
            libs[tag] = path
    bound = {t: bind(p) for t, p in libs.items()}
    launch(bound['product'][1]); torch.cuda.synchronize()
    ref = out[:nv].clone()
    report = dict(valid_samples=nv, tiles=tiles, libs={}, trace=None)
    for _ in range(40):                                        # clock warm-up
        launch(bound['product'][1])
    torch.cuda.synchronize()
    times = {t: [] for t in bound}
    for r in range(a.rounds):
        for t, (_, f) in bound.items():
            for _ in range(5):
                launch(f)
            times[t].append(timed(f))
    for t, (_, f) in bound.items():
        out.fill_(float('nan')); launch(f); torch.cuda.synchronize()
        got = out[:nv]
        sig = ref[:, 3].clamp(min=0)
        report['libs'][t] = dict(ms=times[t], ms_min=min(times[t]), max_abs_diff_vs_product=float((got - ref).abs().max()),
                                 sigma_rel_to_max=float((got[:, 3].clamp(min=0) - sig).abs().max() / sig.max()))
        print(f'[lib] {t:10s} ms {" ".join(f"{x:.3f}" for x in times[t])}   |diff| vs product {report["libs"][t]["max_abs_diff_vs_product"]:.2e}')
    bt = [timed(bound['product'][1], prec=0) for _ in range(a.rounds)]
    report['bf16_single_product_ms'] = bt
    print(f'[bf16 x1] ms {" ".join(f"{x:.3f}" for x in bt)}')
    flop = nv * bench.FLOP_PER_VALID_SAMPLE
    print(f'[roofline] {a.precision} {flop / (min(times["product"]) * 1e-3) / 1e12 / bench.PEAK_BF16_TFLOPS:.3f}   bf16 x1 {flop / (min(bt) * 1e-3) / 1e12 / bench.PEAK_BF16_TFLOPS:.3f} of the bf16 peak')

    if 'trace' in bound:
        lib, f = bound['trace']
        groups = (tiles + 3) // 4                                   # (the decoder launch: one workgroup per 4 tiles)
        nslot = (groups + a.every - 1) // a.every
        buf = torch.zeros(nslot * 8 * 256, dtype=torch.int32, device=dev)
        lib.sherf_mlp_set_trace.argtypes = [ct.c_void_p, ct.c_int]
        assert lib.sherf_mlp_set_trace(buf.data_ptr(), a.every) == 0
        for _ in range(5):
            launch(f)
        torch.cuda.synchronize()
        t = (buf.cpu().numpy().astype(np.int64).reshape(nslot, 8, 64, 4)[:, :4]) & 0xffffffff
        st = t[:, :, :N_STEPS, :3].copy()                           # [workgroup, wave, step, (compute end, dma landed, barrier left)]
        start, end = t[:, :, 63, 0], t[:, :, 63, 2]
        prev = np.concatenate([start[:, :, None], st[:, :, :-1, 2]], 2)
        comp = ((st[..., 0] - prev) & 0xffffffff)[:, :, :N_STEPS - 1]          # (the last step has no barrier: its stamp 0 only)
        vmw = ((st[..., 1] - st[..., 0]) & 0xffffffff)[:, :, :N_STEPS - 1]
        barw = ((st[..., 2] - st[..., 1]) & 0xffffffff)[:, :, :N_STEPS - 1]
        last = (st[:, :, N_STEPS - 1, 0] - st[:, :, N_STEPS - 2, 2]) & 0xffffffff
        total = (end - start) & 0xffffffff
        rep = dict(traced_workgroups=int(nslot), cycles_per_tile=float(total.mean()), compute_sum=float(comp.sum(2).mean() + last.mean()),
                   vmcnt_wait_sum=float(vmw.sum(2).mean()), barrier_wait_sum=float(barw.sum(2).mean()),
                   per_step=dict(compute=[float(x) for x in comp.mean((0, 1))], vmcnt=[float(x) for x in vmw.mean((0, 1))],
                                 barrier=[float(x) for x in barw.mean((0, 1))], last_step_compute=float(last.mean())),
                   classes={k: dict(steps=v[1] - v[0], compute=float(comp[:, :, v[0]:v[1]].mean()), vmcnt_wait=float(vmw[:, :, v[0]:v[1]].mean()),
                                    barrier_wait=float(barw[:, :, v[0]:v[1]].mean())) for k, v in CLASSES.items()},
                   hw_id_sample=[int(x) for x in t[0, :, 63, 1]])
        report['trace'] = rep
        print(f'[trace] cycles/tile {rep["cycles_per_tile"]:.0f} = compute {rep["compute_sum"]:.0f} + vmcnt {rep["vmcnt_wait_sum"]:.0f} + barrier {rep["barrier_wait_sum"]:.0f}')
        for k, v in rep['classes'].items():
            print(f'        {k:12s} x{v["steps"]:2d}: compute {v["compute"]:7.0f}  vmcnt {v["vmcnt_wait"]:6.0f}  barrier {v["barrier_wait"]:6.0f}')
        print('        per-step compute:', ' '.join(f'{x:.0f}' for x in rep['per_step']['compute']), '| last', f'{rep["per_step"]["last_step_compute"]:.0f}')
        # ---- are the two workgroups of a CU in phase?  every workgroup traced once: start / end stamps + its CU (HW_ID, XCC_ID) ----
        buf2 = torch.zeros(groups * 8 * 256, dtype=torch.int32, device=dev)
        assert lib.sherf_mlp_set_trace(buf2.data_ptr(), 1) == 0
        launch(f); torch.cuda.synchronize()
        t2 = (buf2.cpu().numpy().astype(np.int64).reshape(groups, 8, 64, 4)[:, 0]) & 0xffffffff          # wave 0 of every workgroup
        s0, hw, e0, xcc = t2[:, 63, 0], t2[:, 63, 1], t2[:, 63, 2], t2[:, 63, 3] & 0xf
        cu_key = (xcc << 16) | (hw & 0xff00)                                                              # XCC | SE, SH, CU
        T = float(np.median((e0 - s0) & 0xffffffff))
        phases, slots = [], []
        for key in np.unique(cu_key):
            idx = np.nonzero(cu_key == key)[0]
            idx = idx[np.argsort(s0[idx])]
            for a_, i in enumerate(idx):
                for j in idx[:a_]:
                    if ((s0[i] - s0[j]) & 0xffffffff) < ((e0[j] - s0[j]) & 0xffffffff):                   # j still running when i starts
                        phases.append(((s0[i] - s0[j]) & 0xffffffff) / T)
                        slots.append((int(hw[i] & 0xf), int(hw[j] & 0xf)))
        ph = np.array(phases)
        hist = np.histogram(ph, bins=10, range=(0, 1))[0]
        rep['phase'] = dict(tile_cycles=T, cus=int(len(np.unique(cu_key))), pairs=int(len(ph)), hist_start_offset_over_T=[int(x) for x in hist],
                            wave_slot_pairs_sample=slots[:8])
        print(f'[phase] {len(np.unique(cu_key))} CUs, tile {T:.0f} cycles, co-resident start offsets / T histogram (0..1 in tenths): {hist.tolist()}')
        print(f'        wave slots of co-resident pairs (new, old): {slots[:8]}')
        assert lib.sherf_mlp_set_trace(None, 0) == 0
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
