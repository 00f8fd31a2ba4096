"""In-kernel timeline of the two-tile network kernel `nerf_mlp2_kernel` (profiling aid, GPU box only; -DSHERF_MLP_TRACE=1 build):

    bash tools/build_variants.sh trace && gpurun -- 'python tools/mlp2_trace.py > gpurun_out/mlp2_trace.log'

Every wave keeps s_memtime stamps in lanes of three registers (the kernel's LDS is full): 0 start, 1 ring prologue done, 2 / 3 the
transformer of tile 0 / 1 done, 4 decoder entered, then per decoder step s = 2..42: 8 + 3 (s - 2) + {0 MFMA stream issued, 1 weight DMA landed,
2 barrier left}; 191 end, 190 HW_ID, 189 XCC_ID.  Reports the mean cycles per phase and, from one launch with EVERY workgroup traced, how the
two co-resident workgroups of a CU sit relative to each other (start offsets over the workgroup's duration)."""
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
    ap.add_argument('--every', type=int, default=37)
    ap.add_argument('--config', default='cfg2_dense_ri')
    ap.add_argument('--lib', default=os.path.join(ROOT, 'sherf_amd', 'libsherf_hip_trace.so'))
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'mlp2_trace.json'))
    a = ap.parse_args()
    import bench
    from sherf_amd import _lib
    from sherf_amd.renderer import MLP_PRECISIONS
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    w = bench.make_workload(argparse.Namespace(config=a.config, precision='f16x3', bn_mode='train'), 0.4, dev)
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
    wc = {k: v for k, v in rend._weights(dec, dev, 'f16').items() if k in ('stream', 'wbias')}
    out = torch.empty(tiles * 32, 4, device=dev)
    lib = ct.CDLL(a.lib)
    f = lib.sherf_nerf_mlp2
    f.restype, f.argtypes = ct.c_int, [ct.c_void_p] * 5 + [ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p]
    lib.sherf_mlp_set_trace.argtypes = [ct.c_void_p, ct.c_int]
    launch = lambda: f(A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), MLP_PRECISIONS['f16'], capx, A(out), stream)
    groups = (tiles + 7) // 8

    def traced(every):
        nslot = (groups + every - 1) // every
        buf = torch.zeros(nslot * 4 * 192, dtype=torch.int32, device=dev)
        assert lib.sherf_mlp_set_trace(buf.data_ptr(), every) == 0
        for _ in range(20):
            assert launch() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            launch()
        e1.record(); torch.cuda.synchronize()
        return (buf.cpu().numpy().astype(np.int64).reshape(nslot, 4, 192)) & 0xffffffff, e0.elapsed_time(e1) / 10

    t, ms = traced(a.every)
    d = lambda x, y: (x - y) & 0xffffffff
    st = t[:, :, 8:8 + 3 * 41].reshape(t.shape[0], 4, 41, 3)          # [wg, wave, step 2..42, (mfma issued, dma landed, barrier left)]
    total = d(t[:, :, 191], t[:, :, 0])
    prologue = d(t[:, :, 1], t[:, :, 0])
    tr0, tr1 = d(t[:, :, 2], t[:, :, 1]), d(t[:, :, 3], t[:, :, 2])
    recycle = d(t[:, :, 4], t[:, :, 3])
    prev = np.concatenate([t[:, :, 4][:, :, None], st[:, :, :-1, 2]], 2)
    comp = d(st[..., 0], prev)                                         # per step: from the previous barrier to the end of the step's MFMA issue
    vmw, barw = d(st[:, :, :40, 1], st[:, :, :40, 0]), d(st[:, :, :40, 2], st[:, :, :40, 1])
    tail = d(t[:, :, 191], st[:, :, 40, 0])
    rep = dict(config=a.config, valid_samples=nv, tiles=tiles, kernel_ms=ms, traced_workgroups=int(t.shape[0]), cycles_per_workgroup=float(total.mean()),
               shader_ghz=None, prologue=float(prologue.mean()), transformer_tile0=float(tr0.mean()), transformer_tile1=float(tr1.mean()),
               recycle_barrier=float(recycle.mean()), decoder_compute=float(comp.sum(2).mean()), decoder_vmcnt_wait=float(vmw.sum(2).mean()),
               decoder_barrier_wait=float(barw.sum(2).mean()), tail=float(tail.mean()),
               per_step_compute=[float(x) for x in comp.mean((0, 1))], per_step_barrier=[float(x) for x in barw.mean((0, 1))])
    print(f'[trace] {a.config}: {nv} samples, kernel {ms:.3f} ms (trace build); cycles per workgroup (8 tiles) {rep["cycles_per_workgroup"]:.0f} = prologue {rep["prologue"]:.0f} '
          f'+ T(tile 0) {rep["transformer_tile0"]:.0f} + T(tile 1) {rep["transformer_tile1"]:.0f} + recycle {rep["recycle_barrier"]:.0f} + decoder [compute {rep["decoder_compute"]:.0f} '
          f'+ vmcnt {rep["decoder_vmcnt_wait"]:.0f} + barrier {rep["decoder_barrier_wait"]:.0f}] + tail {rep["tail"]:.0f}')
    print('        per-step compute (steps 2..42):', ' '.join(f'{x:.0f}' for x in rep['per_step_compute']))
    print('        per-step barrier wait         :', ' '.join(f'{x:.0f}' for x in rep['per_step_barrier']))
    # ---- every workgroup traced: residency and phase of the co-resident workgroups ----
    t2, ms2 = traced(1)
    w0 = t2[:, 0]
    s0, e0, hw, xcc = w0[:, 0], w0[:, 191], w0[:, 190], w0[:, 189] & 0xf
    cu_key = (xcc << 16) | (hw & 0xff00)
    T = float(np.median(d(e0, s0)))
    span = float(d(e0.max(), s0.min())) if (e0.max() - s0.min()) < 2 ** 31 else float('nan')
    rep['shader_ghz'] = span / (ms2 * 1e-3) / 1e9 if span == span else None
    phases, simul = [], []
    for key in np.unique(cu_key):
        idx = np.nonzero(cu_key == key)[0]
        idx = idx[np.argsort(s0[idx])]
        for k_, i in enumerate(idx):
            n_live = 0
            for j in idx[:k_]:
                if d(s0[i], s0[j]) < d(e0[j], s0[j]):
                    phases.append(d(s0[i], s0[j]) / T)
                    n_live += 1
            simul.append(n_live)
    hist = np.histogram(np.array(phases), bins=10, range=(0, 1))[0]
    rep['phase'] = dict(workgroup_cycles_median=T, kernel_span_cycles=span, cus=int(len(np.unique(cu_key))), hist_start_offset_over_T=[int(x) for x in hist],
                        live_at_start_hist=[int(x) for x in np.bincount(np.array(simul), minlength=4)[:4]])
    print(f'[phase] {len(np.unique(cu_key))} CUs, workgroup {T:.0f} cycles, kernel span {span:.0f} cycles in {ms2:.3f} ms -> {rep["shader_ghz"]} GHz; '
          f'start offsets of co-resident workgroups / T (tenths): {hist.tolist()}; workgroups already live on the CU at a start: {rep["phase"]["live_at_start_hist"]}')
    assert lib.sherf_mlp_set_trace(None, 0) == 0
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report_clean(rep), open(a.out, 'w'), indent=1)


def report_clean(r):
    return json.loads(json.dumps(r, default=float))


if __name__ == '__main__':
    main()
