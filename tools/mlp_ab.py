"""A/B timings + a bitwise stress test of the per-sample network's launch forms (GPU box only):

    bash tools/build_variants.sh tw3 noperm && gpurun -- 'python tools/mlp_ab.py > gpurun_out/mlp_ab.log'

Renders the bench frame once with the product library, then on the frame's own tokens
  * times, interleaved over several rounds after a clock warm-up, for every libsherf_hip_<tag>.so beside the product library:
    the one-launch kernel (sherf_nerf_mlp) and the two-launch form (sherf_nerf_mlp_split) (the two kernels' own durations: the
    rocprofv3 kernel trace of bench.py);
  * checks every output word of every form / variant against the product's one-launch result;
  * --stress N: N launches of each form of the PRODUCT library while a second stream keeps the chip unevenly busy with random gathers
    (VMEM-heavy, different sizes), every output word compared each time -- the hardware-only failure of round 2's fused gather -> MLP
    experiment showed as ~12 % of the tiles differing from launch to launch; this is the test that would have caught it."""
import argparse
import ctypes as ct
import glob
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--config', default='cfg2_ri')
    ap.add_argument('--precision', default='f16', choices=['f16x3', 'f16', 'bf16'])
    ap.add_argument('--stress', type=int, default=0)
    ap.add_argument('--forms', default='one,two', help='launch forms to time: one (sherf_nerf_mlp), two (sherf_nerf_mlp_split), tt (sherf_nerf_mlp2: two tiles per wave), pp (sherf_nerf_mlp3: epilogues inside the MFMA stream), pe (sherf_nerf_mlp3_pe: pp with the encodings read as fragments the gather wrote); pipe = the round-4 pipelined experiment, if the library has it')
    ap.add_argument('--sustain', type=float, default=0.0, help='after the timings: launch the LAST form back to back for this many seconds (power / clock telemetry: tools/power_probe.py)')
    ap.add_argument('--zero', default='', help="power probe: 'tokens' = zero tokens / extras, 'all' = zero weights too (same instruction stream, less switching; outputs are not compared)")
    ap.add_argument('--only', default='', help='comma list of library tags to run (default: the product library and every libsherf_hip_<tag>.so beside it)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'mlp_ab.json'))
    a = ap.parse_args()
    import bench
    from sherf_amd import _lib
    from sherf_amd.renderer import MLP_PRECISIONS
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    P = MLP_PRECISIONS[a.precision]
    # form `pe` (round 6: sherf_nerf_mlp3_pe) needs the frame's encodings as fragments: the frame is then rendered in the configuration that writes them
    want_pe = 'pe' in a.forms.split(',')
    w = bench.make_workload(argparse.Namespace(config=a.config, precision='f16' if want_pe else 'f16x3', bn_mode='train'), 0.4, dev)
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
    wc = {k: v for k, v in rend._weights(dec, dev, a.precision).items() if k in ('stream', 'wbias')}
    out = torch.empty(tiles * 32, 4, device=dev)
    zfrag = torch.empty((tiles + 8) * 2048, dtype=torch.int32, device=dev)
    counters = ws['counters'].clone()
    counters[3] = 0

    def bind(path):
        lib = ct.CDLL(path)
        one = lib.sherf_nerf_mlp
        one.restype, one.argtypes = ct.c_int, [ct.c_void_p] * 5 + [ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p]
        two = getattr(lib, 'sherf_nerf_mlp_split', None)
        if two is not None:
            two.restype, two.argtypes = ct.c_int, [ct.c_void_p] * 5 + [ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p, ct.c_void_p]
        pipe = getattr(lib, 'sherf_nerf_mlp_pipe', None)
        if pipe is not None:
            pipe.restype, pipe.argtypes = one.restype, one.argtypes
        tt = getattr(lib, 'sherf_nerf_mlp2', None)
        if tt is not None:
            tt.restype, tt.argtypes = one.restype, one.argtypes
        pp = getattr(lib, 'sherf_nerf_mlp3', None)
        if pp is not None:
            pp.restype, pp.argtypes = one.restype, one.argtypes
        pe = getattr(lib, 'sherf_nerf_mlp3_pe', None)
        if pe is not None:
            pe.restype, pe.argtypes = ct.c_int, [ct.c_void_p] * 6 + [ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p]
        return one, two, pipe, tt, pp, pe

    def launch(fn, form):
        if form == 'pe':
            return fn[5](A(counters), A(ws['tokens']), A(ws['extras']), A(ws['pefrag']), A(wc['stream']), A(wc['wbias']), P, capx, A(out), stream)
        if form in ('one', 'pipe', 'tt', 'pp'):
            return fn[{'one': 0, 'pipe': 2, 'tt': 3, 'pp': 4}[form]](A(counters), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), P, capx, A(out), stream)
        return fn[1](A(counters), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), P, capx, A(zfrag), A(out), stream)

    def timed(fn, form, iters=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            rc = launch(fn, form)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        return e0.elapsed_time(e1) / iters

    if a.zero:
        ws = dict(ws)
        ws['tokens'] = torch.zeros_like(ws['tokens']); ws['extras'] = torch.zeros_like(ws['extras'])
        if ws.get('pefrag') is not None:
            ws['pefrag'] = torch.zeros_like(ws['pefrag'])
        if a.zero == 'all':
            wc = {k: torch.zeros_like(v) for k, v in wc.items()}
    libs = {'product': os.path.join(ROOT, 'sherf_amd', 'libsherf_hip.so')}
    for path in sorted(glob.glob(os.path.join(ROOT, 'sherf_amd', 'libsherf_hip_*.so'))):
        tag = os.path.basename(path)[len('libsherf_hip_'):-3]
        if tag not in ('bwd', 'ops', 'trace') and not tag.startswith('nn') and (not a.only or tag in a.only.split(',')):
            libs[tag] = path
    bound = {t: bind(p) for t, p in libs.items()}
    launch(bound['product'], 'one'); torch.cuda.synchronize()
    ref = out[:nv].clone()
    forms = [f for f in a.forms.split(',') if f]
    arms = [(t, form) for t, fn in bound.items() for form in forms
            if form == 'one' or (form == 'two' and fn[1] is not None) or (form == 'pipe' and fn[2] is not None and a.precision != 'f16x3')
            or (form == 'tt' and fn[3] is not None and a.precision != 'f16x3') or (form == 'pp' and fn[4] is not None and a.precision != 'f16x3')
            or (form == 'pe' and fn[5] is not None and a.precision == 'f16' and ws.get('pefrag') is not None)]
    for _ in range(40):                                         # clock warm-up
        launch(bound['product'], 'one')
    torch.cuda.synchronize()
    times = {arm: [] for arm in arms}
    for _ in range(a.rounds):
        for arm in arms:
            for _ in range(5):
                launch(bound[arm[0]], arm[1])
            times[arm].append(timed(bound[arm[0]], arm[1]))
    flop = nv * bench.FLOP_PER_VALID_SAMPLE
    report = dict(valid_samples=nv, tiles=tiles, precision=a.precision, arms={})
    for arm in arms:
        out.fill_(float('nan')); launch(bound[arm[0]], arm[1]); torch.cuda.synchronize()
        diff = float((out[:nv] - ref).abs().max())
        ms = min(times[arm])
        report['arms']['%s/%s' % arm] = dict(ms=times[arm], ms_min=ms, frac_of_peak=flop / (ms * 1e-3) / 1e12 / bench.PEAK_BF16_TFLOPS, max_abs_diff_vs_product=diff,
                                            ticket_word=int(counters[3]))
        print(f'[arm] {arm[0]:10s} {arm[1]:3s} ms {" ".join(f"{x:.3f}" for x in times[arm])}  frac {flop / (ms * 1e-3) / 1e12 / bench.PEAK_BF16_TFLOPS:.3f}  '
              f'|diff| vs product/one {diff:.2e}  counters[3] {int(counters[3])}')

    if a.sustain > 0:
        import time
        arm = arms[-1]
        t0 = time.perf_counter(); n = 0
        print(f'[sustain] start {time.time():.3f}', flush=True)
        while time.perf_counter() - t0 < a.sustain:
            for _ in range(200):
                launch(bound[arm[0]], arm[1])
            torch.cuda.synchronize(); n += 200
        dt = time.perf_counter() - t0
        print(f'[sustain] end {time.time():.3f}: {arm[0]}/{arm[1]} x {n} launches in {dt:.2f} s = {1e3 * dt / n:.4f} ms per launch (back to back, incl. launch gaps)', flush=True)
    if a.stress:
        torch.manual_seed(0)
        side = torch.cuda.Stream(dev)
        big = torch.randn(64 << 20, device=dev)
        idxs = [torch.randint(0, big.numel(), (n,), device=dev) for n in (1 << 18, 1 << 21, 1 << 23, 3 << 20)]
        bad = {}
        for form in forms:
            if form in ('pipe', 'tt', 'pp', 'pe') and a.precision == 'f16x3':
                continue
            n_bad_launches, n_bad_words = 0, 0
            for it in range(a.stress):
                with torch.cuda.stream(side):                       # uneven co-resident load: a few random gathers of different sizes
                    for k in range(1 + it % 3):
                        _ = big[idxs[(it + k) % len(idxs)]].sum()
                out.fill_(float('nan'))
                assert launch(bound['product'], form) == 0
                d = (out[:nv] != ref).any(1)
                nb = int(d.sum())
                if nb:
                    n_bad_launches += 1; n_bad_words += nb
            torch.cuda.synchronize()
            bad[form] = dict(launches=a.stress, launches_with_a_difference=n_bad_launches, differing_samples=n_bad_words)
            print(f'[stress] {form}: {a.stress} launches under side-stream load, {n_bad_launches} with a differing word ({n_bad_words} samples in all)')
        report['stress'] = bad
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
