"""Per-device selection of the launch shape of `sherf_nerf_mlp` (the MFMA kernel that bounds the frame) and of the
schedule variant of `sherf_gather_tokens` (`tune_gather`, same protocol).

The kernel exists in several shapes behind one C entry point (`shape` argument, include/sherf_hip.h): the same arithmetic in a
different workgroup decomposition ('4x2', '8x1split', '8x1split2', '8x1persist') or merely a different instruction schedule
('8x1il8', '8x1prio', '8x1prio_il8').  Which one is fastest depends on how the hardware's issue logic overlaps the VALU epilogues
with the MFMA chains -- something only a measurement on the device answers -- so instead of baking one in, `tune_mlp` times every
shape on the samples of the frame the renderer has just rendered and verifies each against the default shape's output ON THE
DEVICE before it may be chosen:

    rend(...)                                    # one frame, fills the renderer's workspace
    report = tune.tune_mlp(rend, decoder)        # {'best': '8x1prio', 'shapes': {name: {'ms', 'max_abs_diff', 'ok'}}}
    rend.mlp_shape = report['best']

A shape is eligible only if its (r, g, b, sigma) equal the default's to `tol` (default 0: bit-identical; every shape feeds the
same operands through the same MFMA chain in the same order, so anything else is a bug, not rounding).  There is no CPU or torch
path here: the function raises off-GPU like everything else in the package.  `bench.py` runs it in a child process (a candidate
that faults must not take the measurement down with it) and falls back to '8x1' on any failure.
"""
import ctypes as _ct

import torch

from . import _lib
from .renderer import MLP_SHAPES

DEFAULT = '8x1'
# shapes that use `tokens` as scratch (z_0 / z_1 fragments overwrite the head of every tile): the input must be restored after them
_CLOBBERS_TOKENS = ('8x1split', '8x1split2')


def _time_launches(fn, iters, dev):
    """Average duration in ms of `iters` back-to-back calls of `fn`, from HIP events on the stream `fn` launches on."""
    st = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


def tune_mlp(renderer, decoder, candidates=None, iters=20, warmup=3, tol=0.0, exact_capacity=False):
    """Times `sherf_nerf_mlp` in every candidate shape on the renderer's last frame.  Returns a report dict; never changes the
    renderer (the caller assigns `renderer.mlp_shape`).  Requires a preceding forward on the GPU with the bf16x3 MLP.
    exact_capacity: launch for the frame's own sample count (what SHERF_FRAME_EXACT_GRIDS does) instead of the buffers' capacity --
    the difference between the two timings of a shape is what its grid of empty workgroups costs."""
    last = getattr(renderer, 'last', None)
    if not last:
        raise RuntimeError('tune_mlp needs a rendered frame: call the renderer once first')
    ws, cap = last['ws'], int(last['cap'])
    dev = ws['tokens'].device
    wc = renderer._weights(decoder, dev)
    prec = {'bf16': 0, 'bf16x3': 1}[renderer.mlp_precision]
    names = list(candidates) if candidates is not None else [n for n in MLP_SHAPES]
    if prec == 0:
        names = [n for n in names if n in ('8x1', '4x2')]            # the other shapes are instantiated for bf16x3 only
    if DEFAULT not in names:
        names.insert(0, DEFAULT)
    names.sort(key=lambda n: n != DEFAULT)                              # the reference output first
    nv = min(int(ws['counters'][0]), cap)
    if exact_capacity:
        cap = min(cap, max((nv + 255) // 256 * 256, 256))
    tiles = (nv + 31) // 32
    tok_floats = tiles * 3 * 8 * 32 * 4
    tokens0 = ws['tokens'][:tok_floats].clone()
    out = torch.empty(max(tiles * 32, 1), 4, device=dev, dtype=torch.float32)
    A = _lib.addr
    stream = _ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch(shape_id):
        _lib.call('sherf_nerf_mlp', A(ws['counters']), A(ws['tokens']), A(ws['extras']), A(wc['stream']), A(wc['wbias']), prec,
                  shape_id, cap, A(out), stream)

    report, ref = {}, None
    for name in names:
        sid = MLP_SHAPES[name]
        entry = dict(shape_id=sid)
        try:
            ws['tokens'][:tok_floats].copy_(tokens0)
            out.fill_(float('nan'))
            launch(sid)
            torch.cuda.synchronize(dev)
            got = out[:nv].clone()
            if name in _CLOBBERS_TOKENS:
                ws['tokens'][:tok_floats].copy_(tokens0)
            if ref is None:
                ref = got
                entry.update(max_abs_diff=0.0, ok=bool(torch.isfinite(got).all()) if nv else True)
            else:
                diff = float((got - ref).abs().max()) if nv else 0.0
                entry.update(max_abs_diff=diff, ok=bool(diff <= tol))        # NaN compares False -> not ok
            for _ in range(warmup):
                launch(sid)
            # (a shape that clobbers `tokens` re-reads its own scratch from the second launch on: same work, other data)
            entry['ms'] = _time_launches(lambda: launch(sid), iters, dev)
        except RuntimeError as ex:                # launch refused by the library (argument check) -> not eligible
            entry.update(ok=False, error=str(ex)[:200])
        report[name] = entry
    ws['tokens'][:tok_floats].copy_(tokens0)
    ok = {n: e for n, e in report.items() if e.get('ok') and 'ms' in e}
    best = min(ok, key=lambda n: ok[n]['ms']) if ok else DEFAULT
    # a shape has to beat the default by more than the timing noise to replace it
    if best != DEFAULT and DEFAULT in ok and ok[best]['ms'] > 0.98 * ok[DEFAULT]['ms']:
        best = DEFAULT
    return dict(best=best, valid_samples=nv, capacity=cap, iters=iters, shapes=report)


def tune_gather(renderer, decoder, iters=20, warmup=3):
    """Times `sherf_gather_tokens` with the voxel-row loads under one branch per corner (default) and issued unconditionally
    (`mode | 4`, include/sherf_hip.h) on the renderer's last frame; the variant is eligible only if tokens and extras equal the
    default's.  -> {'best': 'branch' | 'branchless' | 'branchless128', 'variants': {...}}; the caller sets `renderer.gather_branchless`
    (False / True / '128')."""
    last = getattr(renderer, 'last', None)
    if not last:
        raise RuntimeError('tune_gather needs a rendered frame: call the renderer once first')
    ws, cap, b = last['ws'], int(last['cap']), last['bwd']
    dev = ws['tokens'].device
    wc = renderer._weights(decoder, dev)
    _, planes_f, feat_f, img4 = last['keep']
    P, (Hf, Wf), (H, W) = planes_f.shape[1], feat_f.shape[:2], img4.shape[:2]
    bounds = b['bounds'].detach().to(dtype=torch.float32).contiguous().view(-1)
    vox_sh = (_ct.c_int32 * 3)(*b['vox_sh'])
    nv = min(int(ws['counters'][0]), cap)
    tiles = (nv + 31) // 32
    nt, ne = tiles * 3 * 8 * 32 * 4, tiles * 12 * 32
    tokens0, extras0 = ws['tokens'][:nt].clone(), ws['extras'][:ne].clone()
    A = _lib.addr
    stream = _ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch(mode):
        _lib.call('sherf_gather_tokens', A(ws['counters']), A(ws['geom']), A(planes_f), P, A(feat_f), Hf, Wf, A(img4), H, W,
                  last['levels_struct'], A(wc['tok_bias']), A(bounds), A(b['vox_min']), vox_sh, mode, cap, A(ws['tokens']), A(ws['extras']),
                  stream)

    report, ref = {}, None
    for name, mode in (('branch', 0), ('branchless', 4), ('branchless128', 12)):
        ws['tokens'][:nt].fill_(float('nan')); ws['extras'][:ne].fill_(float('nan'))
        launch(mode)
        torch.cuda.synchronize(dev)
        got = (ws['tokens'][:nt].clone(), ws['extras'][:ne].clone())
        if ref is None:
            ref = got
            entry = dict(max_abs_diff=0.0, ok=bool(torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()))
        else:
            diff = max(float((got[0] - ref[0]).abs().max()), float((got[1] - ref[1]).abs().max())) if nv else 0.0
            entry = dict(max_abs_diff=diff, ok=bool(diff <= 0.0))
        for _ in range(warmup):
            launch(mode)
        entry['ms'] = _time_launches(lambda: launch(mode), iters, dev)
        report[name] = entry
    ws['tokens'][:nt].copy_(tokens0); ws['extras'][:ne].copy_(extras0)
    ok = [n for n in report if report[n]['ok']]
    best = min(ok, key=lambda n: report[n]['ms']) if ok else 'branch'
    if best != 'branch' and ('branch' not in ok or report[best]['ms'] > 0.98 * report['branch']['ms']):
        best = 'branch'
    return dict(best=best, valid_samples=nv, iters=iters, variants=report)


def _wall_ms(fn, iters, dev):
    """Wall-clock ms per call of `fn` over `iters` calls bracketed by device synchronisations (what bench.py measures)."""
    import time
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / iters


FRAME_KEYS = ('mlp_shape', 'gather_branchless', 'exact_grids')


def tune_frame(render, renderer, candidates, iters=10, warmup=2):
    """Times WHOLE frames under each candidate setting of the renderer's launch switches (dicts over FRAME_KEYS; the first one is
    the baseline) and verifies every candidate's (rgb, depth, acc) against the baseline's bit for bit.  `render()` renders one frame
    through `renderer` and returns its three outputs.  The renderer's switches are restored; -> {'best': index, 'frames': [...]}."""
    dev = next(renderer.parameters()).device
    saved = {k: getattr(renderer, k) for k in FRAME_KEYS}
    report, ref = [], None
    try:
        for cand in candidates:
            for k in FRAME_KEYS:
                setattr(renderer, k, cand.get(k, saved[k]))
            entry = {k: getattr(renderer, k) for k in FRAME_KEYS}
            try:
                got = [t.clone() for t in render()]
                torch.cuda.synchronize(dev)
                if ref is None:
                    ref = got
                    entry['ok'] = bool(all(torch.isfinite(t).all() for t in got[:1] + got[2:]))      # (depth may be inf by definition)
                else:
                    entry['ok'] = bool(all(torch.equal(a, b) for a, b in zip(got, ref)))
                for _ in range(warmup):
                    render()
                entry['ms'] = _wall_ms(render, iters, dev)
            except RuntimeError as ex:
                entry.update(ok=False, error=str(ex)[:200])
            report.append(entry)
    finally:
        for k, v in saved.items():
            setattr(renderer, k, v)
    ok = [i for i, e in enumerate(report) if e.get('ok') and 'ms' in e]
    best = min(ok, key=lambda i: report[i]['ms']) if ok else 0
    if best != 0 and (0 not in ok or report[best]['ms'] > 0.99 * report[0]['ms']):      # must beat the baseline by more than the noise
        best = 0
    return dict(best=best, iters=iters, frames=report)
