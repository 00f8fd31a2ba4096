"""Backward of the rendering hot path (BASELINE config 5: forward render + backward through HIP kernels).

How it is verified:
  * the mathematics: oracle/backward_explicit.py (hand-derived backward) == autograd through the oracle == the unmodified
    reference's gradients (tests/test_backward_math.py, tests/golden/grad_*.npz);
  * the orchestration (shapes, strides, transposition flags, accumulation, order): the functions below run on the CPU against
    a torch emulation of every C entry point and reproduce those gradients (tests/test_backward_dense.py);
  * the kernels (csrc/bwd_dense.hip, csrc/bwd_gemm.hip, csrc/bwd_encoder.hip, and the backward entry points of the forward library):
    from their unchanged source on the CPU (tests/test_hipcpu_kernels.py, tests/test_hipcpu_frame.py) and on the MI355X against the
    reference's gradient fingerprints (tests/test_gpu_backward.py, part of `-m gpu`).

Stages (backward order):  composite  ->  dense (decoder + transformer)  ->  taps / slot fusion  ->  sparse encoder.
"""
import ctypes

import torch

from . import _lib
from .backward_dense import HipOps, Mat, dense_backward
from .backward_encoder import encoder_backward
from .backward_taps import taps_backward


def composite_backward(renderer, d_rgb, d_acc, ray_directions, near, far, white_back=False):
    """Gradient of the loss w.r.t. the per-sample (rgb, sigma) produced by the last `renderer.forward` call.

    d_rgb [1,R,3], d_acc [1,R,1] (or [R,3] / [R]): gradients w.r.t. the outputs of ImportanceRenderer.forward; the ray
    tensors are the ones that call was given.  Returns [Nv, 4] in the compact sample order of the forward
    (renderer.last['ws']['cs_idx']): columns dL/d rgb (3), dL/d sigma (raw, pre-ReLU)."""
    last = renderer.last
    if last is None:
        raise RuntimeError('composite_backward needs the workspace of a preceding forward call')
    ws, R, S, cap = last['ws'], last['R'], last['S'], last['cap']
    f32 = lambda t, *shape: t.detach().to(dtype=torch.float32).contiguous().view(*shape)
    P = _lib.ptr
    nv = int(ws['counters'][0])                                   # (the forward is long done: the loss has been formed from its output)
    # rows of the frame's valid samples only: the kernel touches rows < nv (every ray's base + count lies below it), and the buffer used to be
    # capacity = R x S rows -- a 268 MB zero fill per step at 512 x 512 x 64 for 12 MB of gradient
    out = torch.zeros(max(nv, 1), 4, device=ws['sample_out'].device)
    _lib.call('sherf_composite_compact_bwd', P(ws['ray_base']), P(ws['ray_cnt']), P(ws['cs_idx']), P(ws['sample_out']),
              P(f32(ray_directions, R, 3)), P(f32(near, R)), P(f32(far, R)), R, S, 1 if white_back else 0,
              P(f32(d_rgb, R, 3)), P(f32(d_acc, R)), P(out), _lib.stream())
    return out[:nv]


def render_backward(renderer, decoder, d_rgb, d_acc):
    """Full backward of the last `renderer.forward(...)` (training mode).  d_rgb [1,R,3], d_acc [1,R,1].
    -> dict(params={reference name: grad}, planes [1,3,32,P,P], obs_feat [1,64,Hf,Wf], vertex_feat [N,32])."""
    last = renderer.last
    if last is None or 'bwd' not in last:
        raise RuntimeError('render_backward needs a preceding forward call')
    ws, pl, b = last['ws'], last['plan'], last['bwd']
    dev = ws['sample_out'].device
    f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
    state = {'renderer.' + k: v for k, v in renderer.state_dict().items()}
    state.update({'decoder.' + k: v for k, v in decoder.state_dict().items()})
    ops = HipOps()
    # ---- a16 ----
    d_sample = composite_backward(renderer, d_rgb, d_acc, b['ray_d'], b['near'], b['far'], b['white_back']).contiguous()
    n = d_sample.shape[0]
    if n == 0:
        raise RuntimeError('render_backward: no valid sample in the last frame')
    # ---- a13 + a14 ----
    tok, ext = Mat.empty(n, 96, dev), Mat.empty(n, 12, dev)                 # (sherf_bwd_untile writes both in full)
    ops.untile(ws['tokens'], ws['extras'], n, tok, ext)
    d_tin, grads, dWb_pe = dense_backward(ops, state, tok, ext, Mat(d_sample.view(-1), n, 4), use_trans=getattr(renderer, 'transformer', None) is not None)
    # ---- a10-a13 ----
    planes, obs_feat = f32(b['planes']), f32(b['obs_feat'])
    P_, (Hf, Wf) = planes.shape[-1], obs_feat.shape[-2:]
    L, meta, shapes = pl['L'], pl['meta'], pl['shapes']
    tap_meta = [m for m in meta if m['tap']]
    levels_ctx = [dict(raw=Mat.of(m['out']), bnparam=Mat(m['bnp'].view(-1), 1, 3 * m['cout']), n_rows=L[m['lev']]['n_rows'],
                       cap=L[m['lev']]['cap'], C=m['cout']) for m in tap_meta]
    bounds, vox_min = f32(b['bounds']).view(6), f32(b['vox_min']).view(3)
    vox_sh = (ctypes.c_int32 * 3)(*b['vox_sh'])

    def scatter(d_tiled, d_planes_f, d_feat_f, d_rows, d_bias):
        # samples binned by coarse voxel cell, accumulated in LDS windows, one device atomic per touched address per bin (the direct
        # form, sherf_gather_tokens_bwd: 20.3 ms of same-address atomics per step at 512 x 512 x 64)
        words = ctypes.c_int64(0)
        _lib.call('sherf_gather_bwd_scratch_words', last['levels_struct'], last['cap'], ctypes.byref(words))
        scratch = torch.empty(words.value, dtype=torch.int32, device=dev)
        _lib.call('sherf_gather_tokens_bwd_binned', _lib.ptr(ws['counters']), _lib.ptr(ws['geom']), _lib.ptr(d_tiled), P_, Hf, Wf, b['H'], b['W'],
                  last['levels_struct'], _lib.ptr(bounds), _lib.ptr(vox_min), vox_sh, last['cap'], ops._p(d_planes_f), ops._p(d_feat_f),
                  ops._p(d_rows[0]), ops._p(d_rows[1]), ops._p(d_rows[2]), ops._p(d_bias), _lib.ptr(scratch), words.value, _lib.stream())

    ctx = dict(n=n, P=P_, Hf=Hf, Wf=Wf, planes=Mat(planes.view(-1), 96, P_ * P_), obs_feat=Mat(obs_feat.view(-1), 64, Hf * Wf),
               levels=levels_ctx, scatter=scatter)
    taps = taps_backward(ops, state, ctx, d_tin, dWb_pe)
    grads.update(taps['grads'])
    # ---- a11 ----
    lev_ctx = [dict(keys=L[i]['keys'], wp=L[i]['wp'], n_rows=L[i]['n_rows'], dims=tuple(shapes[i]), cap=L[i]['cap']) for i in range(4)]
    layers = [dict(wname='renderer.encoder_3d.' + m['wname'], bname='renderer.encoder_3d.' + m['bname'], cin=m['cin'], cout=m['cout'],
                   down=m['down'], tap=m['tap'], lev_in=m['lev_in'], lev_out=m['lev'], raw=Mat.of(m['out']),
                   bnparam=Mat(m['bnp'].view(-1), 1, 3 * m['cout']), stats=Mat(pl['stats_flat'], 1, 2 * m['cout'], off=m['stats_off']))
              for m in meta]
    N = b['coord'].shape[0]
    ectx = dict(levels=lev_ctx, mult=L[0]['mult'], n_total=L[0]['n_total'], coord=b['coord'], N=N, g0=Mat.of(L[0]['g0']), layers=layers)
    d_feat, g_enc = encoder_backward(ops, state, ectx, taps['d_levels'])
    grads.update(g_enc)
    return dict(params=grads, planes=taps['d_planes'].tensor().view(1, 3, 32, P_, P_).clone(),
                obs_feat=taps['d_obs_feat'].tensor().view(1, 64, Hf, Wf).clone(), vertex_feat=d_feat.tensor().clone())


class RenderFunction(torch.autograd.Function):
    """autograd node around ImportanceRenderer.forward: (planes, obs_input_feature, sparse-voxel features, *parameters) ->
    (rgb, depth, acc).  Opt-in per renderer (`renderer.enable_autograd = True`; `sherf_amd.install()` sets it for the class).
    The depth output carries no gradient (as in the reference's losses, loss.py:103-176)."""

    @staticmethod
    def forward(ctx, renderer, decoder, call, planes, obs_feat, vox_feat, *params):
        with torch.no_grad():
            rgb, depth, acc = call()
        ctx.renderer, ctx.decoder = renderer, decoder
        ctx.names = [n for n, _ in _named_params(renderer, decoder)]
        # the backward reads the frame's compact samples, taps and encoder activations from the renderer's WORKSPACE (renderer.last), which
        # the next forward of the same renderer on the same stream overwrites: the node remembers WHICH frame it recorded
        ctx.frame = renderer.last
        ctx.mark_non_differentiable(depth)
        return rgb, depth, acc

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_acc):
        if ctx.renderer.last is not ctx.frame:
            raise RuntimeError('sherf_amd: backward of a frame whose workspace has been overwritten -- the same ImportanceRenderer rendered another '
                               'frame between this forward and its backward (gradient accumulation over several forwards, a second generator pass, '
                               'a grad-enabled forward in between).  Run every backward before the next forward of the same renderer, as the '
                               'reference loop does (training_loop.py:354-386), or use one renderer per frame in flight.')
        out = render_backward(ctx.renderer, ctx.decoder, d_rgb, d_acc)
        grads = []
        for name, p in _named_params(ctx.renderer, ctx.decoder):
            g = out['params'].get(name)
            grads.append(None if g is None else g.view(p.shape).to(p.dtype))
        return (None, None, None, out['planes'], out['obs_feat'], out['vertex_feat'], *grads)


def _named_params(renderer, decoder):
    return [('renderer.' + n, p) for n, p in renderer.named_parameters()] + [('decoder.' + n, p) for n, p in decoder.named_parameters()]
