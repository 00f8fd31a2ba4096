"""Backward of the rendering hot path (BASELINE config 5: forward render + backward through HIP kernels).

EXPERIMENTAL / incomplete: round 1 ships the forward path.  What is here was written after the round's GPU budget was
spent and has only been cross-checked on the CPU (tests/test_backward_math.py restates each kernel's formulas in numpy and
compares them with autograd through the oracle); the GPU tests are marked `gpu_experimental`, outside `-m gpu`.

Built so far:  composite_backward  (d rgb_final, d acc) -> d (rgb, sigma) of every compact sample   [csrc/composite.hip]
Still missing: MLP / transformer backward, gather scatter, fold un-projection, sparse-encoder backward.
"""
import torch

from . import _lib


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
    out = torch.zeros(cap, 4, device=ws['sample_out'].device)
    _lib.call('sherf_composite_compact_bwd', P(ws['ray_base']), P(ws['ray_cnt']), P(ws['cs_idx']), P(ws['sample_out']),
              P(f32(ray_directions, R, 3)), P(f32(near, R)), P(f32(far, R)), R, S, 1 if white_back else 0,
              P(f32(d_rgb, R, 3)), P(f32(d_acc, R)), P(out), _lib.stream())
    nv = int(ws['counters'][0])
    return out[:nv]
