"""Fused bias + activation + gain + clamp as a HIP operator: drop-in for `torch_utils.ops.bias_act` (bias_act.py:54-88).

Same signature, activation table and defaults as the reference; first and second order gradients through the same kernel
(`grad` = 1 / 2), as bias_act.py:144-205.  `impl='cuda'` (the default, name kept for drop-in compatibility) is the HIP kernel of
libsherf_hip_ops.so and needs GPU tensors: unlike the reference it does NOT fall back silently when the tensor lives on the CPU or
the library is missing.  `impl='ref'` -- the reference's own second implementation, plain PyTorch ops -- runs only when the caller
asks for it by name.  Verified against the unmodified reference's outputs on the CPU (kernel source) and on the MI355X (tests/test_gpu_ops.py)."""
import math

import torch

from . import _lib

# name -> (default alpha, default gain, kernel index, which forward tensors the derivative reads, has a 2nd derivative)   bias_act.py:23-33
activation_funcs = {
    'linear': dict(def_alpha=0.0, def_gain=1.0, idx=1, ref='', has_2nd_grad=False),
    'relu': dict(def_alpha=0.0, def_gain=math.sqrt(2), idx=2, ref='y', has_2nd_grad=False),
    'lrelu': dict(def_alpha=0.2, def_gain=math.sqrt(2), idx=3, ref='y', has_2nd_grad=False),
    'tanh': dict(def_alpha=0.0, def_gain=1.0, idx=4, ref='y', has_2nd_grad=True),
    'sigmoid': dict(def_alpha=0.0, def_gain=1.0, idx=5, ref='y', has_2nd_grad=True),
    'elu': dict(def_alpha=0.0, def_gain=1.0, idx=6, ref='y', has_2nd_grad=True),
    'selu': dict(def_alpha=0.0, def_gain=1.0, idx=7, ref='y', has_2nd_grad=True),
    'softplus': dict(def_alpha=0.0, def_gain=1.0, idx=8, ref='y', has_2nd_grad=True),
    'swish': dict(def_alpha=0.0, def_gain=math.sqrt(2), idx=9, ref='x', has_2nd_grad=True),
}
_DTYPES = {torch.float32: 0, torch.float16: 1}


def _resolve(act, alpha, gain, clamp):
    if act not in activation_funcs:
        raise RuntimeError(f'bias_act: unknown activation {act!r}')
    spec = activation_funcs[act]
    if clamp is not None and clamp < 0:
        raise RuntimeError('bias_act: clamp must be non-negative')
    return (spec, float(spec['def_alpha'] if alpha is None else alpha), float(spec['def_gain'] if gain is None else gain),
            float(-1 if clamp is None else clamp))


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(gain * act(x + b)) with `b` broadcast along `dim` (bias_act.py:54-88)."""
    if not isinstance(x, torch.Tensor):
        raise RuntimeError('bias_act: x must be a tensor')
    if impl not in ('ref', 'cuda'):
        raise RuntimeError(f'bias_act: impl must be "ref" or "cuda", got {impl!r}')
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        if not (isinstance(b, torch.Tensor) and b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]):
            raise RuntimeError('bias_act: b must be a 1-D tensor matching x.shape[dim]')
    if impl == 'ref':
        return _bias_act_ref(x, b, dim, act, alpha, gain, clamp)
    if not x.is_cuda:
        raise RuntimeError('bias_act(impl="cuda") needs a GPU tensor; the HIP path has no CPU fallback (pass impl="ref" explicitly)')
    if x.dtype not in _DTYPES:
        raise RuntimeError(f'bias_act: unsupported dtype {x.dtype} (float32 / float16)')
    return _BiasAct.apply(x, b, dim, act, alpha, gain, clamp)


def _bias_act_ref(x, b, dim, act, alpha, gain, clamp):
    """The reference's `impl='ref'` (bias_act.py:93-123) restated with stock PyTorch ops."""
    if b is not None:
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    F = torch.nn.functional
    x = {'linear': lambda t: t, 'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, alpha), 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
         'elu': F.elu, 'selu': F.selu, 'softplus': F.softplus, 'swish': lambda t: torch.sigmoid(t) * t}[act](x)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def _layout(x, dim):
    """-> (x dense in its own memory format, step of the bias index, channels_last?)  (bias_act.cpp:57,81-84)"""
    cl = x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1 and dim == 1
    x = x.contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
    step = 1 if cl else int(math.prod(x.shape[dim + 1:]))
    return x, step, cl


def _launch(x, b, xref, yref, dy, grad, dim, spec, alpha, gain, clamp):
    """One kernel launch; every tensor shares x's shape and memory format."""
    x, step, cl = _layout(x, dim)
    fmt = torch.channels_last if cl else torch.contiguous_format
    same = lambda t: None if t is None else t.to(x.dtype).contiguous(memory_format=fmt)
    xref, yref, dy = same(xref), same(yref), same(dy)
    bb = None if b is None else b.to(x.dtype).contiguous()
    y = torch.empty_like(x, memory_format=fmt)
    P = lambda t: _lib.ptr(t, channels_last_ok=cl)
    _lib.call_ops('sherf_bias_act', P(x), P(bb), P(xref), P(yref), P(dy), P(y), x.numel(), step, 1 if b is None else b.shape[0], grad,
                  spec['idx'], alpha, gain, clamp, _DTYPES[x.dtype], _lib.stream())
    return y


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, act, alpha, gain, clamp):
        spec = activation_funcs[act]
        identity = act == 'linear' and gain == 1 and clamp < 0 and b is None
        y = x if identity else _launch(x, b, None, None, None, 0, dim, spec, alpha, gain, clamp)
        keep_x = 'x' in spec['ref'] or spec['has_2nd_grad']
        ctx.save_for_backward(x if keep_x else None, b if keep_x else None, y if 'y' in spec['ref'] else None)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return y

    @staticmethod
    def backward(ctx, dy):
        dim, act, alpha, gain, clamp = ctx.cfg
        x, b, y = ctx.saved_tensors
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if act != 'linear' or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, dim, act, alpha, gain, clamp)
        if ctx.needs_input_grad[1]:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, dim, act, alpha, gain, clamp):
        spec = activation_funcs[act]
        dx = _launch(dy, b, x, y, None, 1, dim, spec, alpha, gain, clamp)
        ctx.save_for_backward(dy if spec['has_2nd_grad'] else None, x, b, y)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        dy, x, b, y = ctx.saved_tensors
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, dim, act, alpha, gain, clamp)
        if spec['has_2nd_grad'] and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _launch(d_dx, b, x, y, dy, 2, dim, spec, alpha, gain, clamp)
        if spec['has_2nd_grad'] and ctx.needs_input_grad[2]:
            d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
        return d_dy, d_x, d_b, None, None, None, None, None, None
