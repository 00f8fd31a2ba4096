"""Pad / zero-upsample / FIR-filter / downsample as ONE HIP gather kernel: drop-in for `torch_utils.ops.upfirdn2d`
(upfirdn2d.py:139-176 and the helpers :66-136, :279-389).  Same functions, argument conventions and defaults; the gradient is the
same operator with up and down exchanged and the filter flipped (upfirdn2d.py:252-270).  `impl='cuda'` = the HIP kernel (GPU
tensors, no silent fallback), `impl='ref'` = stock PyTorch ops on request.  Verified against the reference's outputs on the CPU and on the MI355X (tests/test_gpu_ops.py)."""
import torch

from . import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1}


def _scaling(s):
    sx, sy = (s, s) if isinstance(s, int) else s
    if not (isinstance(sx, int) and isinstance(sy, int) and sx >= 1 and sy >= 1):
        raise RuntimeError(f'upfirdn2d: bad scaling {s!r}')
    return sx, sy


def _padding(p):
    p = [p, p] if isinstance(p, int) else list(p)
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    if len(p) != 4 or not all(isinstance(v, int) for v in p):
        raise RuntimeError(f'upfirdn2d: bad padding {p!r}')
    return p


def _filter_size(f):
    if f is None:
        return 1, 1
    if not (isinstance(f, torch.Tensor) and f.ndim in (1, 2)):
        raise RuntimeError('upfirdn2d: f must be a 1-D (separable) or 2-D tensor or None')
    return int(f.shape[-1]), int(f.shape[0])                    # (fw, fh)


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """FIR filter in the form upfirdn2d() takes (upfirdn2d.py:66-113): [fh, fw], or [taps] when separable."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if f.ndim > 2 or f.numel() == 0:
        raise RuntimeError('setup_filter: f must have 0..2 dimensions and at least one element')
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if f.ndim != (1 if separable else 2):
        raise RuntimeError('setup_filter: a 2-D filter cannot be separable')
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return (f * gain ** (f.ndim / 2)).to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """x [N,C,H,W] -> [N,C,OH,OW] (upfirdn2d.py:115-137)."""
    if impl not in ('ref', 'cuda'):
        raise RuntimeError(f'upfirdn2d: impl must be "ref" or "cuda", got {impl!r}')
    if not (isinstance(x, torch.Tensor) and x.ndim == 4):
        raise RuntimeError('upfirdn2d: x must be a [N, C, H, W] tensor')
    up, down, padding = _scaling(up), _scaling(down), _padding(padding)
    _filter_size(f)
    if impl == 'ref':
        return _upfirdn2d_ref(x, f, up, down, padding, flip_filter, gain)
    if not x.is_cuda:
        raise RuntimeError('upfirdn2d(impl="cuda") needs a GPU tensor; the HIP path has no CPU fallback (pass impl="ref" explicitly)')
    if x.dtype not in _DTYPES:
        raise RuntimeError(f'upfirdn2d: unsupported dtype {x.dtype} (float32 / float16)')
    return _Upfirdn2d.apply(x, f, up, down, tuple(padding), bool(flip_filter), float(gain))


def _upfirdn2d_ref(x, f, up, down, padding, flip_filter, gain):
    """The reference's `impl='ref'` (upfirdn2d.py:139-193) restated: explicit zero stuffing, padding, grouped conv2d, slicing."""
    (upx, upy), (downx, downy), (px0, px1, py0, py1) = up, down, padding
    N, C, H, W = x.shape
    f = torch.ones([1, 1], dtype=torch.float32, device=x.device) if f is None else f
    z = x.new_zeros(N, C, H, upy, W, upx)
    z[:, :, :, 0, :, 0] = x
    z = z.reshape(N, C, H * upy, W * upx)
    z = torch.nn.functional.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    f = (f * gain ** (f.ndim / 2)).to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.ndim)))
    w = f[None, None].repeat([C, 1] + [1] * f.ndim)
    if f.ndim == 2:
        z = torch.nn.functional.conv2d(z, w, groups=C)
    else:
        z = torch.nn.functional.conv2d(z, w.unsqueeze(2), groups=C)
        z = torch.nn.functional.conv2d(z, w.unsqueeze(3), groups=C)
    return z[:, :, ::downy, ::downx]


def _launch(x, f2, up, down, pad, flip_filter, gain):
    x = x.contiguous()
    N, C, H, W = x.shape
    fh, fw = f2.shape
    px0, px1, py0, py1 = pad
    OW = (W * up[0] + px0 + px1 - fw + down[0]) // down[0]
    OH = (H * up[1] + py0 + py1 - fh + down[1]) // down[1]
    if OW < 1 or OH < 1:
        raise RuntimeError('upfirdn2d: the up-sampled, padded image is smaller than the filter')
    y = torch.empty(N, C, OH, OW, dtype=x.dtype, device=x.device)
    f2 = f2.to(device=x.device, dtype=torch.float32).contiguous()
    P = _lib.ptr
    for c0 in range(0, N * C, 65535):                               # the plane index rides in gridDim.z
        n_pl = min(65535, N * C - c0)
        xs, ys = x.view(N * C, H, W)[c0:c0 + n_pl], y.view(N * C, OH, OW)[c0:c0 + n_pl]
        _lib.call_ops('sherf_upfirdn2d', P(xs), P(f2), P(ys), 1, n_pl, H, W, fh, fw, up[0], up[1], down[0], down[1], px0, px1, py0, py1,
                      1 if flip_filter else 0, gain, _DTYPES[x.dtype], _lib.stream())
    return y


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, pad, flip_filter, gain):
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        if f.ndim == 1 and f.shape[0] == 1:
            f = f.square().unsqueeze(0)                             # separable 1-tap = full 1x1 (upfirdn2d.py:216-217)
        if f.ndim == 2:
            y = _launch(x, f, up, down, pad, flip_filter, gain)
        else:                                                       # separable: a row pass, then a column pass (:222-224)
            y = _launch(x, f.unsqueeze(0), (up[0], 1), (down[0], 1), (pad[0], pad[1], 0, 0), flip_filter, 1.0)
            y = _launch(y, f.unsqueeze(1), (1, up[1]), (1, down[1]), (0, 0, pad[2], pad[3]), flip_filter, gain)
        ctx.save_for_backward(f)
        ctx.cfg = (x.shape, up, down, pad, flip_filter, gain)
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        (_, _, ih, iw), (upx, upy), (downx, downy), (px0, _, py0, _), flip_filter, gain = ctx.cfg
        _, _, oh, ow = dy.shape
        fw, fh = _filter_size(f)
        p = (fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _Upfirdn2d.apply(dy, f, (downx, downy), (upx, upy), p, not flip_filter, gain)
        return dx, None, None, None, None, None, None


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Same-size FIR filtering (upfirdn2d.py:279-313)."""
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    return upfirdn2d(x, f, padding=[px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2], flip_filter=flip_filter,
                     gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Up-sampling by an integer factor (upfirdn2d.py:317-352)."""
    upx, upy = _scaling(up)
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Down-sampling by an integer factor (upfirdn2d.py:356-389)."""
    downx, downy = _scaling(down)
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
