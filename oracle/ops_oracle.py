"""TEST INFRASTRUCTURE (only tests/ may import this).  CPU restatement, in numpy, of the reference's two custom operators:

  bias_act   -- torch_utils/ops/bias_act.py:93-123 (`_bias_act_ref`) for the forward; its first / second derivatives are the closed
                forms the reference's CUDA kernel evaluates (bias_act.cu:27-151), written here in terms of the pre-activation.
  upfirdn2d  -- torch_utils/ops/upfirdn2d.py:139-193 (`_upfirdn2d_ref`): zero stuffing, padding / cropping, correlation with the
                (by default flipped) filter, decimation.

Pinned: tests/test_ops_oracle.py checks every function against tests/golden/ops.npz, the outputs (and autograd derivatives) of the
unmodified reference's own `_ref` implementations (oracle/make_golden_ops.py)."""
import numpy as np

SELU_SCALE, SELU_ALPHA = 1.0507009873554804934193349852946, 1.6732632423543772848170429916717
DEF = {'linear': (0.0, 1.0), 'relu': (0.0, np.sqrt(2)), 'lrelu': (0.2, np.sqrt(2)), 'tanh': (0.0, 1.0), 'sigmoid': (0.0, 1.0),
       'elu': (0.0, 1.0), 'selu': (0.0, 1.0), 'softplus': (0.0, 1.0), 'swish': (0.0, np.sqrt(2))}       # bias_act.py:23-33


def _sig(v):
    return 1.0 / (1.0 + np.exp(-v))


def _act(v, act, alpha):
    """(act(v), act'(v), act''(v)) in float64."""
    one, zero = np.ones_like(v), np.zeros_like(v)
    if act == 'linear':
        return v, one, zero
    if act == 'relu':
        return np.maximum(v, 0), (v > 0) * one, zero
    if act == 'lrelu':
        return np.where(v > 0, v, alpha * v), np.where(v > 0, 1.0, alpha), zero
    if act == 'tanh':
        t = np.tanh(v); return t, 1 - t * t, -2 * t * (1 - t * t)
    if act == 'sigmoid':
        s = _sig(v); return s, s * (1 - s), s * (1 - s) * (1 - 2 * s)
    if act == 'elu':
        e = np.exp(np.minimum(v, 0)); return np.where(v >= 0, v, e - 1), np.where(v >= 0, 1.0, e), np.where(v >= 0, 0.0, e)
    if act == 'selu':
        e = np.exp(np.minimum(v, 0)); k = SELU_SCALE * SELU_ALPHA
        return np.where(v >= 0, SELU_SCALE * v, k * (e - 1)), np.where(v >= 0, SELU_SCALE, k * e), np.where(v >= 0, 0.0, k * e)
    if act == 'softplus':
        s = _sig(v); return np.logaddexp(v, 0), s, s * (1 - s)
    if act == 'swish':
        s = _sig(v); d1 = s + v * s * (1 - s); return v * s, d1, 2 * s * (1 - s) + v * s * (1 - s) * (1 - 2 * s)
    raise KeyError(act)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, dy=None, ddx=None):
    """-> dict(y) and, when dy is given, dx = dL/dx and db; when ddx (the gradient arriving at dx) is given as well, the second
    backward's d2_x (w.r.t. x) and d2_dy (w.r.t. dy).  bias_act.py:93-123, :144-205."""
    alpha = DEF[act][0] if alpha is None else alpha
    gain = DEF[act][1] if gain is None else gain
    v = x.astype(np.float64)
    if b is not None:
        v = v + b.astype(np.float64).reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    a0, a1, a2 = _act(v, act, alpha)
    y = a0 * gain
    inside = np.ones_like(y, bool) if clamp is None else (y > -clamp) & (y < clamp)      # the clamp passes gradients strictly inside
    out = dict(y=(y if clamp is None else np.clip(y, -clamp, clamp)).astype(np.float32))
    if dy is not None:
        dx = dy.astype(np.float64) * gain * a1 * inside
        out['dx'] = dx.astype(np.float32)
        if b is not None:
            out['db'] = dx.sum(tuple(i for i in range(x.ndim) if i != dim)).astype(np.float32)
        if ddx is not None:
            out['d2_dy'] = (ddx.astype(np.float64) * gain * a1 * inside).astype(np.float32)
            out['d2_x'] = (ddx.astype(np.float64) * dy.astype(np.float64) * gain * a2 * inside).astype(np.float32)
    return out


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """x [N,C,H,W], f [fh,fw] or [taps] (separable) or None -> [N,C,OH,OW].  upfirdn2d.py:139-193."""
    upx, upy = (up, up) if isinstance(up, int) else up
    downx, downy = (down, down) if isinstance(down, int) else down
    p = [padding] * 4 if isinstance(padding, int) else list(padding)
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    px0, px1, py0, py1 = p
    N, C, H, W = x.shape
    f = np.ones((1, 1), np.float64) if f is None else f.astype(np.float64)
    if f.ndim == 1:
        f = np.outer(f, f)                      # the two separable passes of the reference equal one pass with the outer product
    z = np.zeros((N, C, H * upy, W * upx), np.float64)
    z[:, :, ::upy, ::upx] = x                                                              # :163-165
    z = np.pad(z, [(0, 0), (0, 0), (max(py0, 0), max(py1, 0)), (max(px0, 0), max(px1, 0))])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]   # :168-169
    k = f * gain
    if not flip_filter:
        k = k[::-1, ::-1]                                                                  # :174-175 (then conv2d = correlation)
    fh, fw = k.shape
    oh, ow = z.shape[2] - fh + 1, z.shape[3] - fw + 1
    y = np.zeros((N, C, oh, ow), np.float64)
    for i in range(fh):
        for j in range(fw):
            y += k[i, j] * z[:, :, i:i + oh, j:j + ow]
    return y[:, :, ::downy, ::downx].astype(np.float32)                                    # :192


def upfirdn2d_grad(dy, f, x_shape, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """dL/dx: the adjoint, as the same operator with up <-> down and the filter flipped (upfirdn2d.py:252-270)."""
    upx, upy = (up, up) if isinstance(up, int) else up
    downx, downy = (down, down) if isinstance(down, int) else down
    p = [padding] * 4 if isinstance(padding, int) else list(padding)
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    px0, _, py0, _ = p
    _, _, ih, iw = x_shape
    _, _, oh, ow = dy.shape
    f2 = np.ones((1, 1), np.float32) if f is None else (np.outer(f, f) if f.ndim == 1 else f)
    fh, fw = f2.shape
    q = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
    return upfirdn2d(dy, f2, up=(downx, downy), down=(upx, upy), padding=q, flip_filter=not flip_filter, gain=gain)
