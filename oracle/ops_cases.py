"""TEST INFRASTRUCTURE. The seeded test cases shared by oracle/make_golden_ops.py (reference outputs), the oracle's own checks
and the kernel tests: inputs are regenerated from seeds everywhere, tests/golden/ops.npz holds OUTPUTS of the reference only."""
import numpy as np

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']

# (name, shape, dim, act, alpha, gain, clamp, with bias)
BIAS_ACT_CASES = [(f'ba_{a}', (3, 5, 7, 6), 1, a, None, None, None, True) for a in ACTS] + [
    ('ba_lrelu_clamp', (2, 8, 9, 5), 1, 'lrelu', 0.1, 1.7, 0.4, True),
    ('ba_swish_clamp', (2, 8, 9, 5), 1, 'swish', None, None, 0.6, True),
    ('ba_linear_gain_nobias', (4, 33), 1, 'linear', None, 0.5, None, False),
    ('ba_fc_lrelu', (6, 19), 1, 'lrelu', None, None, None, True),
    ('ba_dim0', (5, 11), 0, 'tanh', None, 2.0, 1.5, True),
    ('ba_ragged_1d', (1003,), 0, 'sigmoid', None, None, None, False),
]

# (name, x shape, filter spec, up, down, padding, flip_filter, gain)   filter spec: taps list (separable -> outer product unless
# 'sep'), or a 2-D list
UPFIRDN_CASES = [
    ('uf_up2', (2, 3, 8, 10), [1, 3, 3, 1], 2, 1, [2, 1, 2, 1], False, 4.0),            # upsample2d of the StyleGAN2 blocks
    ('uf_down2', (2, 3, 12, 10), [1, 3, 3, 1], 1, 2, [1, 1, 1, 1], False, 1.0),          # downsample2d
    ('uf_blur', (1, 4, 9, 7), [1, 3, 3, 1], 1, 1, [2, 1, 2, 1], False, 1.0),             # filter2d (same size)
    ('uf_crop', (1, 2, 11, 13), [1, 2, 1], 1, 1, [-1, 0, -2, 1], False, 1.0),            # negative padding = cropping
    ('uf_asym', (1, 2, 6, 9), [[1, 2, 3], [0, -1, 4]], (3, 2), (2, 3), [3, 2, 1, 4], False, 0.7),
    ('uf_asym_flip', (1, 2, 6, 9), [[1, 2, 3], [0, -1, 4]], (3, 2), (2, 3), [3, 2, 1, 4], True, 0.7),
    ('uf_sep12', (1, 2, 16, 16), ('sep', list(range(1, 13))), 2, 2, [5, 5, 5, 5], False, 2.0),   # separable 12-tap: two passes
    ('uf_identity', (1, 3, 5, 5), None, 1, 1, 0, False, 1.0),
]


def rng(name):
    return np.random.RandomState(abs(hash_name(name)) % (2 ** 31))


def hash_name(name):                       # stable across processes (python's hash() is salted)
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % (2 ** 61 - 1)
    return h


def bias_act_inputs(case):
    name, shape, dim, act, alpha, gain, clamp, with_b = case
    r = rng(name)
    x = (r.standard_normal(shape) * 1.5).astype(np.float32)
    b = (r.standard_normal(shape[dim]) * 0.5).astype(np.float32) if with_b else None
    dy = r.standard_normal(shape).astype(np.float32)               # incoming gradient of the first backward
    ddx = r.standard_normal(shape).astype(np.float32)              # incoming gradient of the second backward
    return x, b, dy, ddx


def upfirdn_inputs(case):
    name, shape, fspec, up, down, pad, flip, gain = case
    r = rng(name)
    x = r.standard_normal(shape).astype(np.float32)
    dy_seed = r.randint(1 << 30)
    if fspec is None:
        f = None
    elif isinstance(fspec, tuple) and fspec[0] == 'sep':
        f = np.asarray(fspec[1], np.float32); f = f / f.sum()
    else:
        f = np.asarray(fspec, np.float32)
        if f.ndim == 1:
            f = np.outer(f, f)
        f = f / f.sum()
    return x, f, dy_seed
