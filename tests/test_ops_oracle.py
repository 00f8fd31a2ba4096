"""bias_act / upfirdn2d: the numpy oracle (oracle/ops_oracle.py) and the wrappers' `impl='ref'` and padding helpers against the
outputs and autograd derivatives of the UNMODIFIED reference's `_ref` implementations (tests/golden/ops.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import ops_cases as C, ops_oracle as OO

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ops.npz'))


def _close(a, b, tol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize('case', C.BIAS_ACT_CASES, ids=[c[0] for c in C.BIAS_ACT_CASES])
def test_bias_act_oracle_matches_reference(case):
    name, shape, dim, act, alpha, gain, clamp, with_b = case
    x, b, dy, ddx = C.bias_act_inputs(case)
    o = OO.bias_act(x, b, dim, act, alpha, gain, clamp, dy=dy, ddx=ddx)
    _close(o['y'], GOLD[name + '.y'])
    _close(o['dx'], GOLD[name + '.dx'])
    if with_b:
        _close(o['db'], GOLD[name + '.db'], 2e-5)
    if name + '.d2_x' in GOLD.files:
        _close(o['d2_x'], GOLD[name + '.d2_x'], 5e-6)
        _close(o['d2_dy'], GOLD[name + '.d2_dy'])


@pytest.mark.parametrize('case', C.UPFIRDN_CASES, ids=[c[0] for c in C.UPFIRDN_CASES])
def test_upfirdn2d_oracle_matches_reference(case):
    name, shape, fspec, up, down, pad, flip, gain = case
    x, f, dy_seed = C.upfirdn_inputs(case)
    y = OO.upfirdn2d(x, f, up, down, pad, flip, gain)
    _close(y, GOLD[name + '.y'])
    dy = np.random.RandomState(dy_seed).standard_normal(y.shape).astype(np.float32)
    _close(OO.upfirdn2d_grad(dy, f, x.shape, up, down, pad, flip, gain), GOLD[name + '.dx'], 5e-6)


def test_wrappers_ref_path_and_padding_helpers():
    """sherf_amd.bias_act / upfirdn2d with impl='ref' (explicit) and the filter2d / upsample2d / downsample2d padding rules."""
    from sherf_amd import bias_act as B, upfirdn2d as U
    for case in C.BIAS_ACT_CASES:
        name, shape, dim, act, alpha, gain, clamp, with_b = case
        x, b, dy, ddx = C.bias_act_inputs(case)
        y = B.bias_act(torch.from_numpy(x), None if b is None else torch.from_numpy(b), dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp,
                       impl='ref')
        _close(y.numpy(), GOLD[name + '.y'])
    for case in C.UPFIRDN_CASES:
        name, shape, fspec, up, down, pad, flip, gain = case
        x, f, _ = C.upfirdn_inputs(case)
        y = U.upfirdn2d(torch.from_numpy(x), None if f is None else torch.from_numpy(f), up=up, down=down, padding=pad, flip_filter=flip,
                        gain=gain, impl='ref')
        _close(y.numpy(), GOLD[name + '.y'])
    x = torch.from_numpy(C.rng('wrappers').standard_normal((1, 2, 8, 6)).astype(np.float32))
    f = U.setup_filter([1, 3, 3, 1])
    _close(f.numpy(), GOLD['wr.filter'])
    _close(U.setup_filter(list(range(1, 9))).numpy(), GOLD['wr.sep'])
    _close(U.filter2d(x, f, impl='ref').numpy(), GOLD['wr.filter2d'])
    _close(U.upsample2d(x, f, up=2, impl='ref').numpy(), GOLD['wr.upsample2d'])
    _close(U.downsample2d(x, f, down=2, impl='ref').numpy(), GOLD['wr.downsample2d'])


def test_hip_path_refuses_cpu_tensors():
    from sherf_amd import bias_act as B, upfirdn2d as U
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        B.bias_act(torch.zeros(2, 3), act='lrelu')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        U.upfirdn2d(torch.zeros(1, 1, 4, 4), None)
    with pytest.raises(RuntimeError):
        B.bias_act(torch.zeros(2, 3), act='nope', impl='ref')
