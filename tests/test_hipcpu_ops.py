"""The HIP kernels of libsherf_hip_ops.so (csrc/ops_bias_act.hip, csrc/ops_upfirdn2d.hip) executed on the CPU from their unchanged
source (tests/hipcpu) through the drop-in wrappers sherf_amd.bias_act / sherf_amd.upfirdn2d, autograd included, against the
outputs and derivatives of the unmodified reference's `_ref` implementations (tests/golden/ops.npz)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import ops_cases as C
from sherf_amd import _lib
from tests import gpu_common as G
from tests.hipcpu import build_cpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ops.npz'))


@pytest.fixture(scope='module')
def ops(tmp_path_factory):
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang (_Float16) for the host build')
    path = build_cpu.build('sherf_hipcpu_ops', ['ops_lib.hip', 'ops_bias_act.hip', 'ops_upfirdn2d.hip'], str(tmp_path_factory.mktemp('hipcpu_ops')),
                           compiler=build_cpu.CLANG)
    mp = pytest.MonkeyPatch()
    mp.setattr(_lib, 'LIB_OPS_PATH', path); mp.setattr(_lib, '_lib_ops', None)
    mp.setattr(_lib, 'ptr', lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr()))
    mp.setattr(_lib, 'stream', lambda: ctypes.c_void_p(0))
    from sherf_amd import bias_act, upfirdn2d
    yield bias_act, upfirdn2d
    mp.undo()


def _dev(a, dtype=torch.float32, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype).as_subclass(G._HostTensor)
    return t.requires_grad_(True) if grad else t


def _close(a, b, tol):
    a, b = np.asarray(G.plain(a).float().numpy(), np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize('case', C.BIAS_ACT_CASES, ids=[c[0] for c in C.BIAS_ACT_CASES])
def test_bias_act_kernel_forward_and_both_derivatives(ops, case):
    B, _ = ops
    name, shape, dim, act, alpha, gain, clamp, with_b = case
    x, b, dy, ddx = C.bias_act_inputs(case)
    xt, bt, dyt = _dev(x, grad=True), (_dev(b, grad=True) if with_b else None), _dev(dy, grad=True)
    y = B.bias_act(xt, bt, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    _close(y, GOLD[name + '.y'], 3e-6)
    gs = torch.autograd.grad(y, [xt] + ([bt] if with_b else []), dyt, create_graph=True)
    _close(gs[0], GOLD[name + '.dx'], 3e-6)
    if with_b:
        _close(gs[1], GOLD[name + '.db'], 2e-5)
    if name + '.d2_x' in GOLD.files:
        g2 = torch.autograd.grad(gs[0], [xt, dyt], _dev(ddx), allow_unused=True)
        _close(g2[1], GOLD[name + '.d2_dy'], 3e-6)
        _close(g2[0] if g2[0] is not None else torch.zeros_like(xt), GOLD[name + '.d2_x'], 1e-5)


def test_bias_act_kernel_channels_last_and_half(ops):
    B, _ = ops
    case = C.BIAS_ACT_CASES[2]                                                   # lrelu, [3, 5, 7, 6]
    name = case[0]
    x, b, dy, _ = C.bias_act_inputs(case)
    xt = _dev(x).contiguous(memory_format=torch.channels_last).as_subclass(G._HostTensor)
    y = B.bias_act(xt, _dev(b), act='lrelu')
    assert G.plain(y).is_contiguous(memory_format=torch.channels_last)            # keeps the memory format (bias_act.py:148)
    _close(y, GOLD[name + '.y'], 3e-6)
    yh = B.bias_act(_dev(x, torch.float16), _dev(b, torch.float16), act='lrelu')
    assert yh.dtype == torch.float16
    _close(yh, GOLD[name + '.y'], 4e-3)


@pytest.mark.parametrize('case', C.UPFIRDN_CASES, ids=[c[0] for c in C.UPFIRDN_CASES])
def test_upfirdn2d_kernel_forward_and_gradient(ops, case):
    _, U = ops
    name, shape, fspec, up, down, pad, flip, gain = case
    x, f, dy_seed = C.upfirdn_inputs(case)
    xt = _dev(x, grad=True)
    ft = None if f is None else torch.from_numpy(f)
    y = U.upfirdn2d(xt, ft, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    _close(y, GOLD[name + '.y'], 3e-6)
    dy = np.random.RandomState(dy_seed).standard_normal(tuple(y.shape)).astype(np.float32)
    _close(torch.autograd.grad(y, xt, _dev(dy))[0], GOLD[name + '.dx'], 5e-6)
    yh = U.upfirdn2d(_dev(x, torch.float16), ft, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    _close(yh, GOLD[name + '.y'], 6e-3)


def test_upfirdn2d_helpers_through_the_kernel(ops):
    _, U = ops
    x = _dev(C.rng('wrappers').standard_normal((1, 2, 8, 6)).astype(np.float32))
    f = U.setup_filter([1, 3, 3, 1])
    _close(U.filter2d(x, f), GOLD['wr.filter2d'], 3e-6)
    _close(U.upsample2d(x, f, up=2), GOLD['wr.upsample2d'], 3e-6)
    _close(U.downsample2d(x, f, down=2), GOLD['wr.downsample2d'], 3e-6)
