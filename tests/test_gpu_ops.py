"""bias_act / upfirdn2d HIP kernels on the MI355X: the same checks as tests/test_hipcpu_ops.py (goldens of the unmodified reference's
`_ref` implementations, both derivatives, channels_last, float16) with real device tensors and libsherf_hip_ops.so.
Under `-m gpu` since round 2 (first hardware run: green after the channels_last pointer check was fixed)."""
import numpy as np
import pytest
import torch

from oracle import ops_cases as C
from tests import test_hipcpu_ops as H

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]


@pytest.fixture(scope='module')
def gpu_ops():
    from sherf_amd import bias_act, upfirdn2d
    mp = pytest.MonkeyPatch()

    def dev(a, dtype=torch.float32, grad=False):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()
        return t.requires_grad_(True) if grad else t
    mp.setattr(H, '_dev', dev)
    yield bias_act, upfirdn2d
    mp.undo()


@pytest.mark.parametrize('case', C.BIAS_ACT_CASES, ids=[c[0] for c in C.BIAS_ACT_CASES])
def test_bias_act(gpu_ops, case):
    H.test_bias_act_kernel_forward_and_both_derivatives(gpu_ops, case)


def test_bias_act_layouts(gpu_ops):
    H.test_bias_act_kernel_channels_last_and_half(gpu_ops)


@pytest.mark.parametrize('case', C.UPFIRDN_CASES, ids=[c[0] for c in C.UPFIRDN_CASES])
def test_upfirdn2d(gpu_ops, case):
    H.test_upfirdn2d_kernel_forward_and_gradient(gpu_ops, case)


def test_upfirdn2d_helpers(gpu_ops):
    H.test_upfirdn2d_helpers_through_the_kernel(gpu_ops)


def test_bandwidth_report(gpu_ops):
    """Both operators are HBM-bound: report achieved GB/s on StyleGAN2-sized tensors (algorithmic bytes = read x + write y)."""
    B, U = gpu_ops
    x = torch.randn(4, 512, 256, 256, device='cuda'); b = torch.randn(512, device='cuda')
    f = U.setup_filter([1, 3, 3, 1]).cuda()
    for name, fn, nbytes in (('bias_act lrelu', lambda: B.bias_act(x, b, act='lrelu'), 2 * x.numel() * 4),
                             ('upsample2d x2', lambda: U.upsample2d(x[:1], f), 5 * x[:1].numel() * 4)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'{name}: {ms:.3f} ms, {nbytes / ms / 1e6:.0f} GB/s of 8000')
