"""SURVEY section 8(f) rank 3, executed: the reference's OWN generator step -- `training.loss.StyleGAN2Loss.accumulate_gradients`
(loss.py:103-176) + the flat-gradient update of training_loop.py:354-386 -- driven around this package's renderer after
`sherf_amd.install.install()`, against the same step on the unmodified reference (tests/ref_trainstep_child.py; forward and backward
kernels run from their real source on the host, tests/hipcpu).  cv2 / pytorch_msssim / lpips are the documented stand-ins of
oracle/ref_shims in BOTH runs.  Needs /root/reference (build container only)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir('/root/reference/sherf'), reason='needs the reference checkout (build container only)')
def test_reference_training_step_drives_our_renderer():
    from tests.hipcpu import build_cpu
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the kernels')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ref_trainstep_child.py')], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1500, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith('TRAINSTEP_JSON ')]
    assert r.returncode == 0 and line, r.stderr[-1500:]
    res = json.loads(line[-1][len('TRAINSTEP_JSON '):])
    print(json.dumps(res)[:1500])
    assert res['hosted_renderer'] == 'sherf_amd.renderer' and res['loss_module'].startswith('/root/reference/')
    it0, it1 = res['iterations']
    # first iteration: identical weights on both sides -> every loss term and every one of the 240 gradients agrees
    for a, b in zip(it0['loss_new'], it0['loss_ref']):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (it0['loss_new'], it0['loss_ref'])
    assert it0['n_grads'] == 240 and it0['grad_cosine_min'] > 0.9999 and it0['grad_rel_err_max'] < 3e-2, it0
    assert it0['renderer_grad_rel']['decoder.pts_linears.0.weight'] < 1e-3 and it0['renderer_grad_rel']['renderer.conv1d_reprojection.weight'] < 1e-3
    # second iteration: after one Adam(beta1 = 0) step of each side (an lr * sign(g) move: gradient entries at rounding level take either sign)
    for a, b in zip(it1['loss_new'], it1['loss_ref']):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (it1['loss_new'], it1['loss_ref'])
    assert it1['grad_cosine_min'] > 0.98 and res['update_mismatch_l1'] < 0.1, (it1['grad_cosine_min'], res['update_mismatch_l1'])


def test_ssim_stand_in_is_an_independent_pin_of_ours():
    """sherf_amd.loss.ssim (separable windows) against the dense 2-D window formulation the reference's loss is driven with in the test
    above (oracle/ref_shims/pytorch_msssim) -- two independent restatements of the published definition, values and gradients."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_shims'))
    try:
        import pytorch_msssim
    finally:
        sys.path.pop(0)
    from sherf_amd import loss as L
    g = torch.Generator().manual_seed(3)
    for shape in ((2, 3, 40, 33), (1, 3, 9, 30), (1, 3, 64, 64)):
        x = torch.rand(*shape, generator=g, dtype=torch.float64).requires_grad_(True)
        y = (x.detach() + 0.1 * torch.randn(*shape, generator=g, dtype=torch.float64)).clamp(0, 1)
        a = L.ssim(x, y, data_range=1, size_average=False)
        b = pytorch_msssim.ssim(x, y, data_range=1, size_average=False)
        assert torch.allclose(a, b, atol=1e-12), (a, b)
        ga, = torch.autograd.grad(a.sum(), x)
        gb, = torch.autograd.grad(b.sum(), x)
        assert torch.allclose(ga, gb, atol=1e-12)
