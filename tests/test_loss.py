"""sherf_amd/loss.py: the reconstruction terms of loss.py:103-176 and the weight update of training_loop.py:354-386.
pytorch_msssim / lpips / cv2 are absent offline (parity unpinned): SSIM and the bounding rectangle are checked against independent
evaluations of their definitions."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from sherf_amd import loss as L


def _ssim_scipy(x, y, win=11, sigma=1.5, K=(0.01, 0.03), data_range=1.0):
    """float64, one image [C,H,W]: Gaussian correlation along both axes cropped to the 'valid' region."""
    c = np.arange(win) - win // 2
    g = np.exp(-c ** 2 / (2 * sigma ** 2)); g /= g.sum()

    def blur(t):
        for ax in (1, 2):
            if t.shape[ax] >= win:
                t = ndimage.correlate1d(t, g, axis=ax, mode='constant')
                t = np.take(t, np.arange(win // 2, t.shape[ax] - win // 2), axis=ax)
        return t
    x, y = x.astype(np.float64), y.astype(np.float64)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    m1, m2 = blur(x), blur(y)
    s1, s2, s12 = blur(x * x) - m1 * m1, blur(y * y) - m2 * m2, blur(x * y) - m1 * m2
    return float((((2 * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))).reshape(x.shape[0], -1).mean(-1).mean())


@pytest.mark.parametrize('shape', [(3, 40, 33), (3, 11, 64), (3, 9, 30)])
def test_ssim_matches_its_definition(shape):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, *shape, generator=g)
    y = (x + 0.1 * torch.randn(2, *shape, generator=g)).clamp(0, 1)
    got = L.ssim(x, y, data_range=1, size_average=False)
    assert got.shape == (2,)
    for i in range(2):
        assert abs(float(got[i]) - _ssim_scipy(x[i].numpy(), y[i].numpy())) < 2e-5
    assert abs(float(L.ssim(x, x)[0]) - 1.0) < 1e-6
    assert abs(float(L.ssim(x, y, size_average=True)) - float(got.mean())) < 1e-7


def test_bounding_rect_is_the_extent_of_the_nonzero_pixels():
    m = torch.zeros(20, 30, dtype=torch.bool)
    assert L.bounding_rect(m) == (0, 0, 0, 0)
    m[4:9, 7:21] = True; m[6, 25] = True
    assert L.bounding_rect(m) == (7, 4, 19, 5)
    assert L.bounding_rect(torch.ones(3, 5)) == (0, 0, 5, 3)


class _StubG(torch.nn.Module):
    """mapping / synthesis with the reference's signatures; the 'image' is a learnable tensor."""

    def __init__(self, H, W):
        super().__init__()
        self.img = torch.nn.Parameter(torch.zeros(1, 3, H, W))
        self.acc = torch.nn.Parameter(torch.full((1, 1, H, W), 0.5))
        self.unused = torch.nn.Parameter(torch.ones(4))
        self.seen = {}

    def mapping(self, z, c, input_img=None, update_emas=False):
        self.seen['c_sum'] = float(c.abs().sum()); self.seen['input_img'] = input_img
        return z[:, None, :]

    def synthesis(self, ws, input_data, c, neural_rendering_resolution=None, use_sr_module=True, update_emas=False, noise_mode='random'):
        self.seen['noise_mode'], self.seen['res'] = noise_mode, neural_rendering_resolution
        return {'image': self.img, 'image_raw': self.img, 'image_depth': self.acc.detach(), 'weights_image': self.acc}


def _batch(H=24, W=28, seed=1):
    g = torch.Generator().manual_seed(seed)
    mask = torch.zeros(1, 1, H * W, dtype=torch.bool)
    mm = torch.zeros(H, W, dtype=torch.bool); mm[3:20, 5:25] = True
    mask[0, 0] = mm.reshape(-1)
    return {'img_all': torch.rand(1, 1, 3, H, W, generator=g), 'mask_at_box_all': mask, 'obs_img_all': torch.rand(1, 1, 3, H, W, generator=g),
            'bkgd_msk_all': (torch.rand(1, 1, H * W, generator=g) > 0.5).to(torch.uint8)}, mm


def test_terms_and_training_step_follow_the_reference_formula():
    H, W = 24, 28
    d, mm = _batch(H, W)
    G = _StubG(H, W)
    lpips_fn = lambda a, b: ((a - b) ** 2).mean().reshape(1) * 3.0
    loss = L.ReconstructionLoss(torch.device('cpu'), G, lpips_fn=lpips_fn, neural_rendering_resolution_initial=48)
    opt = torch.optim.SGD(G.parameters(), lr=0.1)
    z, c = torch.randn(1, 8), torch.ones(1, 25)
    out = L.training_step(G, opt, loss, d, z, c, gain=2, num_gpus=1)
    total, img_l, acc_l, ssim_s, lp, dgen = [t.detach() for t in out]
    # what G was called with (loss.py:68-84): conditioning zeroed, the observation image into mapping, noise off
    assert G.seen['c_sum'] == 0.0 and G.seen['noise_mode'] == 'none' and G.seen['res'] == 48 and torch.equal(G.seen['input_img'], d['obs_img_all'][:, 0])
    # the terms, written out (the image parameter started at 0 -> prediction 0.5 everywhere)
    real = d['img_all'][0, 0]
    exp_img = float(((0.5 - real[:, mm]) ** 2).mean())
    bk = d['bkgd_msk_all'].reshape(H, W).float()
    exp_acc = float(((0.5 - bk[mm]) ** 2).mean())
    crop_gt = real[:, 3:20, 5:25][None]
    exp_ssim = float(L.ssim(torch.full_like(crop_gt, 0.5), crop_gt)[0])
    exp_lp = float(lpips_fn(torch.full_like(crop_gt, 0.5), crop_gt))
    assert abs(float(img_l) - exp_img) < 1e-6 and abs(float(acc_l) - exp_acc) < 1e-6
    assert abs(float(ssim_s) - exp_ssim) < 1e-6 and abs(float(lp) - exp_lp) < 1e-6 and float(dgen) == 0.0
    assert abs(float(total) - (100 * exp_img + 10 * exp_acc + (1 - exp_ssim) + exp_lp)) < 1e-4
    # the step: d(2 * 10 * acc_loss)/d acc = 2 * 10 * 2 (0.5 - bk) / n inside the mask, 0 outside; SGD with lr 0.1
    n = int(mm.sum())
    exp_acc_param = 0.5 - 0.1 * (2 * 10 * 2 * (0.5 - bk) / n) * mm
    assert torch.allclose(G.acc.detach()[0, 0], exp_acc_param, atol=1e-6)
    assert torch.equal(G.unused.detach(), torch.ones(4))                       # no gradient -> untouched (training_loop.py:372)
    assert not any(p.requires_grad for p in G.parameters())                      # requires_grad_(False) after the phase (:369)
    assert loss.accumulate_gradients('Dmain', d) is None


def test_update_weights_sanitises_non_finite_gradients():
    m = torch.nn.Linear(3, 1)
    m.weight.grad = torch.tensor([[float('nan'), float('inf'), -float('inf')]]); m.bias.grad = torch.tensor([2.0])
    w0, b0 = m.weight.detach().clone(), m.bias.detach().clone()
    L.update_weights(m, torch.optim.SGD(m.parameters(), lr=1e-6), num_gpus=1)
    assert torch.allclose(m.weight.detach(), w0 - 1e-6 * torch.tensor([[0.0, 1e5, -1e5]])) and torch.allclose(m.bias.detach(), b0 - 2e-6)
