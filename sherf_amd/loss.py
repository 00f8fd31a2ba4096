"""Reconstruction loss and weight update of SHERF's training loop -- the caller side of the path, restated
so that BASELINE config 5 (forward + backward through the HIP kernels, gradients averaged over the GPUs) runs without the reference's
GAN scaffolding.  Host-side torch code over a handful of small tensors per step: nothing here is a hot kernel.

What the reference does per generator step (loss.py:103-176, training_loop.py:354-386):

    gen = G.synthesis(G.mapping(z, c, input_img=obs_img), input_data, c, use_sr_module=..., noise_mode='none')
    m   = mask_at_box                                   # pixels whose ray hit the SMPL bounding box
    L   = 100 * mse(gen.image_raw[m] / 2 + 0.5, img[m]) + 10 * mse(gen.weights_image[m], bkgd_msk[m])
          + sum_b (1 - ssim(crop_b(gen), crop_b(img))) + sum_b lpips(crop_b(gen), crop_b(img))          # crop = boundingRect(m_b)
    L.mean().mul(gain).backward();  flat-grad all_reduce / world, nan_to_num(0, 1e5, -1e5);  opt.step()

`ssim` follows pytorch_msssim 's defaults (11-tap Gaussian, sigma 1.5, K = (0.01, 0.03), valid convolution, per-image mean): that
package is absent offline, so this restatement is **parity unpinned** against it (checked against an independent scipy evaluation of
the same definition, tests/test_loss.py).  LPIPS needs the VGG weights (not available offline): it is a constructor argument
(`lpips_fn`, default: the term is zero).  The discriminator phases of the reference's loop are no-ops in SHERF (loss_Dgen = 0,
training_loop.py:364-367) and are not reproduced.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import dist as sdist


def img2mse(x, y):
    """loss.py:25."""
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    """loss.py:26."""
    return -10.0 * torch.log(x) / np.log(10.0)


def bounding_rect(mask):
    """cv2.boundingRect of a binary mask [H,W] (loss.py:156): (x, y, w, h) of the non-zero pixels, (0, 0, 0, 0) when empty."""
    ys, xs = torch.nonzero(mask, as_tuple=True)
    if ys.numel() == 0:
        return 0, 0, 0, 0
    x0, x1, y0, y1 = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())
    return x0, y0, x1 - x0 + 1, y1 - y0 + 1


def _gauss_window(size, sigma, dtype, device):
    c = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(X, Y, data_range=1.0, size_average=False, win_size=11, win_sigma=1.5, K=(0.01, 0.03)):
    """Structural similarity of image batches [N,C,H,W] with pytorch_msssim.ssim's defaults (the call at loss.py:159:
    data_range=1, size_average=False -> one value per image): separable Gaussian window, 'valid' convolution, mean over channels and
    pixels of  (2 mu_x mu_y + C1)(2 s_xy + C2) / ((mu_x^2 + mu_y^2 + C1)(s_x^2 + s_y^2 + C2)).  An image side shorter than the
    window is left unfiltered along that side, as in that package."""
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    g = _gauss_window(win_size, win_sigma, X.dtype, X.device)
    ch = X.shape[1]

    def blur(t):
        if t.shape[2] >= win_size:
            t = F.conv2d(t, g.view(1, 1, -1, 1).expand(ch, 1, -1, 1), groups=ch)
        if t.shape[3] >= win_size:
            t = F.conv2d(t, g.view(1, 1, 1, -1).expand(ch, 1, 1, -1), groups=ch)
        return t
    mu1, mu2 = blur(X), blur(Y)
    s1 = blur(X * X) - mu1 * mu1
    s2 = blur(Y * Y) - mu2 * mu2
    s12 = blur(X * Y) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    val = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1) * cs).flatten(2).mean(-1).mean(1)        # [N]
    return val.mean() if size_average else val


class ReconstructionLoss:
    """The 'Gmain' phase of the reference's StyleGAN2Loss (loss.py:68-84, 103-176) with its call signature; other phases do nothing
    (as in SHERF, where the discriminator terms are zero).  `lpips_fn(pred, gt) -> [N]` or None."""

    def __init__(self, device, G, D=None, lpips_fn=None, neural_rendering_resolution_initial=64, **_ignored):
        self.device, self.G, self.D, self.lpips_fn = device, G, D, lpips_fn
        self.neural_rendering_resolution_initial = neural_rendering_resolution_initial

    def run_G(self, input_data, z, c, neural_rendering_resolution, use_sr_module=True, update_emas=False):
        c_cond = torch.zeros_like(c)                                                      # swapping_prob = 0 -> zeros (loss.py:73)
        ws = self.G.mapping(z, c_cond, input_img=input_data['obs_img_all'][:, 0], update_emas=update_emas)
        out = self.G.synthesis(ws, input_data, c, neural_rendering_resolution=neural_rendering_resolution, use_sr_module=use_sr_module,
                               update_emas=update_emas, noise_mode='none')
        return out, ws

    def terms(self, gen_img, input_data):
        """-> (loss [1], img_loss, acc_loss, ssim_sum [1], lpips_sum [1]) from a synthesis output dict and the batch."""
        real = input_data['img_all'][:, 0]
        B, _, H, W = real.shape
        m = input_data['mask_at_box_all'][:, 0].reshape(B, H, W).bool()
        img_loss = img2mse(gen_img['image_raw'].permute(0, 2, 3, 1)[m] / 2 + 0.5, real.permute(0, 2, 3, 1)[m])
        bk = input_data['bkgd_msk_all'].reshape(B, -1, H, W).to(torch.int8).permute(0, 2, 3, 1)
        acc_loss = img2mse(gen_img['weights_image'].permute(0, 2, 3, 1)[m], bk[m])
        ssim_sum = torch.zeros(1, device=real.device)
        lpips_sum = torch.zeros(1, device=real.device)
        for i in range(B):
            x, y, w, h = bounding_rect(m[i])
            pred = gen_img['image_raw'][i][:, y:y + h, x:x + w].unsqueeze(0) / 2 + 0.5
            gt = real[i][:, y:y + h, x:x + w].unsqueeze(0)
            ssim_sum = ssim_sum + ssim(pred, gt, data_range=1, size_average=False)
            if self.lpips_fn is not None:
                lpips_sum = lpips_sum + self.lpips_fn(pred, gt).reshape(-1)
        loss = 100 * img_loss + 10 * acc_loss + (1 - ssim_sum) + lpips_sum                # loss.py:165
        return loss, img_loss, acc_loss, ssim_sum, lpips_sum

    def accumulate_gradients(self, phase, input_data, real_img=None, real_c=None, gen_z=None, gen_c=None, gain=1, cur_nimg=0,
                             use_sr_module=True, recons_loss=True, rank=0):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
        if phase not in ('Gmain', 'Gboth'):
            return None
        gen_img, _ = self.run_G(input_data, gen_z, gen_c, self.neural_rendering_resolution_initial, use_sr_module=use_sr_module)
        loss, img_loss, acc_loss, ssim_sum, lpips_sum = self.terms(gen_img, input_data)
        loss.mean().mul(gain).backward()                                                  # loss.py:173
        return loss, img_loss, acc_loss, ssim_sum, lpips_sum, torch.zeros((), device=loss.device)


def update_weights(module, opt, num_gpus=None, scheduler=None):
    """training_loop.py:370-386: average the gradients that exist over the GPUs as ONE flat all-reduce, sanitise them, step."""
    sdist.allreduce_flat_grads([p for p in module.parameters() if p.numel() > 0], world_size=num_gpus)
    opt.step()
    if scheduler is not None:
        scheduler.step()


def training_step(G, opt, loss, input_data, z, c, gain=1, num_gpus=None, use_sr_module=False, scheduler=None):
    """One generator step as training_loop.py:354-386 runs it (batch_gpu = 1, the SHERF setting): zero the gradients, accumulate the
    reconstruction loss, exchange + sanitise the gradients, step the optimiser.  -> the loss tuple of accumulate_gradients."""
    opt.zero_grad(set_to_none=True)
    G.requires_grad_(True)
    out = loss.accumulate_gradients(phase='Gmain', input_data=input_data, gen_z=z, gen_c=c, gain=gain, use_sr_module=use_sr_module)
    G.requires_grad_(False)
    update_weights(G, opt, num_gpus=num_gpus, scheduler=scheduler)
    return out
