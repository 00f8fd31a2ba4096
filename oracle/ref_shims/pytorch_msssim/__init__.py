"""Stand-in for `pytorch_msssim.ssim` (training/loss.py:22,159; the package is absent offline): TEST INFRASTRUCTURE.  The published
definition with that package's defaults -- 11 x 11 Gaussian window of sigma 1.5, K = (0.01, 0.03), 'valid' filtering, mean over
channels and pixels -- written as ONE dense 2-D window per channel (not the separable form sherf_amd/loss.py uses), so that it is an
independent formulation to pin that one against (tests/test_loss.py).  "parity unpinned" against the real package."""
import torch
import torch.nn.functional as F


def ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    c = torch.arange(win_size, dtype=X.dtype, device=X.device) - win_size // 2
    g = torch.exp(-(c ** 2) / (2 * win_sigma ** 2)); g = g / g.sum()
    gy = g if X.shape[2] >= win_size else torch.ones(1, dtype=X.dtype, device=X.device)      # a side shorter than the window is not filtered
    gx = g if X.shape[3] >= win_size else torch.ones(1, dtype=X.dtype, device=X.device)
    w2 = (gy[:, None] * gx[None, :])[None, None].expand(X.shape[1], 1, -1, -1)
    blur = lambda t: F.conv2d(t, w2, groups=X.shape[1])
    mu1, mu2 = blur(X), blur(Y)
    s1, s2, s12 = blur(X * X) - mu1 * mu1, blur(Y * Y) - mu2 * mu2, blur(X * Y) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))
    val = m.flatten(2).mean(-1).mean(1)
    return val.mean() if size_average else val
