"""Stand-in for `lpips.LPIPS` (training/loss.py:23,28,160; needs VGG weights that are not available offline): TEST INFRASTRUCTURE.  A
fixed, differentiable perceptual-style distance -- squared differences of three scales of average-pooled, channel-normalised images --
so that the term exists, has a gradient and is identical wherever the reference's loss is driven in the tests.  NOT LPIPS."""
import torch
import torch.nn.functional as F


class LPIPS(torch.nn.Module):
    def __init__(self, net='vgg', **kw):
        super().__init__()

    def to(self, *a, **k):
        return self

    def forward(self, a, b):
        d = 0
        for s in (1, 2, 4):
            pa = F.avg_pool2d(a, s) if min(a.shape[-2:]) >= s else a
            pb = F.avg_pool2d(b, s) if min(b.shape[-2:]) >= s else b
            na = pa / (pa.norm(dim=1, keepdim=True) + 1e-3)
            nb = pb / (pb.norm(dim=1, keepdim=True) + 1e-3)
            d = d + ((na - nb) ** 2).sum(1).mean((-1, -2))
        return d.reshape(-1, 1, 1, 1)
