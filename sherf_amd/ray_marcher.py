"""MI355X-native `MipRayMarcher2`: drop-in for
/root/reference/sherf/training/volumetric_rendering/ray_marcher.py:20-70 on dense inputs.
(ImportanceRenderer itself composites its compacted samples with sherf_composite_compact.)"""
import torch
import torch.nn as nn

from . import _lib


class MipRayMarcher2(nn.Module):
    def __init__(self):
        super().__init__()

    def run_forward(self, colors, densities, depths, rays_d, rendering_options):
        if rendering_options['clamp_mode'] != 'relu':
            raise NotImplementedError("SHERF uses clamp_mode='relu' (train.py:332); softplus is not implemented in HIP")
        if not colors.is_cuda:
            raise RuntimeError('sherf_amd.MipRayMarcher2 runs on the GPU only (no CPU fallback)')
        if colors.shape[-1] != 3:
            raise RuntimeError(f'sherf_amd.MipRayMarcher2 composites 3 colour channels (SHERF: rgb), got {colors.shape[-1]}')
        if torch.is_grad_enabled() and any(t.requires_grad for t in (colors, densities, depths, rays_d)):
            raise RuntimeError('sherf_amd.MipRayMarcher2 (dense) is forward only: gradients flow through ImportanceRenderer '
                               '(enable_autograd, sherf_amd/backward.py); call it under torch.no_grad()')
        B, R, S = densities.shape[:3]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        c, s, t, d = f32(colors).view(B * R, S, 3), f32(densities).view(B * R, S), f32(depths).view(B * R, S), f32(rays_d).view(B * R, 3)
        mn, mx = torch.aminmax(t)                       # global clamp range (ray_marcher.py:57)
        dminmax = torch.stack([mn, mx]).contiguous()
        rgb = torch.empty(B * R, 3, device=c.device); dep = torch.empty(B * R, device=c.device)
        w = torch.empty(B * R, S, device=c.device)
        P = _lib.ptr
        _lib.call('sherf_composite_dense', P(c), P(s), P(t), P(d), B * R, S, 1 if rendering_options.get('white_back', False) else 0,
                  P(dminmax), P(rgb), P(dep), P(w), _lib.stream())
        return rgb.view(B, R, 3), dep.view(B, R, 1), w.view(B, R, S, 1)

    def forward(self, colors, densities, depths, rays_d, rendering_options):
        return self.run_forward(colors, densities, depths, rays_d, rendering_options)
