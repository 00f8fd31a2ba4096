"""MI355X-native `RaySampler`: drop-in for
/root/reference/sherf/training/volumetric_rendering/ray_sampler.py:17-61, plus the dataset-side ray generation
(get_rays / get_near_far, training/RenderPeople_dataset.py:14-27, 68-101, 129-134) moved onto the GPU."""
import torch
import torch.nn as nn

from . import _lib


class RaySampler(nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_origins_h, self.ray_directions, self.depths, self.image_coords, self.rendering_options = None, None, None, None, None

    def forward(self, cam2world_matrix, intrinsics, resolution):
        if not cam2world_matrix.is_cuda:
            raise RuntimeError('sherf_amd.RaySampler runs on the GPU only (no CPU fallback)')
        N = cam2world_matrix.shape[0]
        M = resolution * resolution
        c2w = cam2world_matrix.detach().float().contiguous().view(N, 16)
        K = intrinsics.detach().float().contiguous().view(N, 9)
        o = torch.empty(N, M, 3, device=c2w.device); d = torch.empty(N, M, 3, device=c2w.device)
        _lib.call('sherf_ray_sampler', _lib.ptr(c2w), _lib.ptr(K), N, int(resolution), _lib.ptr(o), _lib.ptr(d), _lib.stream())
        return o, d


def dataset_rays(K, R, T, bounds, H, W):
    """(ray_o [H*W,3], ray_d [H*W,3], near [H*W], far [H*W], mask_at_box [H*W] bool) for one camera, fp32 on device, computed on
    the reference's precision ladder (float64 camera algebra, float32 rays, float64 slab test: csrc/rays.hip).
    K [3,3], R [3,3], T [3] or [3,1] (float64 like the dataset's arrays; float32 inputs are widened), bounds [2,3] =
    posed-vertex world bounds +-5 cm (RenderPeople_dataset.py:216-219)."""
    if not K.is_cuda:
        raise RuntimeError('sherf_amd.dataset_rays runs on the GPU only (no CPU fallback)')
    dev = K.device
    Kinv = torch.linalg.inv(K.double()).contiguous()
    Rc, Tc, b = R.double().contiguous(), T.double().reshape(3).contiguous(), bounds.double().contiguous().view(6)
    n = H * W
    o = torch.empty(n, 3, device=dev); d = torch.empty(n, 3, device=dev)
    nr = torch.empty(n, device=dev); fr = torch.empty(n, device=dev)
    m = torch.empty(n, dtype=torch.uint8, device=dev)
    P = _lib.ptr
    _lib.call('sherf_dataset_rays', P(Kinv), P(Rc), P(Tc), P(b), H, W, P(o), P(d), P(nr), P(fr), P(m), _lib.stream())
    return o, d, nr, fr, m.bool()
