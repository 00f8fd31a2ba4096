"""GPU tests of the backward kernels -- NOT part of `-m gpu`: the kernels were written after round 1's GPU budget was
spent and have not run on hardware yet.  Run with `pytest -m gpu_experimental` on an MI355X; once green they move under
the `gpu` marker.  Each kernel is checked against the oracle's stage-boundary gradients."""
import numpy as np
import pytest
import torch

from oracle import sherf_oracle as O
from tests import gpu_common as G

pytestmark = [pytest.mark.gpu_experimental, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]


@pytest.mark.parametrize('cfg', ['tiny', 'tiny_nv'])
def test_composite_backward_kernel(cfg):
    from sherf_amd.backward import composite_backward
    fx = G.fixture(cfg)
    loss, g = O.gradients_from_fixture(fx, G.seeded_state(), stages=True)
    h = G.hip_render(cfg)
    R = h['rgb'].shape[0]
    rs = np.random.RandomState(11)
    t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
    t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
    d_rgb = (2.0 * (h['rgb'] - t_rgb) / (R * 3)).cuda()
    d_acc = (2.0 * (h['acc'] - t_acc) / R).cuda()
    d = G.to_cuda(fx['input_data'])
    out = composite_backward(h['rend'], d_rgb, d_acc, d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0]).cpu()
    assert G.rel(out[:, :3], g['stage.sample_rgb']) < 1e-3
    assert G.rel(out[:, 3], g['stage.sample_sigma']) < 1e-3
