"""CPU checks of the formulas the backward kernels implement (csrc/composite.hip: composite_compact_bwd_kernel, ...):
each kernel's arithmetic restated in numpy, loop for loop, against autograd through the oracle
(oracle.sherf_oracle.gradients_from_fixture(stages=True)), itself pinned to the reference's backward."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, sherf_oracle as O


@pytest.fixture(scope='module')
def state(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    return {n: torch.from_numpy(fixtures.seeded_param(n, s)) for n, s in shapes.items() if fixtures.seeded_param(n, s) is not None}


def composite_bwd_like_the_kernel(t, valid, S, sample_rgb, sample_sigma, ray_d, d_rgb, d_acc, white_back=False):
    """composite_compact_bwd_kernel in numpy (float32 arithmetic, two forward sweeps per ray)."""
    f = np.float32
    out = np.zeros((valid.shape[0], 4), f)
    ray_of = valid // S
    for r in np.unique(ray_of):
        idx = np.nonzero(ray_of == r)[0]                      # the ray's compact samples, ascending k
        ks = valid[idx] - r * S
        dn = f(np.sqrt((ray_d[r].astype(f) ** 2).sum()))
        g = f(2.0) * d_rgb[r].astype(f)
        gconst = f(d_acc[r]) - (g.sum() if white_back else f(0))
        def sweep(total):
            T, prefix = f(1), f(0)
            tot = f(0)
            for i, k in zip(idx, ks):
                delta = (f(1e10) if k == S - 1 else t[r, k + 1] - t[r, k]) * dn
                e = np.exp(-(max(sample_sigma[i], f(0)) * delta), dtype=f)
                alpha = f(1) - e
                w = alpha * T
                gw = (g * sample_rgb[i]).sum() + gconst
                tot += gw * w
                if total is not None:
                    prefix += gw * w
                    dalpha = gw * T - (total - prefix) / (f(1) - alpha + f(1e-10))
                    out[i, :3] = g * w
                    out[i, 3] = dalpha * delta * e if sample_sigma[i] > 0 else 0
                T = T * (f(1) - alpha + f(1e-10))
            return tot
        sweep(sweep(None))
    return out


@pytest.mark.parametrize('cfg', ['tiny', 'tiny_nv'])
def test_composite_backward_formula(cfg, state):
    fx = fixtures.renderer_inputs(cfg)
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    r = O.render_from_fixture(fx, state, training=True)
    R, S = r['t'].shape
    rs = np.random.RandomState(11)                                           # the stub loss's targets (O.stub_loss)
    t_rgb = rs.uniform(-1, 1, (1, R, 3)).astype(np.float32)[0]
    t_acc = rs.uniform(0, 1, (1, R, 1)).astype(np.float32)[0, :, 0]
    d_rgb = 2.0 * (r['rgb'].numpy() - t_rgb) / (R * 3)
    d_acc = 2.0 * (r['acc'].numpy() - t_acc) / R
    ray_d = fx['input_data']['ray_d_all'][0, 0]
    ours = composite_bwd_like_the_kernel(r['t'].numpy(), r['valid'].numpy(), S, r['sample_rgb'].numpy(), r['sample_sigma'].numpy(),
                                         ray_d, d_rgb.astype(np.float32), d_acc.astype(np.float32))
    ref_rgb, ref_sig = g['stage.sample_rgb'].numpy(), g['stage.sample_sigma'].numpy()
    assert np.abs(ours[:, :3] - ref_rgb).max() < 1e-5 * np.abs(ref_rgb).max() + 1e-12
    assert np.abs(ours[:, 3] - ref_sig).max() < 2e-4 * np.abs(ref_sig).max() + 1e-12
