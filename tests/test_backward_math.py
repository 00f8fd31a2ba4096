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


# ---------------------------------------------------------------------------------------------------------------------
# explicit (hand-derived) backward, stage by stage, against autograd through the oracle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module', params=['tiny_nv', 'tiny'])
def case(request, state):
    fx = fixtures.renderer_inputs(request.param)
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
    return fx, r, g


def _close(a, b, tol=1e-4):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-14


def _image_grads(r):
    R = r['rgb'].shape[0]
    rs = np.random.RandomState(11)
    t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
    t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
    return 2.0 * (r['rgb'] - t_rgb) / (R * 3), 2.0 * (r['acc'] - t_acc) / R


def test_explicit_composite(case):
    from oracle import backward_explicit as BX
    fx, r, g = case
    R, S = r['t'].shape
    col = torch.zeros(R * S, 3); sig = torch.full((R * S,), -80.0)
    col[r['valid']] = r['sample_rgb']; sig[r['valid']] = r['sample_sigma']
    d_rgb, d_acc = _image_grads(r)
    ray_d = torch.from_numpy(fx['input_data']['ray_d_all'][0, 0])
    d_col, d_sig = BX.composite_bwd(col.view(R, S, 3), sig.view(R, S), r['t'], ray_d, d_rgb, d_acc)
    assert _close(d_col.reshape(-1, 3)[r['valid']], g['stage.sample_rgb'])
    assert _close(d_sig.reshape(-1)[r['valid']], g['stage.sample_sigma'], 2e-4)


def test_explicit_decoder_transformer_fuse(case, state):
    from oracle import backward_explicit as BX
    fx, r, g = case
    pe_x, pe_v = O.positional_encoding(r['x_c'], 6), O.positional_encoding(r['v_c'], 4)
    d_z, gd = BX.decoder_bwd(state, pe_x, r['tokens_out'], pe_v, g['stage.sample_rgb'], g['stage.sample_sigma'])
    assert _close(d_z, g['stage.tokens_out'])
    for k, v in gd.items():
        assert _close(v, g[k], 2e-4), k
    d_tok, gt = BX.transformer_bwd(state, r['tokens_in'], g['stage.tokens_out'])
    assert _close(d_tok, g['stage.tokens_in'], 2e-4)
    for k, v in gt.items():
        assert _close(v, g[k], 5e-4), k
    bounds = torch.from_numpy(fx['input_data']['t_world_bounds']).view(2, 3)
    tri = O.triplane_features(torch.from_numpy(fx['planes'])[0], r['x_c'], bounds)
    d_tri, d_f2d, d_f3d, gf = BX.fuse_bwd(state, tri, r['f2d'], r['f3d'], g['stage.tokens_in'])
    assert _close(d_f2d, g['stage.f2d']) and _close(d_f3d, g['stage.f3d'])
    for k, v in gf.items():
        assert _close(v, g[k], 2e-4), k
    # taps: tri-plane scatter, pixel-aligned scatter, conv1d_projection, voxel trilinear scatter
    d_planes = BX.triplane_bwd(fx['planes'].shape[1:], r['x_c'], bounds, d_tri)
    assert _close(d_planes, g['input.planes'][0], 2e-4)
    H, W = fx['input_data']['obs_img_all'].shape[-2:]
    d_feat = BX.pixel_aligned_bwd(fx['obs_feat'].shape[1:], (H, W), r['uv'], g['stage.f2d'])
    assert _close(d_feat, g['input.obs_feat'][0], 2e-4)
    d_raw, gp = BX.projection_bwd(state, r['f3d_raw'], g['stage.f3d'])
    assert _close(d_raw, g['stage.f3d_raw'])
    for k, v in gp.items():
        assert _close(v, g[k], 2e-4), k
    off = 0
    for i, (keys, feats, shape) in enumerate(r['taps']):
        C = feats.shape[1]
        d_lv = BX.trilinear_sparse_bwd(keys, feats.shape[0], shape, r['grid'], g['stage.f3d_raw'][:, off:off + C])
        assert _close(d_lv, g[f'stage.level{i}'], 2e-4), i
        off += C


def test_explicit_sparse_encoder(case, state):
    """BatchNorm (batch statistics over the reference's row set, duplicate voxels included) + sparse conv backward."""
    from oracle import backward_explicit as BX
    fx, r, g = case
    sp = r['sp_input']
    with torch.no_grad():
        taps, cache = BX.encoder_forward_cached(state, torch.from_numpy(fx['vertex_feat']), sp['coord'], sp['out_sh'])
        for (k1, f1, s1), (k2, f2, s2) in zip(taps, r['taps']):
            assert torch.equal(k1, k2) and _close(f1, f2, 1e-6)
        d_vf, ge = BX.encoder_bwd(state, cache, [g[f'stage.level{i}'] for i in range(3)])
    assert _close(d_vf, g['input.vertex_feat'], 5e-4)
    names = [k for k in g if k.startswith('renderer.encoder_3d.')]
    assert len(names) == 39 and set(names) == set(ge)
    for k in names:
        assert _close(ge[k], g[k], 2e-3), (k, float((ge[k] - g[k]).abs().max()), float(g[k].abs().max()))


@pytest.mark.parametrize('cfg', ['tiny_nv', 'tiny'])
def test_explicit_chain_matches_reference_gradients(cfg, state, golden_dir):
    """The whole explicit chain against the fingerprints of the UNMODIFIED reference's gradients (make_golden.run_grad)."""
    from oracle import backward_explicit as BX
    ref = np.load(os.path.join(golden_dir, f'grad_{cfg}.npz'))
    loss, grads = BX.backward_from_fixture(fixtures.renderer_inputs(cfg), state)
    assert abs(loss - float(ref['loss'])) < 1e-5 * abs(float(ref['loss']))
    names = [k for k in ref.files if k not in ('loss', 'ref_cpu_seconds')]
    assert set(names) == set(grads), (set(names) ^ set(grads))
    for k in names:
        ours, r = O.grad_fingerprint(grads[k]), ref[k]
        assert abs(ours[2] - r[2]) < 5e-3 * r[2] + 1e-30, (k, ours[2], r[2])                           # L2 norm
        assert np.linalg.norm(ours[3:] - r[3:]) < 2e-2 * np.linalg.norm(r[3:]) + 1e-30, k            # 64 sampled entries


def test_folded_formulation_of_the_tap_backward(case, state):
    """The HIP path applies conv1d_projection / conv1d_reprojection to the TABLES (linearity of interpolation); its backward
    is one scatter of d_tokens into the folded tables' gradients plus per-texel / per-row unfold products."""
    from oracle import backward_explicit as BX
    fx, r, g = case
    r = dict(r)
    r['_bounds'] = torch.from_numpy(fx['input_data']['t_world_bounds']).view(2, 3)
    H, W = fx['input_data']['obs_img_all'].shape[-2:]
    out = BX.folded_taps_bwd(state, torch.from_numpy(fx['planes'])[0], torch.from_numpy(fx['obs_feat'])[0], (H, W), r, g['stage.tokens_in'])
    assert _close(out['d_planes'], g['input.planes'][0], 2e-4)
    assert _close(out['d_obs_feat'], g['input.obs_feat'][0], 2e-4)
    for i in range(3):
        assert _close(out['d_levels'][i], g[f'stage.level{i}'], 5e-4), i
    for k, v in out['grads'].items():
        assert _close(v, g[k], 5e-4), k
