"""CPU check of the backward ORCHESTRATION of the dense stage (sherf_amd/backward_dense.py) against autograd through the
oracle, with the C entry points replaced by their torch emulation (tests/bwd_emulator.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, sherf_oracle as O
from sherf_amd.backward_dense import Mat, dense_backward
from tests.bwd_emulator import EmuOps


@pytest.fixture(scope='module')
def state(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    return {n: torch.from_numpy(fixtures.seeded_param(n, s)) for n, s in shapes.items() if fixtures.seeded_param(n, s) is not None}


def _close(a, b, tol):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-14


@pytest.mark.parametrize('cfg', ['tiny_nv', 'tiny'])
def test_dense_backward_orchestration(cfg, state):
    fx = fixtures.renderer_inputs(cfg)
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
        n = r['x_c'].shape[0]
        # what the forward kernels hand over: gather tokens WITHOUT the slot-2 rgb encoding, and the extras rows
        Wb = state['renderer.conv1d_reprojection.weight'][:, 32:64, 0]
        tok = r['tokens_in'].clone()
        tok[:, 2] -= O.positional_encoding(r['tap_rgb'], 5)[:, :32] @ Wb.t()
        ext = torch.zeros(n, 12)
        ext[:, 0:3], ext[:, 3:6], ext[:, 6:9] = r['x_c'], r['v_c'], r['tap_rgb']
        d_sample = torch.cat([g['stage.sample_rgb'], g['stage.sample_sigma'][:, None]], 1).contiguous()
        d_tin, grads, dWb_pe = dense_backward(EmuOps(), state, Mat(tok.reshape(-1).clone(), n, 96), Mat(ext.reshape(-1).clone(), n, 12),
                                              Mat(d_sample.reshape(-1).clone(), n, 4))
    assert _close(d_tin.tensor().view(n, 3, 32), g['stage.tokens_in'], 5e-4)
    want = [k for k in g if k.startswith('decoder.') or k.startswith('renderer.transformer.')]
    assert set(want) == set(grads), set(want) ^ set(grads)
    for k in want:
        assert grads[k].shape == g[k].shape, k
        assert _close(grads[k], g[k], 1e-3), (k, float((grads[k] - g[k]).abs().max()), float(g[k].abs().max()))
    assert _close(dWb_pe, g['stage.tokens_in'][:, 2].t() @ O.positional_encoding(r['tap_rgb'], 5)[:, :32], 1e-4)


@pytest.mark.parametrize('cfg', ['tiny_nv', 'tiny'])
def test_taps_backward_orchestration(cfg, state):
    """sherf_amd/backward_taps.py on the emulated entry points; the scatter kernel (sherf_gather_tokens_bwd) is stood in for
    by the oracle's stencils writing the kernel's folded, channel-last layouts."""
    from oracle import backward_explicit as BX
    from sherf_amd.backward_taps import taps_backward
    fx = fixtures.renderer_inputs(cfg)
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
        n = r['x_c'].shape[0]
        planes = torch.from_numpy(fx['planes'])[0]; obs_feat = torch.from_numpy(fx['obs_feat'])[0]
        P, (Hf, Wf) = planes.shape[-1], obs_feat.shape[-2:]
        H, W = fx['input_data']['obs_img_all'].shape[-2:]
        bounds = torch.from_numpy(fx['input_data']['t_world_bounds']).view(2, 3)
        taps, cache = BX.encoder_forward_cached(state, torch.from_numpy(fx['vertex_feat']), r['sp_input']['coord'], r['sp_input']['out_sh'])
        conv_entries = [e for e in cache if e[0] == 'conv']
        tap_layers = [conv_entries[i] for i in (4, 8, 12)]                 # conv1[1], conv2[2], conv3[2]: the tapped layers
        levels = []
        for (keys, feats, shape), ent in zip(taps, tap_layers):
            xh, inv, y, xh0, y0, mult, n_rows = ent[5]
            gamma, beta = state[ent[2] + '.weight'], state[ent[2] + '.bias']
            rows, C = feats.shape
            cap = rows + 5                                                  # padded capacity, like the product's buffers
            # raw and the (scale, shift) the forward's consumers use: y = raw * scale + shift
            scale = gamma * inv
            raw = (xh / inv + (-(xh0 / inv)))                               # raw = xh / inv + mean, mean = -xh0 / inv
            shift = beta - (-(xh0 / inv)) * scale
            rawp = torch.zeros(cap, C); rawp[:rows] = raw
            levels.append(dict(raw=Mat(rawp.reshape(-1), cap, C), bnparam=Mat(torch.cat([scale, shift, torch.relu(shift)]).clone(), 1, 3 * C),
                               n_rows=torch.tensor(rows), cap=cap, C=C, keys=keys, shape=shape))
            assert _close(torch.relu(raw * scale + shift), feats, 1e-5)     # mult == 1 above level 0: feats are the activations

        def scatter(d_tiled, d_planes_f, d_feat_f, d_rows, d_bias):         # stands in for sherf_gather_tokens_bwd
            tiles = (n + 31) // 32
            d_tok = d_tiled.view(tiles, 3, 8, 32, 4).permute(0, 3, 1, 2, 4).reshape(tiles * 32, 3, 32)[:n]
            dpf = BX.triplane_bwd((3, 32, P, P), r['x_c'], bounds, d_tok.permute(1, 0, 2))              # [3,32,P,P]
            d_planes_f.tensor().copy_(dpf.permute(0, 2, 3, 1).reshape(3 * P * P, 32))
            gg = 2.0 * r['uv'] / torch.tensor([W, H], dtype=torch.float32) - 1.0
            dff = BX._grid_sample_2d_bwd((64, Hf, Wf), gg[:, 0], gg[:, 1], True, d_tok[:, :2].reshape(n, 64))
            d_feat_f.tensor().copy_(dff.permute(1, 2, 0).reshape(Hf * Wf, 64))
            for lv, dr in zip(levels, d_rows):
                rows = int(lv['n_rows'])
                dr.tensor()[:rows].copy_(BX.trilinear_sparse_bwd(lv['keys'], rows, lv['shape'], r['grid'], d_tok.reshape(n, 96)))
            d_bias.tensor().copy_(d_tok.sum(0).reshape(1, 96))

        ctx = dict(n=n, P=P, Hf=Hf, Wf=Wf, planes=Mat(planes.reshape(-1).clone(), 96, P * P), obs_feat=Mat(obs_feat.reshape(-1).clone(), 64, Hf * Wf),
                   levels=levels, scatter=scatter)
        d_tin = Mat(g['stage.tokens_in'].reshape(-1).clone(), n, 96)
        dWb_pe = g['stage.tokens_in'][:, 2].t() @ O.positional_encoding(r['tap_rgb'], 5)[:, :32]
        out = taps_backward(EmuOps(), state, ctx, d_tin, dWb_pe)
    assert _close(out['d_planes'].tensor().view(3, 32, P, P), g['input.planes'][0], 2e-4)
    assert _close(out['d_obs_feat'].tensor().view(64, Hf, Wf), g['input.obs_feat'][0], 2e-4)
    for i, lv in enumerate(levels):
        assert _close(out['d_levels'][i].tensor()[:int(lv['n_rows'])], g[f'stage.level{i}'], 5e-4), i
    for k, v in out['grads'].items():
        assert v.shape == g[k].shape and _close(v, g[k], 5e-4), k


@pytest.mark.parametrize('cfg', ['tiny_nv', 'tiny'])
def test_encoder_backward_orchestration(cfg, state):
    """sherf_amd/backward_encoder.py on the emulated entry points: every sparse-conv / BatchNorm parameter gradient and the
    per-vertex feature gradient against autograd through the oracle."""
    from oracle import backward_explicit as BX
    from sherf_amd.backward_encoder import encoder_backward
    fx = fixtures.renderer_inputs(cfg)
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
        coord = r['sp_input']['coord']
        N = coord.shape[0]
        taps, cache = BX.encoder_forward_cached(state, torch.from_numpy(fx['vertex_feat']), coord, r['sp_input']['out_sh'])
        _, inv0, uk0, sh0, g0 = cache[0]
        convs = [e for e in cache if e[0] == 'conv']
        # levels: 0 = input voxels, then one per strided conv
        lev_keys, lev_dims = [uk0], [tuple(sh0)]
        for e in convs:
            if e[6]['down']:
                lev_keys.append(e[6]['keys_out']); lev_dims.append(tuple(e[6]['sh_out']))
        pad = 3
        levels = [dict(keys=torch.cat([k, torch.zeros(pad, dtype=k.dtype)]), n_rows=torch.tensor(k.numel()), dims=d, cap=k.numel() + pad)
                  for k, d in zip(lev_keys, lev_dims)]
        mult = torch.cat([torch.bincount(inv0, minlength=uk0.numel()), torch.ones(pad, dtype=torch.long)])
        layers, lev = [], 0
        tap_after = {4, 8, 12}
        for i, e in enumerate(convs):
            _, wname, bname, pairs, g_in, bnc, meta = e
            xh, inv, y, xh0, y0, m_, n_rows = bnc
            lev_out = lev + 1 if meta['down'] else lev
            gamma, beta = state[bname + '.weight'], state[bname + '.bias']
            mean, C = -(xh0 / inv), xh.shape[1]
            scale = gamma * inv
            shift = beta - mean * scale
            cap = levels[lev_out]['cap']
            rawp = torch.zeros(cap, C); rawp[:meta['raw'].shape[0]] = meta['raw']
            layers.append(dict(wname=wname, bname=bname, cin=g_in.shape[1], cout=C, down=meta['down'], tap=i in tap_after, lev_in=lev, lev_out=lev_out,
                               raw=Mat(rawp.reshape(-1), cap, C), bnparam=Mat(torch.cat([scale, shift, torch.relu(shift)]).clone(), 1, 3 * C),
                               stats=Mat(torch.cat([mean, 1.0 / inv ** 2 - 1e-3]).clone(), 1, 2 * C)))
            lev = lev_out
        g0p = torch.zeros(levels[0]['cap'], 32); g0p[:g0.shape[0]] = g0
        ctx = dict(levels=levels, mult=mult, n_total=torch.tensor(N), coord=coord, N=N, g0=Mat(g0p.reshape(-1), levels[0]['cap'], 32), layers=layers)
        d_levels = []
        for i, (keys, feats, shape) in enumerate(taps):
            cap = levels[i + 1]['cap']
            d = torch.zeros(cap, feats.shape[1]); d[:feats.shape[0]] = g[f'stage.level{i}']
            d_levels.append(Mat(d.reshape(-1), cap, feats.shape[1]))
        d_feat, grads = encoder_backward(EmuOps(), state, ctx, d_levels)
    assert _close(d_feat.tensor(), g['input.vertex_feat'], 1e-3)
    want = [k for k in g if k.startswith('renderer.encoder_3d.')]
    assert set(want) == set(grads)
    for k in want:
        assert grads[k].shape == g[k].shape and _close(grads[k], g[k], 3e-3), (k, float((grads[k] - g[k]).abs().max()), float(g[k].abs().max()))
