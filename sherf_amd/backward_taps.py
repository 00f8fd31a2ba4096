"""Backward of the feature taps and of the slot fusion (a10-a13) in the FOLDED formulation of the forward.

forward  tokens[n][s] = tok_bias_s + taps(planes_f[s]) + taps(feat_f[:, 32s:32s+32]) (s < 2) + taps(rows_fold_l[:, 32s:32s+32])
backward (i)  one scatter of d_tokens with the forward's tap weights (sherf_gather_tokens_bwd, csrc/gather.hip)
         (ii) unfold: per-texel / per-row products back through the 32x32 slot projections (sherf_bwd_unfold32, GEMMs)
         (iii) parameter-sized algebra for conv1d_reprojection / conv1d_projection (a few [96, C] matrices, torch on device)
Mirrors oracle/backward_explicit.py: folded_taps_bwd; orchestration checked on the CPU in tests/test_backward_dense.py.
"""
import torch

from .backward_dense import Mat


def taps_backward(ops, state, ctx, d_tin, dWb_pe):
    """d_tin: Mat [n,96] = dL/d tokens_in (from dense_backward); dWb_pe [32,32] its slot-2 rgb-encoding contribution.
    ctx: what the forward left behind --
        n, P, Hf, Wf, planes [3*32, P*P] Mat (NCHW source), obs_feat [64, Hf*Wf] Mat,
        levels: list of 3 dicts(raw Mat [cap,C], bnparam Mat [1,3C], n_rows (device int32 tensor), cap, C),
        scatter(d_tokens_tiled, d_planes_f, d_feat_f, d_rows[3], d_tok_bias): runs sherf_gather_tokens_bwd on the frame's
        geometry (bound by the caller to the forward's workspace).
    Returns dict(d_planes Mat [96, P*P], d_obs_feat Mat [64, Hf*Wf], d_levels [3 Mats [cap,C]], grads {...})."""
    n, P, Hf, Wf = ctx['n'], ctx['P'], ctx['Hf'], ctx['Wf']
    dev = d_tin.buf.device
    Z = lambda r, c: Mat.zeros(r, c, dev)
    E = lambda r, c: Mat.empty(r, c, dev)                           # written in full by the next kernel
    Wr = state['renderer.conv1d_reprojection.weight'].detach().float()[:, :, 0]
    Wp = state['renderer.conv1d_projection.weight'].detach().float()[:, :, 0]
    bp = state['renderer.conv1d_projection.bias'].detach().float()
    Wa, Wb, Wc = Wr[:, 0:32].contiguous(), Wr[:, 32:64].contiguous(), Wr[:, 64:96].contiguous()
    # ---- (i) scatter ----
    tiles = (n + 31) // 32
    d_tiled = Mat.empty(1, tiles * 3 * 8 * 32 * 4, dev).buf            # (sherf_bwd_tile_tokens writes every padded tile)
    ops.tile_tokens(d_tin, n, d_tiled)
    d_planes_f, d_feat_f, d_bias = Z(3 * P * P, 32), Z(Hf * Wf, 64), Z(1, 96)
    d_rows = [Z(lv['cap'], 96) for lv in ctx['levels']]
    ctx['scatter'](d_tiled, d_planes_f, d_feat_f, d_rows, d_bias)
    # ---- (ii) unfold ----
    d_planes, dWa = Z(96, P * P), Z(32, 32)
    ops.unfold32(d_planes_f, Mat.of(Wa), ctx['planes'], P * P, 3, 32, P * P * 32, d_planes, dWa)
    d_obs, dWb = Z(64, Hf * Wf), Z(32, 32)
    ops.unfold32(d_feat_f, Mat.of(Wb), ctx['obs_feat'], Hf * Wf, 2, 64, 32, d_obs, dWb)
    cols = ((0, 32), (32, 96), (96, 192))
    dWc = torch.zeros(32, 32, device=dev)
    dWp = torch.zeros_like(Wp)
    d_levels = []
    for (c0, c1), lv, d_row in zip(cols, ctx['levels'], d_rows):
        C = lv['C']
        Fcat = torch.cat([Wc @ Wp[32 * s:32 * s + 32, c0:c1] for s in range(3)], 0).contiguous()      # [96, C]
        act = E(lv['cap'], C)
        ops.bn_relu_apply(lv['raw'], lv['bnparam'], lv['n_rows'], act)
        d_act, G = E(lv['cap'], C), E(96, C)
        ops.gemm(0, 0, d_row, Mat.of(Fcat), d_act)
        ops.gemm(1, 0, d_row, act, G)
        d_levels.append(d_act)
        # ---- (iii) F_ls = Wc Wp[32s:32s+32, cols]:  dWc += G_ls Wp_ls^T,  dWp_ls += Wc^T G_ls ----
        Gt = G.tensor()
        for s in range(3):
            dWc += Gt[32 * s:32 * s + 32] @ Wp[32 * s:32 * s + 32, c0:c1].t()
            dWp[32 * s:32 * s + 32, c0:c1] += Wc.t() @ Gt[32 * s:32 * s + 32]
    db = d_bias.tensor().view(3, 32)                                   # tok_bias_s = br + Wc bp[32s:32s+32]
    for s in range(3):
        dWc += torch.outer(db[s], bp[32 * s:32 * s + 32])
    d_bp = torch.cat([Wc.t() @ db[s] for s in range(3)])
    grads = {'renderer.conv1d_reprojection.weight': torch.cat([dWa.tensor(), dWb.tensor() + dWb_pe, dWc], 1)[:, :, None].clone(),
             'renderer.conv1d_reprojection.bias': db.sum(0).clone(),
             'renderer.conv1d_projection.weight': dWp[:, :, None].clone(), 'renderer.conv1d_projection.bias': d_bp}
    return dict(d_planes=d_planes, d_obs_feat=d_obs, d_levels=d_levels, grads=grads)
