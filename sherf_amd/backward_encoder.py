"""Backward of the sparse voxel encoder (a11); mirrors oracle/backward_explicit.py: encoder_bwd.

Per conv layer, last to first:  BatchNorm+ReLU backward over the reference's row set (sherf_bwd_bn_relu), weight gradient
over the forward's neighbour pairs (sherf_bwd_conv_wgrad), input gradient (sherf_bwd_conv_dgrad); the three tapped levels
inject the gradients coming from the feature taps; the level-0 aggregation sends every input row its voxel's gradient.
Orchestration checked on the CPU through the emulated entry points (tests/test_backward_dense.py)."""
from .backward_dense import Mat


def encoder_backward(ops, state, ctx, d_levels):
    """ctx: levels [4 dicts(keys, wp, n_rows, dims (D,H,W), cap)], mult, n_total, coord, N, g0 Mat [cap0, 32],
            layers [dicts(wname, bname, cin, cout, down, tap, lev_in, lev_out, raw Mat, bnparam Mat [1,3C], stats Mat [1,2C])].
    d_levels: the three Mats [cap_l, C_l] from taps_backward.  -> (d_vertex_feat Mat [N,32], grads {name: tensor})."""
    dev = ctx['g0'].buf.device
    Z = lambda r, c: Mat.zeros(r, c, dev)
    E = lambda r, c: Mat.empty(r, c, dev)            # written in full (rows < n_rows; the rest is never read) before any read
    grads = {}
    layers, levels = ctx['layers'], ctx['levels']
    d_g, tap_i = None, 2
    for li in range(len(layers) - 1, -1, -1):
        ly = layers[li]
        lo, lin = levels[ly['lev_out']], levels[ly['lev_in']]
        C, Cin = ly['cout'], ly['cin']
        if ly['tap']:
            if d_g is None:
                d_g = d_levels[tap_i]
            else:
                ops.copy2d(d_g, d_levels[tap_i], add=True)
            tap_i -= 1
        at_l0 = ly['lev_out'] == 0
        d_raw, dgam, dbet = E(lo['cap'], C), E(1, C), E(1, C)       # (sherf_bwd_bn_relu zeroes d_raw's rows >= n_rows itself)
        ops.bn_relu_bwd(d_g, ly['raw'], ly['bnparam'], ly['stats'], Mat.of(state[ly['bname'] + '.weight']),
                        ctx['mult'] if at_l0 else None, ctx['n_total'] if at_l0 else lo['n_rows'], lo['n_rows'], d_raw, dgam, dbet)
        grads[ly['bname'] + '.weight'], grads[ly['bname'] + '.bias'] = dgam.tensor().view(-1).clone(), dbet.tensor().view(-1).clone()
        prev = layers[li - 1] if li > 0 else None
        in_raw = prev['raw'] if prev else ctx['g0']
        in_bn = prev['bnparam'] if prev else None
        in_mult = ctx['mult'] if (prev is not None and ly['lev_in'] == 0) else None
        W = state[ly['wname'] + '.weight'].detach().float().contiguous()          # [out,3,3,3,in]
        Wm = Mat(W.view(-1), C, 27 * Cin)
        dW = Z(C, 27 * Cin)
        ops.conv_wgrad(lo, lin, in_raw, Cin, in_bn, in_mult, d_raw, C, int(ly['down']), dW)
        grads[ly['wname'] + '.weight'] = dW.tensor().view(W.shape).clone()
        d_in = Z(lin['cap'], Cin)
        ops.conv_dgrad(lin, lo, d_raw, C, Wm, Cin, int(ly['down']), d_in)
        d_g = d_in
    d_feat = Z(ctx['N'], 32)
    ops.gather_rows(ctx['coord'], ctx['N'], levels[0], d_g, 32, d_feat)
    return d_feat, grads
