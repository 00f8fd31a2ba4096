"""TEST INFRASTRUCTURE. Explicit (hand-derived, no autograd) backward of the rendering hot path, stage by stage.

`oracle.sherf_oracle.gradients_from_fixture` gets the gradients by autograd through the forward restatement and is pinned
to the unmodified reference's backward (tests/golden/grad_*.npz).  This module restates the SAME gradients as explicit
formulas -- the arithmetic a backward kernel has to implement, loop for loop -- and tests/test_backward_math.py checks it
against autograd.  Nothing in the product imports it.

Stages, in backward order (names follow SURVEY.md section 8 rows):
  composite_bwd     a16  (d rgb_final, d acc)            -> d (rgb, sigma) per valid sample
  decoder_bwd       a14  d (rgb, sigma)                  -> d z (fused tokens 0/1) + decoder parameter gradients
  transformer_bwd   a13  d z                             -> d tokens_in + transformer parameter gradients
  fuse_bwd          a13  d tokens_in                     -> d tri-plane taps, d f2d, d f3d + conv1d_reprojection gradients
  taps_bwd      a10-a12  d taps                          -> d planes, d obs_feat, d voxel-level activations + conv1d_projection
  encoder_bwd       a11  d voxel-level activations       -> d vertex_feat + sparse conv / BatchNorm parameter gradients
"""
import math

import torch

from . import sherf_oracle as O

F32 = torch.float32


# ----------------------------------------------------------------------------------------------
# a16 compositing
# ----------------------------------------------------------------------------------------------
def composite_bwd(colors, sigma, t, rays_d, d_rgb, d_acc, white_back=False):
    """colors [R,S,3], sigma [R,S] (raw, -80 where masked), t [R,S], rays_d [R,3]; d_rgb [R,3], d_acc [R].
    -> d_colors [R,S,3], d_sigma [R,S].  (ray_marcher.py:25-64; the depth output carries no gradient.)"""
    delta = torch.cat([t[:, 1:] - t[:, :-1], torch.full_like(t[:, :1], 1e10)], 1) * torch.norm(rays_d, dim=-1)[:, None]
    sp = torch.relu(sigma)
    e = torch.exp(-(sp * delta))
    alpha = 1 - e
    fac = 1 - alpha + 1e-10
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), fac], 1), 1)[:, :-1]
    w = alpha * T
    g = 2.0 * d_rgb                                                   # rgb_final = 2 * (sum w c [+ 1 - acc]) - 1
    gw = (colors * g[:, None, :]).sum(-1) + d_acc[:, None] - (g.sum(-1)[:, None] if white_back else 0.0)
    gww = gw * w
    suffix = gww.flip(1).cumsum(1).flip(1) - gww                      # sum_{m>k} g_w[m] w_m
    d_alpha = gw * T - suffix / fac
    d_sigma = d_alpha * delta * e * (sigma > 0).to(F32)
    return g[:, None, :] * w[..., None], d_sigma


# ----------------------------------------------------------------------------------------------
# a14 NeRF decoder
# ----------------------------------------------------------------------------------------------
def decoder_bwd(state, pe_x, z, pe_v, d_rgb, d_sigma, prefix='decoder.'):
    """Recomputes the forward (triplane.py:285-316), then backpropagates.  Returns (d_z [n,3,32], {param: grad})."""
    W = lambda k: state[prefix + k + '.weight']
    B = lambda k: state[prefix + k + '.bias']
    x0 = torch.cat([pe_x, z[:, 0]], -1)
    ins, hs = [], []
    h = x0
    for i in range(8):
        ins.append(h)
        h = torch.relu(h @ W(f'pts_linears.{i}').t() + B(f'pts_linears.{i}'))
        hs.append(h)
        if i == 4:
            h = torch.cat([x0, h], -1)
    h7 = h
    f = h7 @ W('feature_linear').t() + B('feature_linear')
    vin = torch.cat([f, pe_v, z[:, 1]], -1)
    gpre = vin @ W('views_linear').t() + B('views_linear')
    g = torch.relu(gpre)
    s = torch.sigmoid(g @ W('rgb_linear').t() + B('rgb_linear'))
    grads = {}

    def lin_bwd(name, d_out, inp):
        grads[prefix + name + '.weight'] = d_out.t() @ inp
        grads[prefix + name + '.bias'] = d_out.sum(0)
        return d_out @ W(name)

    d_lin = d_rgb * (1 + 2 * 0.001) * s * (1 - s)
    d_g = lin_bwd('rgb_linear', d_lin, g)
    d_vin = lin_bwd('views_linear', d_g * (gpre > 0).to(F32), vin)
    d_z = torch.zeros_like(z)
    d_z[:, 1] = d_vin[:, 128 + 27:]
    d_h = lin_bwd('feature_linear', d_vin[:, :128], h7) + lin_bwd('alpha_linear', d_sigma[:, None], h7)
    d_x0 = torch.zeros_like(x0)
    for i in range(7, -1, -1):
        d_in = lin_bwd(f'pts_linears.{i}', d_h * (hs[i] > 0).to(F32), ins[i])
        if i == 5:                                                    # its input was cat([x0, h4])
            d_x0 += d_in[:, :x0.shape[1]]
            d_h = d_in[:, x0.shape[1]:]
        elif i == 0:
            d_x0 += d_in
        else:
            d_h = d_in
    d_z[:, 0] = d_x0[:, pe_x.shape[1]:]
    return d_z, grads


# ----------------------------------------------------------------------------------------------
# a13 transformer (one pre-norm layer: attention 3 heads x 16 over the 3 slot tokens, GELU feed-forward)
# ----------------------------------------------------------------------------------------------
def _ln_fwd(x, w, b):
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    inv = 1.0 / torch.sqrt((xc ** 2).mean(-1, keepdim=True) + 1e-5)
    xh = xc * inv
    return xh * w + b, (xh, inv)


def _ln_bwd(d_y, w, cache):
    """LayerNorm backward: d_x = inv * (d_xh - mean(d_xh) - xh * mean(d_xh * xh)); -> (d_x, d_w, d_b)."""
    xh, inv = cache
    d_xh = d_y * w
    d_x = inv * (d_xh - d_xh.mean(-1, keepdim=True) - xh * (d_xh * xh).mean(-1, keepdim=True))
    red = tuple(range(d_y.dim() - 1))
    return d_x, (d_y * xh).sum(red), d_y.sum(red)


def transformer_bwd(state, tok, d_out, prefix='renderer.transformer.layers.0.'):
    """renderer.py:949-993.  tok [n,3,32] (input tokens), d_out [n,3,32] -> (d_tok, {param: grad})."""
    p = prefix
    n = tok.shape[0]
    Wqkv, Wo, bo = state[p + '0.fn.fn.to_qkv.weight'], state[p + '0.fn.fn.to_out.0.weight'], state[p + '0.fn.fn.to_out.0.bias']
    W1, b1 = state[p + '1.fn.fn.net.0.weight'], state[p + '1.fn.fn.net.0.bias']
    W2, b2 = state[p + '1.fn.fn.net.3.weight'], state[p + '1.fn.fn.net.3.bias']
    # ---- forward with caches ----
    h0, c0 = _ln_fwd(tok, state[p + '0.fn.norm.weight'], state[p + '0.fn.norm.bias'])
    qkv = h0 @ Wqkv.t()
    q, k, v = [t_.view(n, 3, 3, 16).permute(0, 2, 1, 3) for t_ in qkv.chunk(3, -1)]          # [n,head,tok,16]
    att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (16 ** -0.5), -1)              # [n,head,tok,tok]
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(n, 3, 48)
    y = o @ Wo.t() + bo + tok
    h1, c1 = _ln_fwd(y, state[p + '1.fn.norm.weight'], state[p + '1.fn.norm.bias'])
    u = h1 @ W1.t() + b1
    cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2.0)))
    ge = u * cdf
    grads = {}
    # ---- feed-forward block: out = ge @ W2^T + b2 + y ----
    grads[p + '1.fn.fn.net.3.weight'] = torch.einsum('nto,nti->oi', d_out, ge)
    grads[p + '1.fn.fn.net.3.bias'] = d_out.sum((0, 1))
    d_ge = d_out @ W2
    d_u = d_ge * (cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi))             # d/du [u Phi(u)]
    grads[p + '1.fn.fn.net.0.weight'] = torch.einsum('nto,nti->oi', d_u, h1)
    grads[p + '1.fn.fn.net.0.bias'] = d_u.sum((0, 1))
    d_h1 = d_u @ W1
    d_y, grads[p + '1.fn.norm.weight'], grads[p + '1.fn.norm.bias'] = _ln_bwd(d_h1, state[p + '1.fn.norm.weight'], c1)
    d_y = d_y + d_out                                                                       # residual
    # ---- attention block: y = o @ Wo^T + bo + tok ----
    grads[p + '0.fn.fn.to_out.0.weight'] = torch.einsum('nto,nti->oi', d_y, o)
    grads[p + '0.fn.fn.to_out.0.bias'] = d_y.sum((0, 1))
    d_o = (d_y @ Wo).view(n, 3, 3, 16).permute(0, 2, 1, 3)                                  # [n,head,tok,16]
    d_att = torch.matmul(d_o, v.transpose(-1, -2))
    d_v = torch.matmul(att.transpose(-1, -2), d_o)
    d_s = att * (d_att - (d_att * att).sum(-1, keepdim=True)) * (16 ** -0.5)               # softmax backward, then the scale
    d_q = torch.matmul(d_s, k)
    d_k = torch.matmul(d_s.transpose(-1, -2), q)
    d_qkv = torch.cat([t_.permute(0, 2, 1, 3).reshape(n, 3, 48) for t_ in (d_q, d_k, d_v)], -1)
    grads[p + '0.fn.fn.to_qkv.weight'] = torch.einsum('nto,nti->oi', d_qkv, h0)
    d_h0 = d_qkv @ Wqkv
    d_tok, grads[p + '0.fn.norm.weight'], grads[p + '0.fn.norm.bias'] = _ln_bwd(d_h0, state[p + '0.fn.norm.weight'], c0)
    return d_tok + d_y, grads


# ----------------------------------------------------------------------------------------------
# a13 slot fusion (conv1d_reprojection 96 -> 32 per slot)
# ----------------------------------------------------------------------------------------------
def fuse_bwd(state, tri, f2d, f3d, d_tok, prefix='renderer.'):
    """-> (d_tri [3,n,32], d_f2d [n,96], d_f3d [n,96], {param: grad})  (renderer.py:423-424)."""
    W = state[prefix + 'conv1d_reprojection.weight'][:, :, 0]
    n = f2d.shape[0]
    comb = torch.cat([tri.permute(1, 0, 2), f2d.view(n, 3, 32), f3d.view(n, 3, 32)], -1)       # [n,3,96]
    grads = {prefix + 'conv1d_reprojection.weight': torch.einsum('nso,nsi->oi', d_tok, comb)[:, :, None],
             prefix + 'conv1d_reprojection.bias': d_tok.sum((0, 1))}
    d_comb = d_tok @ W
    return d_comb[..., :32].permute(1, 0, 2).contiguous(), d_comb[..., 32:64].reshape(n, 96), d_comb[..., 64:].reshape(n, 96), grads


# ----------------------------------------------------------------------------------------------
# a10-a12 feature taps: every tap is linear in the table, so its backward is the same stencil as a scatter-add
# ----------------------------------------------------------------------------------------------
def _bilinear_bwd(shape, px, py, d_out):
    """Transpose of sherf_oracle._bilinear: d_out [n,C] at pixel coords (px,py) -> d_img [C,H,W] (zeros padding)."""
    C, H, W = shape
    x0 = torch.floor(px); y0 = torch.floor(py)
    fx = px - x0; fy = py - y0
    d_img = torch.zeros(C, H * W)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = (x0 + dx).long(); yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            lin = yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)
            d_img.index_add_(1, lin, ((wx * wy * ok.to(F32))[:, None] * d_out).t())
    return d_img.view(C, H, W)


def _grid_sample_2d_bwd(shape, gx, gy, align_corners, d_out):
    C, H, W = shape
    if align_corners:
        px = (gx + 1) / 2 * (W - 1); py = (gy + 1) / 2 * (H - 1)
    else:
        px = ((gx + 1) * W - 1) / 2; py = ((gy + 1) * H - 1) / 2
    return _bilinear_bwd(shape, px, py, d_out)


def triplane_bwd(plane_shape, x_c, bounds, d_tri):
    """d_tri [3,n,32] -> d_planes [3,32,P,P]  (renderer.py:234-243)."""
    nrm = 2 * (x_c - bounds[0:1]) / (bounds[1:2] - bounds[0:1]) - 1
    sel = ((0, 1), (0, 2), (2, 1))
    return torch.stack([_grid_sample_2d_bwd(plane_shape[1:], nrm[:, a], nrm[:, b], False, d_tri[p]) for p, (a, b) in enumerate(sel)])


def pixel_aligned_bwd(feat_shape, img_hw, uv, d_f2d):
    """d_f2d [n,96] -> d_obs_feat [64,Hf,Wf]; the PE(rgb) third of f2d only reaches the (input) image: no gradient kept."""
    H, W = img_hw
    g = 2.0 * uv / torch.tensor([W, H], dtype=F32) - 1.0
    return _grid_sample_2d_bwd(feat_shape, g[:, 0], g[:, 1], True, d_f2d[:, :64])


def trilinear_sparse_bwd(keys, n_rows, shape, g, d_out):
    """Transpose of sherf_oracle.trilinear_sparse: d_out [n,C] -> d_feats [n_rows,C]."""
    D, H, W = shape
    px = (g[:, 0] + 1) / 2 * (W - 1); py = (g[:, 1] + 1) / 2 * (H - 1); pz = (g[:, 2] + 1) / 2 * (D - 1)
    x0, y0, z0 = torch.floor(px), torch.floor(py), torch.floor(pz)
    fx, fy, fz = px - x0, py - y0, pz - z0
    d_feats = torch.zeros(n_rows, d_out.shape[1])
    for dz, wz in ((0, 1 - fz), (1, fz)):
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi, yi, zi = (x0 + dx).long(), (y0 + dy).long(), (z0 + dz).long()
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                k = O._lin(zi, yi, xi, shape)
                pos = torch.searchsorted(keys, k).clamp(max=keys.numel() - 1)
                ok &= keys[pos] == k
                d_feats.index_add_(0, pos, (wx * wy * wz * ok.to(F32))[:, None] * d_out)
    return d_feats


def projection_bwd(state, f3d_raw, d_f3d, prefix='renderer.'):
    """conv1d_projection 192 -> 96 (renderer.py:350): -> (d_f3d_raw [n,192], {param: grad})."""
    Wp = state[prefix + 'conv1d_projection.weight'][:, :, 0]
    return d_f3d @ Wp, {prefix + 'conv1d_projection.weight': (d_f3d.t() @ f3d_raw)[:, :, None],
                        prefix + 'conv1d_projection.bias': d_f3d.sum(0)}


# ----------------------------------------------------------------------------------------------
# a11 sparse voxel encoder: forward with caches (same arithmetic as sherf_oracle.sparse_encoder), then backward
# ----------------------------------------------------------------------------------------------
def _bn_relu_fwd(raw, mult, n_rows, gamma, beta):
    """Training-mode BatchNorm1d(eps 1e-3) + ReLU over the reference's ROW set: `raw` holds the U rows that carry a conv
    result, the other n_rows - U rows are zeros (losing duplicates).  -> per-voxel output, cache."""
    mean = raw.sum(0) / n_rows
    var = (((raw - mean) ** 2).sum(0) + (n_rows - raw.shape[0]) * mean ** 2) / n_rows
    inv = 1.0 / torch.sqrt(var + 1e-3)
    xh = (raw - mean) * inv
    y = xh * gamma + beta
    xh0 = -mean * inv                                    # the zero rows
    y0 = xh0 * gamma + beta
    out = torch.relu(y) + (mult - 1).to(F32)[:, None] * torch.relu(y0)[None]
    return out, (xh, inv, y, xh0, y0, mult, n_rows)


def _bn_relu_bwd(d_out, gamma, cache):
    """-> (d_raw [U,C], d_gamma, d_beta).  Standard BatchNorm backward over all n_rows rows, where the n_rows - U zero rows
    share one value: only the SUM of their upstream gradients matters (d_y0)."""
    xh, inv, y, xh0, y0, mult, n_rows = cache
    d_y = d_out * (y > 0).to(F32)
    d_y0 = ((mult - 1).to(F32)[:, None] * d_out).sum(0) * (y0 > 0).to(F32)
    s1 = d_y.sum(0) + d_y0                               # sum over the row set of dL/dy
    s2 = (d_y * xh).sum(0) + d_y0 * xh0                  # sum of dL/dy * xhat
    d_raw = (gamma * inv / n_rows) * (n_rows * d_y - s1 - xh * s2)
    return d_raw, s2, s1


def encoder_forward_cached(state, feat, coord, out_sh, prefix='renderer.encoder_3d.'):
    """sherf_oracle.sparse_encoder (training mode) keeping what the backward needs.  -> (taps, cache list)."""
    sh = [int(v) for v in out_sh]
    c = coord.long()
    keys = O._lin(c[:, 1], c[:, 2], c[:, 3], sh)
    uk, inv = torch.unique(keys, sorted=True, return_inverse=True)
    mult = torch.bincount(inv, minlength=uk.numel())
    g = torch.zeros(uk.numel(), feat.shape[1]).index_add_(0, inv, feat)
    n_rows = feat.shape[0]
    taps, cache = [], [('input', inv, uk, list(sh), g)]
    for name, kind, nconv in O._ENC_LAYERS:
        if name == 'TAP':
            taps.append((uk.clone(), g.clone(), list(sh)))
            cache.append(('tap',))
            continue
        z = uk // (sh[1] * sh[2]); y = (uk // sh[2]) % sh[1]; x = uk % sh[2]
        if kind == 'subm':
            for ci in range(nconv):
                wname, bname = f'{prefix}{name}.{3 * ci}', f'{prefix}{name}.{3 * ci + 1}'
                W = state[wname + '.weight']                                   # [out,3,3,3,in]
                raw = torch.zeros(uk.numel(), W.shape[0])
                pairs = []                                                     # (tap, output rows, input rows)
                for kk in range(27):
                    kz, ky, kx = kk // 9, (kk // 3) % 3, kk % 3
                    qz, qy, qx = z + kz - 1, y + ky - 1, x + kx - 1
                    ok = (qz >= 0) & (qz < sh[0]) & (qy >= 0) & (qy < sh[1]) & (qx >= 0) & (qx < sh[2])
                    qk = O._lin(qz, qy, qx, sh)
                    pos = torch.searchsorted(uk, qk).clamp(max=uk.numel() - 1)
                    ok &= uk[pos] == qk
                    o = torch.nonzero(ok)[:, 0]
                    if o.numel():
                        raw[o] += g[pos[o]] @ W[:, kz, ky, kx, :].t()
                        pairs.append((kk, o, pos[o]))
                g_in = g
                g, bnc = _bn_relu_fwd(raw, mult, n_rows, state[bname + '.weight'], state[bname + '.bias'])
                cache.append(('conv', wname, bname, pairs, g_in, bnc,
                              dict(keys_in=uk, sh_in=list(sh), keys_out=uk, sh_out=list(sh), down=False, raw=raw)))
        else:
            wname, bname = f'{prefix}{name}.0', f'{prefix}{name}.1'
            W = state[wname + '.weight']
            osh = [(d - 1) // 2 + 1 for d in sh]
            key_all, src_all, k_all = [], [], []
            for kk in range(27):
                kz, ky, kx = kk // 9, (kk // 3) % 3, kk % 3
                nz, ny, nx = z + 1 - kz, y + 1 - ky, x + 1 - kx
                ok = (nz % 2 == 0) & (ny % 2 == 0) & (nx % 2 == 0)
                oz, oy, ox = nz // 2, ny // 2, nx // 2
                ok &= (oz >= 0) & (oz < osh[0]) & (oy >= 0) & (oy < osh[1]) & (ox >= 0) & (ox < osh[2])
                o = torch.nonzero(ok)[:, 0]
                key_all.append(O._lin(oz, oy, ox, osh)[o]); src_all.append(o); k_all.append(torch.full_like(o, kk))
            key_all = torch.cat(key_all); src_all = torch.cat(src_all); k_all = torch.cat(k_all)
            nuk, ninv = torch.unique(key_all, sorted=True, return_inverse=True)
            raw = torch.zeros(nuk.numel(), W.shape[0])
            Wf = W.reshape(W.shape[0], 27, W.shape[4])
            pairs = []
            for kk in range(27):
                m = k_all == kk
                if m.any():
                    raw.index_add_(0, ninv[m], g[src_all[m]] @ Wf[:, kk, :].t())
                    pairs.append((kk, ninv[m], src_all[m]))
            g_in = g
            meta = dict(keys_in=uk, sh_in=list(sh), keys_out=nuk, sh_out=list(osh), down=True, raw=raw)
            uk, sh = nuk, osh
            mult = torch.ones(uk.numel(), dtype=torch.long)
            n_rows = uk.numel()
            g, bnc = _bn_relu_fwd(raw, mult, n_rows, state[bname + '.weight'], state[bname + '.bias'])
            cache.append(('conv', wname, bname, pairs, g_in, bnc, meta))
    return taps, cache


def encoder_bwd(state, cache, d_taps):
    """d_taps: gradients of the three tapped levels' activations (in tap order).  -> (d_vertex_feat [N,32], {param: grad}).
    Per conv layer: BatchNorm+ReLU backward over the row set, then for every tap's (output row, input row) pairs
    d_in[input] += d_raw[output] @ W_tap and dW_tap += d_raw[output]^T in[input]."""
    grads = {}
    d_g = None
    tap_i = len(d_taps) - 1
    for entry in reversed(cache):
        if entry[0] == 'tap':
            d_g = d_taps[tap_i] if d_g is None else d_g + d_taps[tap_i]
            tap_i -= 1
        elif entry[0] == 'conv':
            _, wname, bname, pairs, g_in, bnc = entry[:6]
            W = state[wname + '.weight']                                       # [out,3,3,3,in]
            d_raw, grads[bname + '.weight'], grads[bname + '.bias'] = _bn_relu_bwd(d_g, state[bname + '.weight'], bnc)
            dW = torch.zeros_like(W).reshape(W.shape[0], 27, W.shape[4])
            d_in = torch.zeros_like(g_in)
            Wf = W.reshape(W.shape[0], 27, W.shape[4])
            for kk, o, src in pairs:
                d_in.index_add_(0, src, d_raw[o] @ Wf[:, kk, :])
                dW[:, kk, :] = d_raw[o].t() @ g_in[src]
            grads[wname + '.weight'] = dW.reshape(W.shape)
            d_g = d_in
        else:                                                                   # level-0 aggregation of duplicate rows
            return d_g[entry[1]], grads
    raise AssertionError('cache without input entry')


# ----------------------------------------------------------------------------------------------
# the whole chain
# ----------------------------------------------------------------------------------------------
def backward_from_fixture(fx, state):
    """Forward by the restatement (training mode), then every stage's explicit backward in order, under the stub loss of
    BASELINE config 5 (sherf_oracle.stub_loss).  -> (loss, {name: grad}) with the same keys as
    sherf_oracle.gradients_from_fixture: every renderer / decoder parameter + 'input.planes|obs_feat|vertex_feat'."""
    import numpy as np
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
        R, S = r['t'].shape
        loss = O.stub_loss(r['rgb'], r['acc'])
        rs = np.random.RandomState(11)
        t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
        t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
        d_rgb_img, d_acc = 2.0 * (r['rgb'] - t_rgb) / (R * 3), 2.0 * (r['acc'] - t_acc) / R
        valid = r['valid']
        col = torch.zeros(R * S, 3); sig = torch.full((R * S,), -80.0)
        col[valid] = r['sample_rgb']; sig[valid] = r['sample_sigma']
        ray_d = torch.from_numpy(np.ascontiguousarray(fx['input_data']['ray_d_all'][0, 0]))
        d_col, d_sig = composite_bwd(col.view(R, S, 3), sig.view(R, S), r['t'], ray_d, d_rgb_img, d_acc,
                                     bool(fx['options'].get('white_back', False)))
        grads = {}
        pe_x, pe_v = O.positional_encoding(r['x_c'], 6), O.positional_encoding(r['v_c'], 4)
        d_z, g = decoder_bwd(state, pe_x, r['tokens_out'], pe_v, d_col.reshape(-1, 3)[valid], d_sig.reshape(-1)[valid]); grads.update(g)
        d_tok, g = transformer_bwd(state, r['tokens_in'], d_z); grads.update(g)
        bounds = torch.from_numpy(np.ascontiguousarray(fx['input_data']['t_world_bounds'])).view(2, 3)
        planes = torch.from_numpy(np.ascontiguousarray(fx['planes']))[0]
        tri = O.triplane_features(planes, r['x_c'], bounds)
        d_tri, d_f2d, d_f3d, g = fuse_bwd(state, tri, r['f2d'], r['f3d'], d_tok); grads.update(g)
        grads['input.planes'] = triplane_bwd(planes.shape, r['x_c'], bounds, d_tri)[None]
        H, W = fx['input_data']['obs_img_all'].shape[-2:]
        grads['input.obs_feat'] = pixel_aligned_bwd(fx['obs_feat'].shape[1:], (H, W), r['uv'], d_f2d)[None]
        d_raw, g = projection_bwd(state, r['f3d_raw'], d_f3d); grads.update(g)
        sp = r['sp_input']
        taps, cache = encoder_forward_cached(state, torch.from_numpy(np.ascontiguousarray(fx['vertex_feat'])), sp['coord'], sp['out_sh'])
        d_taps, off = [], 0
        for keys, feats, shape in taps:
            C = feats.shape[1]
            d_taps.append(trilinear_sparse_bwd(keys, feats.shape[0], shape, r['grid'], d_raw[:, off:off + C]))
            off += C
        grads['input.vertex_feat'], g = encoder_bwd(state, cache, d_taps); grads.update(g)
    return float(loss), grads


# ----------------------------------------------------------------------------------------------
# the same tap / fusion backward in the FOLDED formulation the HIP path uses (sherf_amd/renderer.py: _weights):
# forward  tokens[n][s] = tok_bias_s + taps(planes_f[s]) + taps(feat_f[:, 32s:32s+32]) (s < 2) + taps(rows_fold_l[:, 32s:32s+32])
#          planes_f[p][texel] = Wa planes[p][:, texel],  feat_f[texel][32s:] = Wb obs_feat[32s:32s+32, texel],
#          rows_fold_l[row][32s:] = (Wc Wp[32s:32s+32, cols_l]) act_l[row],  tok_bias_s = br + Wc bp[32s:32s+32]
#          (+ slot 2 gets Wb PE(rgb)[:32] inside the MLP kernel)
# backward = (i) ONE scatter of d_tokens with the forward's tap weights into d_planes_f / d_feat_f / d_rows_fold_l,
#            (ii) small dense "unfold" products per texel / row.
# ----------------------------------------------------------------------------------------------
def folded_taps_bwd(state, fx_planes, fx_obs_feat, img_hw, r, d_tok, prefix='renderer.'):
    """r: forward intermediates of sherf_oracle.render (x_c, uv, grid, taps, tap_rgb); d_tok [n,3,32].
    -> dict(d_planes, d_obs_feat, d_levels[3], grads{conv1d_reprojection.*, conv1d_projection.*})."""
    Wr = state[prefix + 'conv1d_reprojection.weight'][:, :, 0]
    Wp = state[prefix + 'conv1d_projection.weight'][:, :, 0]
    bp = state[prefix + 'conv1d_projection.bias']
    Wa, Wb, Wc = Wr[:, 0:32], Wr[:, 32:64], Wr[:, 64:96]
    n = d_tok.shape[0]
    # ---- (i) scatter with the forward's stencils ----
    d_planes_f = triplane_bwd((3, 32) + tuple(fx_planes.shape[-2:]), r['x_c'], r['_bounds'], d_tok.permute(1, 0, 2))   # [3,32,P,P]
    H, W = img_hw
    g = 2.0 * r['uv'] / torch.tensor([W, H], dtype=F32) - 1.0
    d_feat_f = _grid_sample_2d_bwd((64,) + tuple(fx_obs_feat.shape[-2:]), g[:, 0], g[:, 1], True, d_tok[:, :2].reshape(n, 64))
    d_rows = [trilinear_sparse_bwd(keys, feats.shape[0], shape, r['grid'], d_tok.reshape(n, 96)) for keys, feats, shape in r['taps']]
    d_bias = d_tok.sum(0)                                                                    # [3,32]
    # ---- (ii) unfold ----
    out, dWa, dWb, dWc = {}, torch.zeros(32, 32), torch.zeros(32, 32), torch.zeros(32, 32)
    out['d_planes'] = torch.einsum('oi,pohw->pihw', Wa, d_planes_f)
    dWa = torch.einsum('pohw,pihw->oi', d_planes_f, fx_planes)
    df = d_feat_f.view(2, 32, *d_feat_f.shape[-2:])
    out['d_obs_feat'] = torch.einsum('oi,sohw->sihw', Wb, df).reshape(64, *d_feat_f.shape[-2:])
    dWb = torch.einsum('sohw,sihw->oi', df, fx_obs_feat.view(2, 32, *fx_obs_feat.shape[-2:]))
    dWb = dWb + d_tok[:, 2].t() @ O.positional_encoding(r['tap_rgb'], 5)[:, :32]           # slot 2: PE(rgb) @ Wb^T
    dWp = torch.zeros_like(Wp)
    cols = ((0, 32), (32, 96), (96, 192))
    out['d_levels'] = []
    for (c0, c1), d_row, (keys, act, shape) in zip(cols, d_rows, r['taps']):
        d_act = torch.zeros_like(act)
        for s in range(3):
            F_ls = Wc @ Wp[32 * s:32 * s + 32, c0:c1]                                        # [32, C_l]
            d_blk = d_row[:, 32 * s:32 * s + 32]                                             # [rows, 32]
            d_act += d_blk @ F_ls
            G = d_blk.t() @ act                                                              # dF_ls [32, C_l]
            dWc += G @ Wp[32 * s:32 * s + 32, c0:c1].t()
            dWp[32 * s:32 * s + 32, c0:c1] += Wc.t() @ G
        out['d_levels'].append(d_act)
    d_bp = torch.cat([Wc.t() @ d_bias[s] for s in range(3)])
    for s in range(3):
        dWc += torch.outer(d_bias[s], bp[32 * s:32 * s + 32])
    out['grads'] = {prefix + 'conv1d_reprojection.weight': torch.cat([dWa, dWb, dWc], 1)[:, :, None],
                    prefix + 'conv1d_reprojection.bias': d_bias.sum(0),
                    prefix + 'conv1d_projection.weight': dWp[:, :, None], prefix + 'conv1d_projection.bias': d_bp}
    return out
