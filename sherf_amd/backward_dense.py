"""Backward of the per-sample dense stage (a13 + a14: slot-2 rgb encoding, transformer, NeRF decoder).

The forward of the valid samples is recomputed in fp32, row-major [n, C], as a sequence of MFMA GEMMs (`sherf_bwd_gemm[_bias_act]`,
csrc/bwd_gemm.hip: hand-written, three-part bf16 operand split, no vendor library) and small element-wise HIP kernels, every
activation kept in HBM (~6 KB per valid sample; 288 GB of HBM make that a non-issue), then back-propagated layer by layer.  It
mirrors oracle/backward_explicit.py (decoder_bwd, transformer_bwd) line for line; the orchestration below is checked on the CPU
against a torch emulation of the C entry points (tests/bwd_emulator.py, tests/test_backward_dense.py), the kernels from their source
on the CPU and on the MI355X (tests/test_hipcpu_kernels.py, tests/test_gpu_backward.py).

The fused MFMA forward kernel stays the forward of record; a fused backward of the decoder would replace the recompute.
"""
import ctypes

import torch

from . import _lib


class Mat:
    """Row-major fp32 matrix view [rows, cols] with leading dimension `ld` inside a flat buffer (element offset `off`)."""

    def __init__(self, buf, rows, cols, ld=None, off=0):
        self.buf, self.rows, self.cols, self.ld, self.off = buf, int(rows), int(cols), int(ld if ld is not None else cols), int(off)
        assert self.ld >= self.cols and self.off + (self.rows - 1) * self.ld + self.cols <= buf.numel(), 'Mat out of bounds'

    @staticmethod
    def zeros(rows, cols, device):
        return Mat(torch.zeros(int(rows) * int(cols), dtype=torch.float32, device=device), rows, cols)

    @staticmethod
    def empty(rows, cols, device):
        """Uninitialised: only for buffers the next kernel overwrites completely (a 3n x 144 activation is 1.2 GB at 512 x 512 x 64:
        zero-filling every intermediate cost round 2's step 3.2 ms of fill kernels)."""
        return Mat(torch.empty(int(rows) * int(cols), dtype=torch.float32, device=device), rows, cols)

    @staticmethod
    def empty_ld(rows, cols, ld, device):
        """Uninitialised with padded rows (ld >= cols floats per row; the padding is never written nor read as data): a GEMM operand whose rows are
        16-byte aligned and hold whole 16-float K-blocks takes the streaming kernel (csrc/bwd_gemm.hip)."""
        return Mat(torch.empty(int(rows) * int(ld), dtype=torch.float32, device=device), rows, cols, ld)

    @staticmethod
    def of(t):
        """A contiguous 2-D (or 1-D -> one row) parameter tensor as a Mat (no copy when already fp32 contiguous)."""
        t = t.detach().to(torch.float32).contiguous()
        if t.dim() == 1:
            return Mat(t.view(-1), 1, t.numel())
        return Mat(t.view(-1), t.shape[0], t.numel() // t.shape[0])

    def colslice(self, c0, c1):
        return Mat(self.buf, self.rows, c1 - c0, self.ld, self.off + c0)

    def as_rows(self, rows, cols):
        assert self.ld == self.cols and rows * cols == self.rows * self.cols, 'as_rows needs a dense matrix'
        return Mat(self.buf, rows, cols, cols, self.off)

    def tensor(self):
        return torch.as_strided(self.buf, (self.rows, self.cols), (self.ld, 1), self.off)


class HipOps:
    """The C entry points of include/sherf_hip_bwd.h on Mats."""

    def __init__(self):
        self.st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(m):
        if not m.buf.is_cuda:
            raise RuntimeError('sherf_amd: tensor is not on a GPU; the HIP path has no CPU fallback')
        return ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off)

    @staticmethod
    def _rows_readable(A, K):
        """include/sherf_hip_bwd.h (sherf_bwd_gemm): the streaming tall kernel reads a row of a non-transposed A in whole 16-float blocks (masked in registers),
        so the last row's padding up to 16 ceil(K / 16) floats must lie inside the buffer."""
        blocks = 16 * ((K + 15) // 16)
        streaming = A.ld >= blocks and A.ld % 4 == 0                    # (sherf_bwd_gemm's own routing rule; otherwise the general kernel checks every element)
        return not streaming or A.off + (A.rows - 1) * A.ld + blocks <= A.buf.numel()

    def gemm(self, tA, tB, A, B, C, beta=0.0):
        M, K = (A.cols, A.rows) if tA else (A.rows, A.cols)
        K2, N = (B.cols, B.rows) if tB else (B.rows, B.cols)
        assert K == K2 and C.rows == M and C.cols == N, ('gemm shapes', tA, tB, A.rows, A.cols, B.rows, B.cols, C.rows, C.cols)
        assert tA or self._rows_readable(A, K), ('gemm: row padding of A not readable', A.off, A.rows, A.ld, K, A.buf.numel())
        _lib.call_bwd('sherf_bwd_gemm', int(tA), int(tB), M, N, K, self._p(A), A.ld, self._p(B), B.ld, self._p(C), C.ld, float(beta), self.st)

    def gemm_bias_act(self, tA, tB, A, B, C, bias, act, beta=0.0):
        M, K = (A.cols, A.rows) if tA else (A.rows, A.cols)
        K2, N = (B.cols, B.rows) if tB else (B.rows, B.cols)
        assert K == K2 and C.rows == M and C.cols == N and (bias is None or bias.rows * bias.cols == N), ('gemm_bias_act shapes', M, N, K)
        assert tA or self._rows_readable(A, K), ('gemm_bias_act: row padding of A not readable', A.off, A.rows, A.ld, K, A.buf.numel())
        _lib.call_bwd('sherf_bwd_gemm_bias_act', int(tA), int(tB), M, N, K, self._p(A), A.ld, self._p(B), B.ld, self._p(C), C.ld, float(beta),
                      None if bias is None else self._p(bias), act, self.st)

    def gemm_bias_act_add(self, tB, A, B, C, bias, act, addend):
        """C = act(A . op(B) + bias) + addend (sherf_bwd_gemm_bias_act_add: a Linear joining a residual stream)."""
        K2, N = (B.cols, B.rows) if tB else (B.rows, B.cols)
        assert K2 == A.cols and (C.rows, C.cols) == (A.rows, N) == (addend.rows, addend.cols) and addend.buf is not C.buf, 'gemm_bias_act_add shapes'
        _lib.call_bwd('sherf_bwd_gemm_bias_act_add', int(tB), A.rows, N, A.cols, self._p(A), A.ld, self._p(B), B.ld, self._p(C), C.ld,
                      None if bias is None else self._p(bias), act, self._p(addend), addend.ld, self.st)

    def gemm_dgrad_fused(self, A, B, C, r1_s=None, r1_w=None, mask=None, colsum=None):
        """C = A . B (+ r1_s r1_w) masked by `mask` > 0, colsum += column sums of C (sherf_bwd_gemm_dgrad_fused)."""
        assert B.rows == A.cols and (C.rows, C.cols) == (A.rows, B.cols) and (r1_s is None) == (r1_w is None), 'gemm_dgrad_fused shapes'
        assert r1_s is None or (r1_s.rows == A.rows and r1_s.cols == 1 and r1_w.rows * r1_w.cols == B.cols)
        assert mask is None or (mask.rows, mask.cols) == (C.rows, C.cols)
        o = lambda m: None if m is None else self._p(m)
        _lib.call_bwd('sherf_bwd_gemm_dgrad_fused', A.rows, B.cols, A.cols, self._p(A), A.ld, self._p(B), B.ld, self._p(C), C.ld, o(r1_s),
                      0 if r1_s is None else r1_s.ld, o(r1_w), o(mask), 0 if mask is None else mask.ld, o(colsum), self.st)

    def relu_mask_colsum(self, D, H, out):
        _lib.call_bwd('sherf_bwd_relu_mask_colsum', self._p(D), D.ld, self._p(H), H.ld, D.rows, D.cols, self._p(out), self.st)

    def bias_act(self, Y, bias, act):
        _lib.call_bwd('sherf_bwd_bias_act', self._p(Y), Y.ld, None if bias is None else self._p(bias), Y.rows, Y.cols, act, self.st)

    def relu_mask(self, D, H):
        _lib.call_bwd('sherf_bwd_relu_mask', self._p(D), D.ld, self._p(H), H.ld, D.rows, D.cols, self.st)

    def colsum(self, D, out):
        _lib.call_bwd('sherf_bwd_colsum', self._p(D), D.ld, D.rows, D.cols, self._p(out), self.st)

    def copy2d(self, dst, src, add=False):
        _lib.call_bwd('sherf_bwd_copy2d', self._p(dst), dst.ld, self._p(src), src.ld, src.rows, src.cols, int(add), self.st)

    def pe(self, inp, NF, out):
        _lib.call_bwd('sherf_bwd_pe', self._p(inp), inp.ld, inp.rows, NF, self._p(out), out.ld, self.st)

    def ln_fwd(self, x, w, b, y, xh, inv):
        _lib.call_bwd('sherf_bwd_ln_fwd', self._p(x), self._p(w), self._p(b), x.rows, self._p(y), self._p(xh), self._p(inv), self.st)

    def ln_bwd(self, dy, w, xh, inv, dx, dw, db, addend=None):
        if addend is None:
            _lib.call_bwd('sherf_bwd_ln_bwd', self._p(dy), self._p(w), self._p(xh), self._p(inv), dy.rows, self._p(dx), self._p(dw), self._p(db), self.st)
        else:                                       # dx = LN backward + a residual gradient joining (one pass less)
            assert addend.ld == addend.cols == 32 and addend.rows == dy.rows
            _lib.call_bwd('sherf_bwd_ln_bwd_add', self._p(dy), self._p(w), self._p(xh), self._p(inv), dy.rows, self._p(addend), self._p(dx), self._p(dw),
                          self._p(db), self.st)

    def attn_fwd(self, qkv, att, o):
        _lib.call_bwd('sherf_bwd_attn_fwd', self._p(qkv), qkv.rows, self._p(att), self._p(o), self.st)

    def attn_bwd(self, qkv, att, d_o, d_qkv):
        _lib.call_bwd('sherf_bwd_attn_bwd', self._p(qkv), self._p(att), self._p(d_o), qkv.rows, self._p(d_qkv), self.st)

    def gelu_fwd(self, u, ge):
        _lib.call_bwd('sherf_bwd_gelu_fwd', self._p(u), u.rows * u.cols, self._p(ge), self.st)

    def gelu_bwd(self, d, u):
        _lib.call_bwd('sherf_bwd_gelu_bwd', self._p(d), self._p(u), u.rows * u.cols, self.st)

    def rgb_fwd(self, lin):
        _lib.call_bwd('sherf_bwd_rgb_fwd', self._p(lin), lin.rows * lin.cols, self.st)

    def rgb_bwd(self, d, rgb):
        _lib.call_bwd('sherf_bwd_rgb_bwd', self._p(d), self._p(rgb), rgb.rows * rgb.cols, self.st)

    def untile(self, tokens_tiled, extras_tiled, n, tok, ext):
        _lib.call_bwd('sherf_bwd_untile', _lib.ptr(tokens_tiled), _lib.ptr(extras_tiled), n, self._p(tok), self._p(ext), self.st)

    def tile_tokens(self, d_tok, n, out):
        _lib.call_bwd('sherf_bwd_tile_tokens', self._p(d_tok), n, _lib.ptr(out), self.st)

    def unfold32(self, d_f, W, inp, HW, groups, pix_stride, group_base, d_in, dW):
        _lib.call_bwd('sherf_bwd_unfold32', self._p(d_f), self._p(W), self._p(inp), HW, groups, pix_stride, group_base, self._p(d_in),
                      self._p(dW), self.st)

    # ---- sparse encoder backward (levels are dicts: keys / wp / n_rows int32 tensors, dims (D,H,W), cap) ----
    def bn_relu_bwd(self, d_out, raw, bnparam, stats, gamma, mult, n_total, n_rows, d_raw, dgamma, dbeta):
        sums = Mat.empty(1, 3 * raw.cols + 1, raw.buf.device)                  # [3][C] scratch + the amax word (both zeroed by the call)
        amax = sums.colslice(3 * raw.cols, 3 * raw.cols + 1)
        _lib.call_bwd('sherf_bwd_bn_relu', self._p(d_out), self._p(raw), self._p(bnparam), self._p(stats), self._p(gamma),
                      None if mult is None else _lib.ptr(mult), _lib.ptr(n_total), _lib.ptr(n_rows), raw.rows, raw.cols, self._p(sums),
                      self._p(d_raw), self._p(dgamma), self._p(dbeta), self._p(amax), self.st)
        d_raw.amax = amax                                                       # read by conv_dgrad (bits of max |d_raw|)

    def conv_wgrad(self, lev_out, lev_in, in_raw, Cin, in_bn, in_mult, d_raw, Cout, mode, dW):
        _lib.call_bwd('sherf_bwd_conv_wgrad', _lib.ptr(lev_out['keys']), _lib.ptr(lev_out['n_rows']), *lev_out['dims'], _lib.ptr(lev_in['wp']),
                      *lev_in['dims'], self._p(in_raw), Cin, None if in_bn is None else self._p(in_bn),
                      None if in_mult is None else _lib.ptr(in_mult), self._p(d_raw), Cout, mode, lev_out['cap'], self._p(dW), self.st)

    def conv_dgrad(self, lev_in, lev_out, d_raw, Cout, W, Cin, mode, d_in):
        """On the forward's MFMA sparse-convolution kernel (sherf_svox_conv3_dgrad): the taps mirrored, the channel roles exchanged, d_raw
        scaled into the fp16 split's range by the maximum sherf_bwd_bn_relu left in d_raw.amax.  (The fp32 VALU kernel this replaces,
        sherf_bwd_conv_dgrad, took 16.5 ms of round 2's step; it stays in the library as the check: conv_dgrad_valu.)"""
        if getattr(d_raw, 'amax', None) is None:
            raise RuntimeError('conv_dgrad: d_raw carries no amax (it must come from bn_relu_bwd)')
        Wt = W.tensor().view(Cout, 27, Cin).flip(1).permute(1, 0, 2).contiguous()        # wt[k'][co][ci] = W[co][26 - k'][ci]
        from .voxel import pack_conv_weights
        wp = pack_conv_weights(Wt)
        _lib.call('sherf_svox_conv3_dgrad', _lib.ptr(lev_in['keys']), _lib.ptr(lev_in['n_rows']), *lev_in['dims'], _lib.ptr(lev_out['wp']),
                  *lev_out['dims'], self._p(d_raw), Cout, self._p(d_raw.amax), _lib.ptr(wp), Cin, mode, lev_in['cap'], self._p(d_in), self.st)

    def conv_dgrad_valu(self, lev_in, lev_out, d_raw, Cout, W, Cin, mode, d_in):
        _lib.call_bwd('sherf_bwd_conv_dgrad', _lib.ptr(lev_in['keys']), _lib.ptr(lev_in['n_rows']), *lev_in['dims'], _lib.ptr(lev_out['wp']),
                      *lev_out['dims'], self._p(d_raw), Cout, self._p(W), Cin, mode, lev_in['cap'], self._p(d_in), self.st)

    def gather_rows(self, coord, N, lev0, d_g, C, d_feat):
        _lib.call_bwd('sherf_bwd_gather_rows', _lib.ptr(coord), N, *lev0['dims'], _lib.ptr(lev0['wp']), self._p(d_g), C, self._p(d_feat), self.st)

    def bn_relu_apply(self, raw, bnparam, n_rows, act):
        _lib.call_bwd('sherf_bwd_bn_relu_apply', self._p(raw), self._p(bnparam), _lib.ptr(n_rows), raw.rows, raw.cols, self._p(act), self.st)


def dense_backward(ops, state, tok, ext, d_sample, use_trans=True):
    """tok [n,96] (gather output, row-major: slot tokens incl. bias, WITHOUT the slot-2 rgb encoding), ext [n,12]
    (x_c 0:3, v_c 3:6, tapped rgb 6:9), d_sample [n,4] = dL/d(rgb, sigma) per valid sample (compositing backward).
    `state`: {reference parameter name: tensor} of the renderer (prefix 'renderer.') and decoder ('decoder.').

    Returns (d_tokens_in Mat [n,96], grads {name: tensor}, dWb_pe [32,32]) where dWb_pe is the contribution of the slot-2
    rgb encoding to conv1d_reprojection.weight[:, 32:64] (the rest of that weight's gradient comes from the tap backward)."""
    n, dev = tok.rows, tok.buf.device
    Z = lambda r, c: Mat.zeros(r, c, dev)
    E = lambda r, c: Mat.empty(r, c, dev)                           # (every E below is written in full before it is read)
    EP = lambda r, c: Mat.empty_ld(r, c, (c + 15) // 16 * 16, dev)   # rows padded to whole 16-float blocks (K = 71, 199, 187 operands)
    P = lambda name: Mat.of(state[name])
    grads = {}

    def lin_fwd(x, wname, act, out=None, addend=None):
        W = P(wname + '.weight')                                   # [out, in]
        y = out if out is not None else E(x.rows, W.rows)
        b = P(wname + '.bias') if (wname + '.bias') in state else None
        if addend is not None:
            ops.gemm_bias_act_add(1, x, W, y, b, act, addend)     # (... and the residual stream the output joins)
        else:
            ops.gemm_bias_act(0, 1, x, W, y, b, act)               # (bias + ReLU in the product's store)
        return y

    def lin_bwd(d_out, x, wname, d_in=None, beta=0.0, bias=True, db=None, fuse=None, dgrad=True):
        """grads of y = x W^T + b; returns d_x (accumulated into d_in with beta).  db: the bias gradient when the caller already has it
        (relu_mask_colsum sums the columns while it masks).  fuse = dict(mask=, colsum=[, r1_s=, r1_w=]): x came out of a ReLU -- the data gradient
        is masked by the layer below's activations and its column sums (that layer's bias gradient) are taken in the product's store
        (ops.gemm_dgrad_fused).  dgrad=False: parameter gradients only."""
        W = P(wname + '.weight')
        dW = E(W.rows, W.cols)
        ops.gemm(1, 0, d_out, x, dW)
        grads[wname + '.weight'] = dW.tensor().view(state[wname + '.weight'].shape)          # (dW / db: buffers of their own, nothing else writes them: no copy)
        if bias and (wname + '.bias') in state:
            if db is None:
                db = Z(1, W.rows)
                ops.colsum(d_out, db)
            grads[wname + '.bias'] = db.tensor().view(-1)
        if not dgrad:
            return None
        dx = d_in if d_in is not None else E(d_out.rows, W.cols)
        if fuse is not None:
            assert beta == 0.0
            ops.gemm_dgrad_fused(d_out, W, dx, fuse.get('r1_s'), fuse.get('r1_w'), fuse.get('mask'), fuse.get('colsum'))
        else:
            ops.gemm(0, 0, d_out, W, dx, beta)
        return dx

    # ================= forward recompute =================
    t = 'renderer.transformer.layers.0.'
    if use_trans:
      Wr = state['renderer.conv1d_reprojection.weight'].detach().float()[:, :, 0]
      Wb = Mat.of(Wr[:, 32:64].contiguous())                          # [32 out, 32 in]
      pe_rgb = E(n, 33)
      ops.pe(ext.colslice(6, 9), 5, pe_rgb)
      tin = E(n, 96)                                                  # tokens_in = tok (+ slot 2: PE(rgb)[:32] Wb^T)
      ops.copy2d(tin, tok)
      ops.gemm(0, 1, pe_rgb.colslice(0, 32), Wb, tin.colslice(64, 96), 1.0)
      tin3 = tin.as_rows(3 * n, 32)
      h0, xh0, inv0 = E(3 * n, 32), E(3 * n, 32), E(3 * n, 1)
      ops.ln_fwd(tin3, P(t + '0.fn.norm.weight'), P(t + '0.fn.norm.bias'), h0, xh0, inv0)
      qkv = E(3 * n, 144)
      ops.gemm(0, 1, h0, P(t + '0.fn.fn.to_qkv.weight'), qkv)
      att, o = E(n, 27), E(3 * n, 48)
      ops.attn_fwd(qkv.as_rows(n, 432), att, o.as_rows(n, 144))
      y = lin_fwd(o, t + '0.fn.fn.to_out.0', 0, addend=tin3)          # residual (added in the product's store)
      h1, xh1, inv1 = E(3 * n, 32), E(3 * n, 32), E(3 * n, 1)
      ops.ln_fwd(y, P(t + '1.fn.norm.weight'), P(t + '1.fn.norm.bias'), h1, xh1, inv1)
      u = lin_fwd(h1, t + '1.fn.fn.net.0', 0)
      ge = E(3 * n, 32)
      ops.gelu_fwd(u, ge)
      z = lin_fwd(ge, t + '1.fn.fn.net.3', 0, addend=y)
      z96 = z.as_rows(n, 96)                                          # [n, slot 0 | slot 1 | slot 2]
    else:
      # use_trans = False (renderer.py:261, 427; round 6): the fused tokens go to the decoder as they are -- slots 0 / 1 of the gather's output (slot 2, and with it
      # the rgb encoding's W_b term, is never read)
      z96 = tok
    # ---- decoder ----
    d = 'decoder.'
    x0 = EP(n, 71)
    ops.pe(ext.colslice(0, 3), 6, x0.colslice(0, 39))
    ops.copy2d(x0.colslice(39, 71), z96.colslice(0, 32))
    ins, hs = [], []
    h = x0
    cat5 = EP(n, 199)
    for i in range(8):
        ins.append(h)
        out = cat5.colslice(71, 199) if i == 4 else None            # layer 4 writes straight into cat([x0, h4])
        hi = lin_fwd(h, d + f'pts_linears.{i}', 1, out)
        hs.append(hi)
        h = hi
        if i == 4:
            ops.copy2d(cat5.colslice(0, 71), x0)
            h = cat5
    h7 = hs[7]
    vin = EP(n, 187)
    lin_fwd(h7, d + 'feature_linear', 0, vin.colslice(0, 128))
    ops.pe(ext.colslice(3, 6), 4, vin.colslice(128, 155))
    ops.copy2d(vin.colslice(155, 187), z96.colslice(32, 64))
    g = lin_fwd(vin, d + 'views_linear', 1)
    rgb = lin_fwd(g, d + 'rgb_linear', 0)
    ops.rgb_fwd(rgb)

    # ================= backward =================
    d_lin = E(n, 3)
    ops.copy2d(d_lin, d_sample.colslice(0, 3))
    ops.rgb_bwd(d_lin, rgb)
    d_g = lin_bwd(d_lin, g, d + 'rgb_linear')
    db_v = Z(1, g.cols)
    ops.relu_mask_colsum(d_g, g, db_v)
    d_vin = lin_bwd(d_g, vin, d + 'views_linear', db=db_v, d_in=EP(n, 187))       # (its first 128 columns are the next product's A: aligned rows)
    d_sigma = d_sample.colslice(3, 4)
    # d_h7 = (d_feature . Wf + d_sigma Wa) * [h7 > 0] and its column sums (pts_linears.7's bias gradient) in one product: the sigma head's K = 1
    # product, the ReLU mask and the column sums ride in the store of the feature head's data gradient (round 5; 0.6 ms of passes per step before)
    db_i = Z(1, 128)
    d_h = lin_bwd(d_vin.colslice(0, 128), h7, d + 'feature_linear',
                  fuse=dict(r1_s=d_sigma, r1_w=P(d + 'alpha_linear.weight'), mask=hs[7], colsum=db_i))
    lin_bwd(d_sigma, h7, d + 'alpha_linear', dgrad=False)
    d_x0 = E(n, 71)
    premasked = True                                                # d_h already is masked by hs[i] and db_i holds its column sums
    for i in range(7, -1, -1):
        if not premasked:
            db_i = Z(1, d_h.cols)
            ops.relu_mask_colsum(d_h, hs[i], db_i)
        # the layer below's mask + bias gradient in this layer's data-gradient store, where that layer's output IS this layer's input (not layer 0)
        if i == 5:
            # the skip layer: its input is cat([x0, h4]) -- the data gradient as TWO products over the two column ranges of the weight, so that the x0 part
            # lands in d_x0 (no copy) and the h4 part in an aligned matrix of its own, masked by h4 with layer 4's bias gradient taken in the store (a
            # [n, 199] result sliced at column 71 made layer 4's product misaligned: the general kernel + a separate mask pass, 0.9 ms per step)
            lin_bwd(d_h, ins[5], d + 'pts_linears.5', db=db_i, dgrad=False)
            W5 = P(d + 'pts_linears.5.weight')
            ops.gemm(0, 0, d_h, W5.colslice(0, 71), d_x0)
            db_i, d_h4 = Z(1, 128), E(n, 128)
            ops.gemm_dgrad_fused(d_h, W5.colslice(71, 199), d_h4, mask=hs[4], colsum=db_i)
            d_h, premasked = d_h4, True
            continue
        nxt = Z(1, 128) if i in (7, 6, 4, 3, 2, 1) else None
        d_in = lin_bwd(d_h, ins[i], d + f'pts_linears.{i}', db=db_i, fuse=None if nxt is None else dict(mask=hs[i - 1], colsum=nxt))
        premasked, db_i = nxt is not None, nxt
        if i == 0:
            ops.copy2d(d_x0, d_in, add=True)
        else:
            d_h = d_in
    d_z = E(n, 96)                                                  # slot 2 of the transformer output is never read: its gradient is zero
    d_z.colslice(64, 96).tensor().zero_()                           # (the other two thirds are written in full below)
    ops.copy2d(d_z.colslice(0, 32), d_x0.colslice(39, 71))
    ops.copy2d(d_z.colslice(32, 64), d_vin.colslice(155, 187))
    if not use_trans:
        dWb0 = Z(32, 32)
        return d_z, grads, dWb0.tensor()
    d_out = d_z.as_rows(3 * n, 32)
    # ---- transformer: out = ge W2^T + b2 + y ----
    d_ge = lin_bwd(d_out, ge, t + '1.fn.fn.net.3')
    ops.gelu_bwd(d_ge, u)
    d_h1 = lin_bwd(d_ge, h1, t + '1.fn.fn.net.0')
    d_y, dw, db = E(3 * n, 32), Z(1, 32), Z(1, 32)
    ops.ln_bwd(d_h1, P(t + '1.fn.norm.weight'), xh1, inv1, d_y, dw, db, addend=d_out)          # (+ the residual branch's gradient)
    grads[t + '1.fn.norm.weight'], grads[t + '1.fn.norm.bias'] = dw.tensor().view(-1), db.tensor().view(-1)
    # ---- y = o Wo^T + bo + tokens_in ----
    d_o = lin_bwd(d_y, o, t + '0.fn.fn.to_out.0')
    d_qkv = E(3 * n, 144)
    ops.attn_bwd(qkv.as_rows(n, 432), att, d_o.as_rows(n, 144), d_qkv.as_rows(n, 432))
    d_h0 = lin_bwd(d_qkv, h0, t + '0.fn.fn.to_qkv', bias=False)
    d_tin, dw0, db0 = E(3 * n, 32), Z(1, 32), Z(1, 32)
    ops.ln_bwd(d_h0, P(t + '0.fn.norm.weight'), xh0, inv0, d_tin, dw0, db0, addend=d_y)
    grads[t + '0.fn.norm.weight'], grads[t + '0.fn.norm.bias'] = dw0.tensor().view(-1), db0.tensor().view(-1)
    d_tin96 = d_tin.as_rows(n, 96)
    dWb_pe = E(32, 32)
    ops.gemm(1, 0, d_tin96.colslice(64, 96), pe_rgb.colslice(0, 32), dWb_pe)
    return d_tin96, grads, dWb_pe.tensor()
