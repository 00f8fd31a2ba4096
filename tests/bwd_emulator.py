"""Test double for the C entry points of include/sherf_hip_bwd.h: the same operations on CPU tensors through the same
`Mat` views (row-major, leading dimension, element offset), so the ORCHESTRATION in sherf_amd/backward_dense.py (shapes,
transposition flags, strides, accumulation, order) can be checked without a GPU.  Each method states the kernel it stands
for; the kernels themselves are checked against these semantics on hardware (tests/test_gpu_backward.py)."""
import math

import torch


class EmuOps:
    def gemm(self, tA, tB, A, B, C, beta=0.0):                       # sherf_bwd_gemm
        a = A.tensor().t() if tA else A.tensor()
        b = B.tensor().t() if tB else B.tensor()
        assert a.shape[1] == b.shape[0] and (C.rows, C.cols) == (a.shape[0], b.shape[1])
        C.tensor().copy_(a @ b + beta * C.tensor() if beta != 0.0 else a @ b)        # (beta == 0: C is not read, as in the kernels)

    def gemm_bias_act(self, tA, tB, A, B, C, bias, act, beta=0.0):   # sherf_bwd_gemm_bias_act
        self.gemm(tA, tB, A, B, C, beta)
        self.bias_act(C, bias, act)

    def gemm_bias_act_add(self, tB, A, B, C, bias, act, addend):      # sherf_bwd_gemm_bias_act_add
        add = addend.tensor().clone()
        self.gemm_bias_act(0, tB, A, B, C, bias, act)
        C.tensor().add_(add)

    def gemm_dgrad_fused(self, A, B, C, r1_s=None, r1_w=None, mask=None, colsum=None):      # sherf_bwd_gemm_dgrad_fused (colsum accumulates)
        y = A.tensor() @ B.tensor()
        if r1_s is not None:
            y = y + r1_s.tensor() * r1_w.tensor().reshape(1, -1)
        if mask is not None:
            y = y * (mask.tensor() > 0).float()
        C.tensor().copy_(y)
        if colsum is not None:
            colsum.tensor().add_(y.sum(0, keepdim=True))

    def relu_mask_colsum(self, D, H, out):                           # sherf_bwd_relu_mask_colsum (out accumulates)
        self.relu_mask(D, H)
        self.colsum(D, out)

    def bias_act(self, Y, bias, act):                                # sherf_bwd_bias_act
        y = Y.tensor() + (bias.tensor().view(-1) if bias is not None else 0.0)
        Y.tensor().copy_(torch.relu(y) if act == 1 else y)

    def relu_mask(self, D, H):                                       # sherf_bwd_relu_mask
        D.tensor().mul_((H.tensor() > 0).float())

    def colsum(self, D, out):                                        # sherf_bwd_colsum (accumulates)
        out.tensor().add_(D.tensor().sum(0, keepdim=True))

    def copy2d(self, dst, src, add=False):                           # sherf_bwd_copy2d
        if add:
            dst.tensor().add_(src.tensor())
        else:
            dst.tensor().copy_(src.tensor())

    def pe(self, inp, NF, out):                                      # sherf_bwd_pe
        x = inp.tensor()
        cols = [x]
        for q in range(NF):
            cols += [torch.sin((2.0 ** q) * x), torch.sin((2.0 ** q) * x + math.pi * 0.5)]
        out.tensor()[:, :3 + 6 * NF].copy_(torch.cat(cols, 1))

    def ln_fwd(self, x, w, b, y, xh, inv):                           # sherf_bwd_ln_fwd
        X = x.tensor()
        xc = X - X.mean(-1, keepdim=True)
        iv = 1.0 / torch.sqrt((xc ** 2).mean(-1, keepdim=True) + 1e-5)
        xh.tensor().copy_(xc * iv); inv.tensor().copy_(iv)
        y.tensor().copy_(xc * iv * w.tensor().view(-1) + b.tensor().view(-1))

    def ln_bwd(self, dy, w, xh, inv, dx, dw, db, addend=None):       # sherf_bwd_ln_bwd / sherf_bwd_ln_bwd_add (dw, db accumulate)
        add = None if addend is None else addend.tensor().clone()
        DY, XH = dy.tensor(), xh.tensor()
        g = DY * w.tensor().view(-1)
        dx.tensor().copy_(inv.tensor() * (g - g.mean(-1, keepdim=True) - XH * (g * XH).mean(-1, keepdim=True)))
        if add is not None:
            dx.tensor().add_(add)
        dw.tensor().add_((DY * XH).sum(0, keepdim=True)); db.tensor().add_(DY.sum(0, keepdim=True))

    def attn_fwd(self, qkv, att, o):                                 # sherf_bwd_attn_fwd
        n = qkv.rows
        q, k, v = [t.view(n, 3, 3, 16).permute(0, 2, 1, 3) for t in qkv.tensor().view(n, 3, 144).chunk(3, -1)]
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * 0.25, -1)
        att.tensor().copy_(a.reshape(n, 27))
        o.tensor().copy_(torch.matmul(a, v).permute(0, 2, 1, 3).reshape(n, 144))

    def attn_bwd(self, qkv, att, d_o, d_qkv):                        # sherf_bwd_attn_bwd
        n = qkv.rows
        q, k, v = [t.view(n, 3, 3, 16).permute(0, 2, 1, 3) for t in qkv.tensor().view(n, 3, 144).chunk(3, -1)]
        a = att.tensor().view(n, 3, 3, 3)
        go = d_o.tensor().view(n, 3, 3, 16).permute(0, 2, 1, 3)
        d_att = torch.matmul(go, v.transpose(-1, -2))
        d_v = torch.matmul(a.transpose(-1, -2), go)
        d_s = a * (d_att - (d_att * a).sum(-1, keepdim=True)) * 0.25
        d_q, d_k = torch.matmul(d_s, k), torch.matmul(d_s.transpose(-1, -2), q)
        d_qkv.tensor().copy_(torch.cat([t.permute(0, 2, 1, 3).reshape(n, 3, 48) for t in (d_q, d_k, d_v)], -1).reshape(n, 432))

    def gelu_fwd(self, u, ge):                                       # sherf_bwd_gelu_fwd
        U = u.tensor()
        ge.tensor().copy_(0.5 * U * (1 + torch.erf(U / math.sqrt(2.0))))

    def gelu_bwd(self, d, u):                                        # sherf_bwd_gelu_bwd
        U = u.tensor()
        d.tensor().mul_(0.5 * (1 + torch.erf(U / math.sqrt(2.0))) + U * torch.exp(-0.5 * U * U) / math.sqrt(2 * math.pi))

    def rgb_fwd(self, lin):                                          # sherf_bwd_rgb_fwd
        lin.tensor().copy_(torch.sigmoid(lin.tensor()) * 1.002 - 0.001)

    def rgb_bwd(self, d, rgb):                                       # sherf_bwd_rgb_bwd
        s = (rgb.tensor() + 0.001) / 1.002
        d.tensor().mul_(1.002 * s * (1 - s))

    def tile_tokens(self, d_tok, n, out):                            # sherf_bwd_tile_tokens: [n,96] -> [tile][3][8][32] float4
        tiles = (n + 31) // 32
        pad = torch.zeros(tiles * 32, 96)
        pad[:n] = d_tok.tensor()
        out.copy_(pad.view(tiles, 32, 3, 8, 4).permute(0, 2, 3, 1, 4).reshape(-1))

    def unfold32(self, d_f, W, inp, HW, groups, pix_stride, group_base, d_in, dW):     # sherf_bwd_unfold32
        Wm = W.tensor()                                              # [o, c]
        flat = d_f.buf[d_f.off:]
        for g in range(groups):
            D = torch.as_strided(flat, (HW, 32), (pix_stride, 1), g * group_base)      # [pix, o]
            X = inp.tensor()[32 * g:32 * g + 32]                                       # [c, pix]
            d_in.tensor()[32 * g:32 * g + 32].copy_(Wm.t() @ D.t())
            dW.tensor().add_(D.t() @ X.t())

    def bn_relu_apply(self, raw, bnparam, n_rows, act):              # sherf_bwd_bn_relu_apply
        C = raw.cols
        bn = bnparam.tensor().view(3, C)
        a = torch.relu(raw.tensor() * bn[0] + bn[1])
        a[int(n_rows):] = 0
        act.tensor().copy_(a)

    # ---- sparse encoder backward: levels are dicts(keys sorted int64, n_rows, dims (D,H,W), cap) ----
    @staticmethod
    def _rows_of(lev, z, y, x):
        """row ids of voxels (z,y,x) in a level, -1 where absent / out of range."""
        D, H, W = lev['dims']
        ok = (z >= 0) & (z < D) & (y >= 0) & (y < H) & (x >= 0) & (x < W)
        k = (z * H + y) * W + x
        keys = lev['keys'][:int(lev['n_rows'])].long()
        pos = torch.searchsorted(keys, k.clamp(min=0)).clamp(max=keys.numel() - 1)
        ok &= keys[pos] == k
        return torch.where(ok, pos, torch.full_like(pos, -1))

    @staticmethod
    def _zyx(lev):
        D, H, W = lev['dims']
        k = lev['keys'][:int(lev['n_rows'])].long()
        return k // (H * W), (k // W) % H, k % W

    def bn_relu_bwd(self, d_out, raw, bnparam, stats, gamma, mult, n_total, n_rows, d_raw, dgamma, dbeta):     # sherf_bwd_bn_relu
        C, n, N = raw.cols, int(n_rows), float(int(n_total))
        bn, st = bnparam.tensor().view(3, C), stats.tensor().view(2, C)
        scale, shift, mean, inv = bn[0], bn[1], st[0], 1.0 / torch.sqrt(st[1] + 1e-3)
        x, d = raw.tensor()[:n], d_out.tensor()[:n]
        dy = d * ((x * scale + shift) > 0).float()
        xh = (x - mean) * inv
        s3 = ((mult[:n].float() - 1)[:, None] * d).sum(0) if mult is not None else torch.zeros(C)
        dy0 = s3 * (shift > 0).float()
        s1, s2 = dy.sum(0) + dy0, (dy * xh).sum(0) + dy0 * (-mean * inv)
        d_raw.tensor().zero_()
        d_raw.tensor()[:n].copy_((gamma.tensor().view(-1) * inv / N) * (N * dy - s1 - xh * s2))
        dgamma.tensor().copy_(s2[None]); dbeta.tensor().copy_(s1[None])

    def _act(self, in_raw, in_bn, in_mult, rows):
        x = in_raw.tensor()[rows]
        if in_bn is None:
            return x
        C = in_raw.cols
        bn = in_bn.tensor().view(3, C)
        a = torch.relu(x * bn[0] + bn[1])
        if in_mult is not None:
            a = a + (in_mult[rows].float() - 1)[:, None] * bn[2]
        return a

    def conv_wgrad(self, lev_out, lev_in, in_raw, Cin, in_bn, in_mult, d_raw, Cout, mode, dW):                # sherf_bwd_conv_wgrad
        z, y, x = self._zyx(lev_out)
        out = dW.tensor().view(Cout, 27, Cin)
        for k in range(27):
            kz, ky, kx = k // 9, (k // 3) % 3, k % 3
            nb = self._rows_of(lev_in, 2 * z + kz - 1, 2 * y + ky - 1, 2 * x + kx - 1) if mode else self._rows_of(lev_in, z + kz - 1, y + ky - 1, x + kx - 1)
            o = torch.nonzero(nb >= 0)[:, 0]
            if o.numel():
                out[:, k, :] += d_raw.tensor()[o].t() @ self._act(in_raw, in_bn, in_mult, nb[o])

    def conv_dgrad(self, lev_in, lev_out, d_raw, Cout, W, Cin, mode, d_in):                                    # sherf_bwd_conv_dgrad
        z, y, x = self._zyx(lev_in)
        Wm = W.tensor().view(Cout, 27, Cin)
        res = torch.zeros(int(lev_in['n_rows']), Cin)
        for k in range(27):
            kz, ky, kx = k // 9, (k // 3) % 3, k % 3
            nz, ny, nx = z + 1 - kz, y + 1 - ky, x + 1 - kx
            if mode:
                even = (nz >= 0) & (ny >= 0) & (nx >= 0) & (nz % 2 == 0) & (ny % 2 == 0) & (nx % 2 == 0)
                nb = self._rows_of(lev_out, nz // 2, ny // 2, nx // 2)
                nb = torch.where(even, nb, torch.full_like(nb, -1))
            else:
                nb = self._rows_of(lev_out, nz, ny, nx)
            i = torch.nonzero(nb >= 0)[:, 0]
            if i.numel():
                res[i] += d_raw.tensor()[nb[i]] @ Wm[:, k, :]
        d_in.tensor()[:res.shape[0]].copy_(res)

    def gather_rows(self, coord, N, lev0, d_g, C, d_feat):                                                      # sherf_bwd_gather_rows
        c = coord.long()
        rows = self._rows_of(lev0, c[:, 1], c[:, 2], c[:, 3])
        d_feat.tensor().copy_(torch.where((rows >= 0)[:, None], d_g.tensor()[rows.clamp(min=0)], torch.zeros(N, C)))


def make_level(keys, dims, pad=5):
    """A sparse level from sorted unique voxel keys: (emulator dict, kernel dict with int32 keys + (bits, prefix) records)."""
    D, H, W = dims
    keys = keys.long()
    nwords = (D * H * W + 31) // 32
    bits = torch.zeros(nwords, dtype=torch.int64)
    bits.index_put_((keys // 32,), (torch.ones_like(keys) << (keys % 32)), accumulate=True)
    pop = torch.tensor([bin(int(b)).count('1') for b in bits.tolist()], dtype=torch.int64)
    prefix = torch.cumsum(pop, 0) - pop
    wp = torch.stack([bits, prefix], 1).view(-1)
    wp = torch.where(wp >= 2 ** 31, wp - 2 ** 32, wp).to(torch.int32).view(nwords, 2).contiguous()
    cap = keys.numel() + pad
    kp = torch.cat([keys, torch.zeros(pad, dtype=keys.dtype)])
    emu = dict(keys=kp, n_rows=torch.tensor(keys.numel()), dims=tuple(dims), cap=cap)
    ker = dict(keys=kp.to(torch.int32).contiguous(), wp=wp, n_rows=torch.tensor([keys.numel()], dtype=torch.int32), dims=tuple(dims), cap=cap)
    return emu, ker
