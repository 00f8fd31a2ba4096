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
        C.tensor().copy_(a @ b + beta * C.tensor())

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

    def ln_bwd(self, dy, w, xh, inv, dx, dw, db):                    # sherf_bwd_ln_bwd (dw, db accumulate)
        DY, XH = dy.tensor(), xh.tensor()
        g = DY * w.tensor().view(-1)
        dx.tensor().copy_(inv.tensor() * (g - g.mean(-1, keepdim=True) - XH * (g * XH).mean(-1, keepdim=True)))
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
