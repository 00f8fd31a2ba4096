"""Stand-in for `spconv.pytorch` (spconv-cu113==2.3.3 is pinned by the reference's requirement.txt:24
but is un-vendored and absent offline). TEST INFRASTRUCTURE: lets the unmodified reference
`SparseConvNet` (renderer.py:708-871) run on CPU.

Semantics emulated (spconv 2.x algorithm as published; "parity unpinned" -- no spconv source here):
  * SparseConvTensor(features[N,C], indices[N,4] int32 (b,z,y,x), spatial_shape, batch_size)
  * SubMConv3d: output sites == input rows. Indice pairs are generated from the INPUT side: input row i
    at voxel q and kernel tap k contribute to the row the hash table holds for voxel p = q-(k-1).
    When several rows share a voxel the table holds one of them (here: the lowest row); their features
    therefore SUM at that voxel for every consumer and the other rows receive no output (zeros).
  * SparseConv3d(k=3, s=2, p=1): output sites = every o with an active input in its receptive field,
    out[o] = sum_k W[k] . in[2o + k - 1]; rows are emitted sorted by linear index.
  * weight layout [out, kz, ky, kx, in] (KRSC), cross-correlation like torch.nn.functional.conv3d.
  * .dense(): zeros + index assignment into [B, C, D, H, W].
"""
import math
import torch
import torch.nn as nn

from . import core  # noqa: F401  (triplane.py:137 uses spconv.core.SparseConvTensor)
from .core import SparseConvTensor


class SparseModule(nn.Module):
    pass


def _keys(idx, shape):
    D, H, W = shape
    idx = idx.long()
    return ((idx[:, 0] * D + idx[:, 1]) * H + idx[:, 2]) * W + idx[:, 3]


class _ConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None, **kw):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.k, self.s, self.p = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, kernel_size, kernel_size, kernel_size, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.indice_key = indice_key


class SubMConv3d(_ConvBase):
    def forward(self, x):
        k = self.k
        assert k in (1, 3)
        shape = list(x.spatial_shape)
        idx = x.indices.long()
        keys = _keys(idx, shape)
        ukeys, inv = torch.unique(keys, sorted=True, return_inverse=True)
        N = idx.shape[0]
        winner = torch.full((ukeys.numel(),), N, dtype=torch.long)
        winner.scatter_reduce_(0, inv, torch.arange(N), reduce='amin')
        out = torch.zeros(N, self.out_channels, dtype=x.features.dtype)
        r = k // 2
        for kz in range(k):
            for ky in range(k):
                for kx in range(k):
                    p = idx.clone()
                    p[:, 1] -= kz - r; p[:, 2] -= ky - r; p[:, 3] -= kx - r
                    ok = (p[:, 1] >= 0) & (p[:, 1] < shape[0]) & (p[:, 2] >= 0) & (p[:, 2] < shape[1]) & (p[:, 3] >= 0) & (p[:, 3] < shape[2])
                    pk = _keys(p, shape)
                    pos = torch.searchsorted(ukeys, pk).clamp(max=ukeys.numel() - 1)
                    ok &= ukeys[pos] == pk
                    rows_in = torch.nonzero(ok)[:, 0]
                    if rows_in.numel() == 0:
                        continue
                    contrib = x.features[rows_in] @ self.weight[:, kz, ky, kx, :].t()
                    out.index_add_(0, winner[pos[rows_in]], contrib)
        if self.bias is not None:
            out = out + self.bias
        return SparseConvTensor(out, x.indices, x.spatial_shape, x.batch_size)


class SparseConv3d(_ConvBase):
    def forward(self, x):
        k, s, pad = self.k, self.s, self.p
        shape = list(x.spatial_shape)
        oshape = [(d + 2 * pad - k) // s + 1 for d in shape]
        idx = x.indices.long()
        pair_in, pair_key, pair_k = [], [], []
        for kz in range(k):
            for ky in range(k):
                for kx in range(k):
                    num = torch.stack([idx[:, 1] + pad - kz, idx[:, 2] + pad - ky, idx[:, 3] + pad - kx], 1)
                    ok = (num % s == 0).all(1)
                    o = num // s
                    for c in range(3):
                        ok &= (o[:, c] >= 0) & (o[:, c] < oshape[c])
                    rows = torch.nonzero(ok)[:, 0]
                    if rows.numel() == 0:
                        continue
                    oi = torch.cat([idx[rows, :1], o[rows]], 1)
                    pair_in.append(rows); pair_key.append(_keys(oi, oshape))
                    pair_k.append(torch.full_like(rows, (kz * k + ky) * k + kx))
        pair_in = torch.cat(pair_in); pair_key = torch.cat(pair_key); pair_k = torch.cat(pair_k)
        ukeys, inv = torch.unique(pair_key, sorted=True, return_inverse=True)
        out = torch.zeros(ukeys.numel(), self.out_channels, dtype=x.features.dtype)
        wflat = self.weight.reshape(self.out_channels, k * k * k, self.in_channels)
        for kk in range(k * k * k):
            m = pair_k == kk
            if m.any():
                out.index_add_(0, inv[m], x.features[pair_in[m]] @ wflat[:, kk, :].t())
        if self.bias is not None:
            out = out + self.bias
        D, H, W = oshape
        b = ukeys // (D * H * W); rem = ukeys % (D * H * W)
        oidx = torch.stack([b, rem // (H * W), (rem // W) % H, rem % W], 1).to(x.indices.dtype)
        return SparseConvTensor(out, oidx, oshape, x.batch_size)


class SparseSequential(SparseModule):
    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            else:
                x = x.replace_feature(m(x.features))
        return x
