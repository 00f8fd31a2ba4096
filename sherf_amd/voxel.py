"""Sparse voxel encoder: drop-in parameter container for the reference's `SparseConvNet`
(training/volumetric_rendering/renderer.py:708-871) and its HIP execution (sherf_amd/csrc/svox.hip).

The module tree reproduces the reference's state_dict keys (`encoder_3d.conv0.0.weight` ... spconv KRSC weight
shape [out, 3, 3, 3, in]; `encoder_3d.conv0.1.{weight,bias,running_mean,running_var,num_batches_tracked}`)
so `copy_params_and_buffers(require_all=True)` (training_loop.py:207-208) keeps working.  conv4 / down3 exist
but never influence the output with num_layers=4 (renderer.py:778-792) and are not executed.
"""
import math

import torch
import torch.nn as nn

from . import _lib


class SparseConvTensor:
    """Minimal stand-in for spconv.core.SparseConvTensor as built at triplane.py:137."""

    def __init__(self, features, indices, spatial_shape, batch_size=1):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)


class _SpConv(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = cin, cout, stride
        self.weight = nn.Parameter(torch.empty(cout, 3, 3, 3, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class SubMConv3d(_SpConv):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, 1)


class SparseConv3d(_SpConv):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, 2)


def _bn(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)


def _multi_conv(cin, cout, n):
    mods = []
    for i in range(n):
        mods += [SubMConv3d(cin if i == 0 else cout, cout), _bn(cout), nn.ReLU()]
    return nn.Sequential(*mods)


def _stride_conv(cin, cout):
    return nn.Sequential(SparseConv3d(cin, cout), _bn(cout), nn.ReLU())


def pack_conv_weights(wt):
    """wt [ntaps, Cin, Cout] fp32 -> bf16 MFMA B-operand fragments [ntaps][Cin/16][Cout/32][hi,lo][64 lanes][8] (int16 bits).
    Lane l = (column j = l & 31, half h = l >> 5) holds W[tap][16*kb + 8*h + e][32*cot + j], e = 0..7
    (operand layout of v_mfma_f32_32x32x16_bf16; consumed by sconv3_kernel in csrc/svox.hip)."""
    T, Cin, Cout = wt.shape
    w = wt.float().reshape(T, Cin // 16, 2, 8, Cout // 32, 32).permute(0, 1, 4, 2, 5, 3).reshape(T, Cin // 16, Cout // 32, 64, 8)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], 3).contiguous().view(torch.int16)


# executed part of the network: (module name, number of convs in it); a tap follows conv1, conv2, conv3
_PLAN = (('conv0', 2), ('down0', 1), ('conv1', 2), ('down1', 1), ('conv2', 3), ('down2', 1), ('conv3', 3))


class SparseConvNet(nn.Module):
    def __init__(self, num_layers=4):
        super().__init__()
        assert num_layers == 4, 'the reference instantiates SparseConvNet(num_layers=4) (renderer.py:270)'
        self.num_layers = num_layers
        self.conv0 = _multi_conv(32, 32, 2); self.down0 = _stride_conv(32, 32)
        self.conv1 = _multi_conv(32, 32, 2); self.down1 = _stride_conv(32, 64)
        self.conv2 = _multi_conv(64, 64, 3); self.down2 = _stride_conv(64, 96)
        self.conv3 = _multi_conv(96, 96, 3); self.down3 = _stride_conv(96, 96)
        self.conv4 = _multi_conv(96, 96, 3)
        self._packed = None

    # ---- host-side caches -------------------------------------------------------------------
    def _pack(self, device):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(device),)
        if self._packed is not None and self._packed['key'] == key:
            return self._packed
        layers = []
        for name, n in _PLAN:
            seq = getattr(self, name)
            for i in range(n):
                conv, bn = seq[3 * i], seq[3 * i + 1]
                w = conv.weight.detach().to(device=device, dtype=torch.float32)
                wt = w.permute(1, 2, 3, 4, 0).reshape(27, conv.in_channels, conv.out_channels).contiguous()
                layers.append(dict(wt=pack_conv_weights(wt), cin=conv.in_channels, cout=conv.out_channels, down=conv.stride == 2, bn=bn,
                                   gamma=bn.weight.detach().float().contiguous(), beta=bn.bias.detach().float().contiguous(),
                                   tap=(name in ('conv1', 'conv2', 'conv3') and i == n - 1)))
        self._packed = dict(key=key, layers=layers)
        return self._packed

    def encode(self, sp, fold_mats, ws):
        """Runs the encoder on a SparseConvTensor; returns the three tapped levels as `_lib.VoxLevel`s whose rows
        are already multiplied by `fold_mats[l]` ([C_l, 96]) -- see ImportanceRenderer._weights.

        Per layer: one tiled conv launch (BatchNorm+ReLU of the INPUT applied while gathering, fp64 partial sums
        of the OUTPUT) + one tiny finalize launch that turns the partials into scale/shift for the next layer."""
        feat = sp.features.detach().float().contiguous()
        coord = sp.indices.to(torch.int32).contiguous()
        dev = feat.device
        N = feat.shape[0]
        pk = self._pack(dev)
        shapes = [tuple(sp.spatial_shape)]
        for _ in range(3):
            shapes.append(tuple((d - 1) // 2 + 1 for d in shapes[-1]))
        L, zero_region = ws.voxel_levels(shapes, N, dev)
        st = _lib.stream()
        P = _lib.ptr
        training = self.training
        zero_region.zero_()                                   # every bitmap, level-0 multiplicities and summed features
        l0 = L[0]
        D, H, W = shapes[0]
        _lib.call('sherf_svox_mark_rows', P(coord), N, D, H, W, P(l0['bitmap']), st)
        _lib.call('sherf_svox_scan', P(l0['bitmap']), l0['nwords'], P(l0['prefix']), P(l0['n_rows']), P(l0['chunk_ws']), P(l0['wp']), st)
        _lib.call('sherf_svox_keys', P(l0['bitmap']), P(l0['prefix']), l0['nwords'], P(l0['keys']), st)
        _lib.call('sherf_svox_scatter_rows', P(coord), P(feat), N, 32, D, H, W, P(l0['bitmap']), P(l0['prefix']), P(l0['n_rows']),
                  P(l0['acc_fix']), P(l0['g0']), P(l0['mult']), st)
        lev = 0
        cur, cur_bn = l0['g0'], None                          # raw features of the current level + their BN params (None: raw)
        taps = []
        for li, ly in enumerate(pk['layers']):
            src = L[lev]
            dst = L[lev + 1] if ly['down'] else src
            if ly['down']:
                _lib.call('sherf_svox_mark_down', P(src['keys']), P(src['n_rows']), *shapes[lev], P(dst['bitmap']), src['cap'], st)
                _lib.call('sherf_svox_scan', P(dst['bitmap']), dst['nwords'], P(dst['prefix']), P(dst['n_rows']), P(dst['chunk_ws']), P(dst['wp']), st)
                _lib.call('sherf_svox_keys', P(dst['bitmap']), P(dst['prefix']), dst['nwords'], P(dst['keys']), st)
            out = ws.layer_out(li, dst['cap'], ly['cout'], dev)
            rpb = 32                                          # output rows per workgroup of sconv3_kernel
            parts = ws.partials(li, (dst['cap'] + rpb - 1) // rpb, ly['cout'], dev)
            mult = P(src['mult']) if (lev == 0 and cur_bn is not None) else None
            dlev = lev + 1 if ly['down'] else lev
            _lib.call('sherf_svox_conv3', P(dst['keys']), P(dst['n_rows']), *shapes[dlev], P(src['wp']),
                      *shapes[lev], P(cur), ly['cin'], P(cur_bn) if cur_bn is not None else None, mult, P(ly['wt']), ly['cout'],
                      1 if ly['down'] else 0, dst['cap'], P(out), P(parts), st)
            bn = ly['bn']
            stats = ws.bn_stats(li, ly['cout'], dev)
            if not training:
                stats[0].copy_(bn.running_mean); stats[1].copy_(bn.running_var)
            n_total = l0['n_total'] if dlev == 0 else dst['n_rows']
            bnp = ws.bn_param(li, ly['cout'], dev)
            _lib.call('sherf_svox_bn_finalize', P(parts), P(dst['n_rows']), P(n_total), ly['cout'], rpb, P(ly['gamma']), P(ly['beta']),
                      P(stats), 1 if training else 0, P(bnp), st)
            if training and torch.is_grad_enabled() and bn.track_running_stats:
                with torch.no_grad():     # nn.BatchNorm1d side effect in train mode (momentum 0.01, unbiased variance)
                    n = n_total.float()
                    bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * stats[0])
                    bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * stats[1] * n / (n - 1))
                    bn.num_batches_tracked += 1
            lev, cur, cur_bn = dlev, out, bnp
            if ly['tap']:
                taps.append((lev, out, bnp, ly['cout']))
        levels = (_lib.VoxLevel * 3)()
        keep = []
        for i, (lev, raw, bnp, C) in enumerate(taps):
            rows = ws.fold_out(i, L[lev]['cap'], dev)                   # relu(bn(raw)) @ fold [C, 96] as a pointwise "conv"
            _lib.call('sherf_svox_conv3', None, P(L[lev]['n_rows']), 1, 1, 1, None, 1, 1, 1, P(raw), C, P(bnp), None,
                      P(fold_mats[i]), 96, 2, L[lev]['cap'], P(rows), None, st)
            keep.append(rows)
            levels[i].wp = L[lev]['wp'].data_ptr()
            levels[i].rows = rows.data_ptr()
            levels[i].D, levels[i].H, levels[i].W = shapes[lev]
        return levels, keep, dict(levels=L, taps=taps, shapes=shapes)
