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


def pack_conv_weights(wt, lo_scale=2048.0):
    """wt [ntaps, Cin, Cout] fp32 -> fp16 MFMA B-operand fragments [ntaps][Cin/16][Cout/32][hi,lo][64 lanes][8] (int16 bits):
    hi = fp16(W), lo = fp16((W - hi) * 2^11) ("f16x3": 22 significant bits, three products in the kernel; the single-product instances read
    `hi` only).  The scale (round 6; csrc/svox.hip: SL): W - hi is below 2^-11 |W| and lost its low bits to fp16's 2^-24 floor for weights
    of 0.01 -- the kernel carries both operands' lo halves at 2^11 and folds the factor back into its second accumulator.
    Lane l = (column j = l & 31, half h = l >> 5) holds W[tap][16*kb + 8*h + e][32*cot + j], e = 0..7
    (operand layout of v_mfma_f32_32x32x16_f16; consumed by sconv3_kernel in csrc/svox.hip)."""
    T, Cin, Cout = wt.shape
    w = wt.float().reshape(T, Cin // 16, 2, 8, Cout // 32, 32).permute(0, 1, 4, 2, 5, 3).reshape(T, Cin // 16, Cout // 32, 64, 8)
    hi = w.to(torch.float16)
    lo = ((w - hi.float()) * lo_scale).to(torch.float16)
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
        from .renderer import fast_params, state_key          # (plain dict walks: this runs every frame)
        key = (self.__dict__.get('_key_memo') or state_key(fast_params(self))) + (str(device),)
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
                                   tap=(name in ('conv1', 'conv2', 'conv3') and i == n - 1), wname=f'{name}.{3 * i}', bname=f'{name}.{3 * i + 1}'))
        self._packed = dict(key=key, layers=layers)
        return self._packed

    def plan(self, spatial_shape, N, fold_mats, ws, dev):
        """The native encoder's plan (`sherf_svox_plan`): every persistent buffer of the chain, allocated once per
        (shape, N, weights version) in the workspace `ws` and reused by every frame."""
        pk = self._pack(dev)
        shapes = [tuple(int(v) for v in spatial_shape)]
        for _ in range(3):
            shapes.append(tuple((d - 1) // 2 + 1 for d in shapes[-1]))
        key = (pk['key'], tuple(shapes), N, str(dev), tuple(m.data_ptr() for m in fold_mats))
        cached = getattr(ws, 'vox_plan', None)
        if cached is not None and cached['key'] == key:
            return cached
        L, zero_region = ws.voxel_levels(shapes, N, dev)
        A = _lib.addr
        plan = _lib.SvoxPlan()
        for i, (lv, sh) in enumerate(zip(L, shapes)):
            c = plan.lev[i]
            c.bitmap, c.prefix, c.n_rows, c.chunk_ws = A(lv['bitmap']), A(lv['prefix']), A(lv['n_rows']), A(lv['chunk_ws'])
            c.wp, c.keys, c.n_words, c.cap = A(lv['wp']), A(lv['keys']), lv['nwords'], lv['cap']
            c.D, c.H, c.W = sh
        ctot = sum(ly['cout'] for ly in pk['layers'])
        stats_flat = torch.zeros(2 * ctot, device=dev)                     # all layers' [2][C] stats, one buffer
        lev, off, meta, taps = 0, 0, [], []
        for li, ly in enumerate(pk['layers']):
            dlev = lev + 1 if ly['down'] else lev
            cap, C = L[dlev]['cap'], ly['cout']
            out = ws.layer_out(li, cap, C, dev)
            stats = stats_flat[off:off + 2 * C].view(2, C); off += 2 * C
            bnp = ws.bn_param(li, C, dev)
            c = plan.layers[li]
            c.cin, c.cout, c.down, c.tap = ly['cin'], C, int(ly['down']), int(ly['tap'])
            c.wt, c.gamma, c.beta = A(ly['wt']), A(ly['gamma']), A(ly['beta'])
            c.stats, c.bnparam, c.out = A(stats), A(bnp), A(out)
            c.acc = L[0]['bn_acc'].data_ptr() + li * 8 * 2 * 96 * 8                 # [8][2][C] int64 inside the zero region
            meta.append(dict(bn=ly['bn'], stats=stats, bnp=bnp, out=out, lev=dlev, lev_in=lev, cout=C, cin=ly['cin'], down=bool(ly['down']),
                             tap=bool(ly['tap']), wname=ly['wname'], bname=ly['bname'], stats_off=off - 2 * C))
            if ly['tap']:
                taps.append((dlev, out, bnp, C))
            lev = dlev
        plan.n_layers = len(pk['layers'])
        l0 = L[0]
        plan.acc_fix, plan.g0, plan.mult, plan.n_total = A(l0['acc_fix']), A(l0['g0']), A(l0['mult']), A(l0['n_total'])
        plan.zero_ptr, plan.zero_bytes = A(zero_region), zero_region.numel() * 4
        rows = []
        for i, (tl, _, _, _) in enumerate(taps):
            r = ws.fold_out(i, L[tl]['cap'], dev)
            plan.fold_mat[i], plan.fold_rows[i] = A(fold_mats[i]), A(r)
            rows.append(r)
        ws.vox_plan = dict(key=key, plan=plan, L=L, shapes=shapes, meta=meta, taps=taps, rows=rows, stats_flat=stats_flat,
                           keep=(zero_region, tuple(fold_mats), pk))
        return ws.vox_plan

    def prepare(self, sp, fold_mats, ws):
        """Host-side part of a frame: plan lookup + (eval mode) running statistics into the plan's stats buffer.
        Returns (plan dict, features fp32, coordinates int32)."""
        feat, coord = sp.features, sp.indices
        if feat.dtype is not torch.float32 or feat.requires_grad or not feat.is_contiguous():
            feat = feat.detach().float().contiguous()
        if coord.dtype is not torch.int32 or not coord.is_contiguous():
            coord = coord.to(torch.int32).contiguous()
        pl = self.plan(sp.spatial_shape, feat.shape[0], fold_mats, ws, feat.device)
        if not self.training:
            # eval (the reference's inference path: G_ema.eval(), training_loop.py:196): BatchNorm uses the running
            # statistics, so scale/shift are per-weights constants -- recomputed only when a buffer changes.
            key = tuple((t.data_ptr(), t._version) for m in pl['meta'] for t in (m['bn'].running_mean, m['bn'].running_var,
                                                                                  m['bn'].weight, m['bn'].bias))
            if pl.get('eval_key') != key:
                P = _lib.ptr
                with torch.no_grad():
                    pl['stats_flat'].copy_(torch.cat([t.reshape(-1) for m in pl['meta']
                                                      for t in (m['bn'].running_mean, m['bn'].running_var)]).float())
                for m, ly in zip(pl['meta'], self._pack(feat.device)['layers']):
                    _lib.call('sherf_svox_bn_finalize', None, None, m['cout'], P(ly['gamma']), P(ly['beta']), P(m['stats']), 0,
                              P(m['bnp']), _lib.stream())
                pl['eval_key'] = key
        return pl, feat, coord

    def finish(self, pl, stream=None):
        """nn.BatchNorm1d side effect in train mode (momentum 0.01, unbiased variance); call after the encoder was enqueued.
        Like nn.BatchNorm1d it depends on `self.training` only -- a train-mode forward under torch.no_grad() (the reference's
        test loop renders with G in train mode, training_loop.py:193,311-330) advances the running statistics too.
        `stream`: a raw stream handle to launch on (the frame driver's encoder stream; default: the caller's current stream)."""
        if not self.training:
            return
        import ctypes
        ms = [m for m in pl['meta'] if m['bn'].track_running_stats]
        n = len(ms)
        if not n:
            return
        # the argument arrays are rebuilt only when a buffer has moved (same storage frame after frame: load_state_dict copies in place)
        key = tuple([(m['bn'].running_mean.data_ptr(), m['bn'].running_var.data_ptr(), m['bn'].num_batches_tracked.data_ptr(), m['bn'].momentum)
                     for m in ms])
        args = pl.get('running_args')
        if args is None or args[0] != key:
            VP, A = ctypes.c_void_p * n, _lib.addr
            rows = [pl['L'][0]['n_total'] if m['lev'] == 0 else pl['L'][m['lev']]['n_rows'] for m in ms]
            for m in ms:
                if m['bn'].momentum is None:
                    raise NotImplementedError('BatchNorm1d(momentum=None) (cumulative moving average) is not implemented by the native running-'
                                              'statistics update; the reference uses momentum=0.01 (renderer.py:807)')
            f32, i32, i64 = torch.float32, torch.int32, torch.int64                # the kernel reads raw pointers: check what they point at
            args = pl['running_args'] = (key, (n, VP(*[A(m['stats'], f32) for m in ms]), VP(*[A(m['bn'].running_mean, f32) for m in ms]),
                                               VP(*[A(m['bn'].running_var, f32) for m in ms]), VP(*[A(m['bn'].num_batches_tracked, i64) for m in ms]),
                                               VP(*[A(r, i32) for r in rows]), (ctypes.c_int32 * n)(*[m['cout'] for m in ms]),
                                               (ctypes.c_float * n)(*[float(m['bn'].momentum) for m in ms])))
        _lib.call('sherf_svox_bn_running_update', *args[1], _lib.stream() if stream is None else stream)

    def encode(self, sp, fold_mats, ws):
        """Runs the encoder on a SparseConvTensor (one native call, csrc/svox.hip: sherf_svox_encode); returns the three
        tapped levels as `_lib.VoxLevel`s whose rows are already multiplied by `fold_mats[l]` ([C_l, 96]) -- see
        ImportanceRenderer._weights.  Per layer: one tiled conv launch (BatchNorm+ReLU of the INPUT applied while
        gathering, fp64 partial sums of the OUTPUT) + one tiny finalize launch (scale/shift for the next layer)."""
        import ctypes
        pl, feat, coord = self.prepare(sp, fold_mats, ws)
        levels = (_lib.VoxLevel * 3)()
        _lib.call('sherf_svox_encode', ctypes.byref(pl['plan']), _lib.ptr(coord), _lib.ptr(feat), feat.shape[0],
                  1 if self.training else 0, levels, _lib.stream())
        self.finish(pl)
        return levels, pl['rows'], dict(levels=pl['L'], taps=pl['taps'], shapes=pl['shapes'])
