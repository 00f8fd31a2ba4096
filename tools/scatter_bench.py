"""Times the backward's tap scatter alone on a real frame (config 2 unless given): the direct form, the binned form, and the binned form
with parts switched off (sherf_set_debug bits 14-16: no pixel / voxel / plane taps -- timing only, results incomplete).
    python tools/scatter_bench.py [config]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                                  # noqa: E402
from sherf_amd import _lib                                    # noqa: E402
from sherf_amd.backward_dense import HipOps, Mat              # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    dev = torch.device('cuda', 0)
    a = type('A', (), dict(config=cfg, precision='f16x3', table_precision=None, encoder_precision=None, eval_mode=False))()
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder, TriPlaneGenerator
    from sherf_amd.voxel import SparseConvTensor
    from synthdata import fixtures, synth
    fx, d, to = bench.make_inputs(cfg, 0.4, dev)
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=synth.make_synth_smpl(0), mlp_precision='f16x3')
    dec = NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.', fixtures.variant_of(cfg)); fixtures.load_seeded_state(dec, 'decoder.', fixtures.variant_of(cfg))
    rend.to(dev).train(); dec.to(dev).train()
    gen = TriPlaneGenerator.__new__(TriPlaneGenerator)
    torch.nn.Module.__init__(gen); gen.renderer = rend
    sp_input, _ = gen.prepare_sp_input(d['t_vertices'].float(), gen.canonical_obs_vertices(d))
    sp = SparseConvTensor(to(fx['vertex_feat']), sp_input['coord'], sp_input['out_sh'], 1)
    with torch.no_grad():
        rend(to(fx['planes']), d['obs_img_all'][:, 0], to(fx['obs_feat']), sp, None, sp_input, dec, d['ray_o_all'][:, 0], d['ray_d_all'][:, 0],
             d['near_all'][:, 0], d['far_all'][:, 0], d, dict(fx['options']))
    last, ws = rend.last, rend.last['ws']
    b = last['bwd']
    n = int(ws['counters'][0])
    P, (Hf, Wf) = fx['planes'].shape[-1], fx['obs_feat'].shape[-2:]
    L, taps = last['plan']['L'], last['plan']['taps']
    tiles = (n + 31) // 32
    d_tiled = torch.randn(tiles * 3072, device=dev) * 1e-6
    ops = HipOps()
    f32 = lambda t: t.detach().float().contiguous()
    bounds, vox_min = f32(b['bounds']).view(6), f32(b['vox_min']).view(3)
    words = ctypes.c_int64(0)
    _lib.call('sherf_gather_bwd_scratch_words', last['levels_struct'], last['cap'], ctypes.byref(words))
    scratch = torch.empty(words.value, dtype=torch.int32, device=dev)
    outs = [Mat.zeros(3 * P * P, 32, dev), Mat.zeros(Hf * Wf, 64, dev), Mat.zeros(1, 96, dev)]
    rows = [Mat.zeros(L[t[0]]['cap'], 96, dev) for t in taps]
    args = (_lib.ptr(ws['counters']), _lib.ptr(ws['geom']), _lib.ptr(d_tiled), P, Hf, Wf, b['H'], b['W'], last['levels_struct'], _lib.ptr(bounds),
            _lib.ptr(vox_min), (ctypes.c_int32 * 3)(*b['vox_sh']), last['cap'], ops._p(outs[0]), ops._p(outs[1]), ops._p(rows[0]), ops._p(rows[1]),
            ops._p(rows[2]), ops._p(outs[2]))

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
    lib = _lib.lib()
    cnt = None
    print(f'{cfg}: {n} valid samples, levels ' + ' '.join(f"{int(L[t[0]]['n_rows'])}" for t in taps) + f' rows, P {P}, feature map {Hf}x{Wf}')
    print(f'direct form                         {timed(lambda: _lib.call("sherf_gather_tokens_bwd", *args, _lib.stream())):8.3f} ms')
    for name, bits in (('sorted by finest cell, run-length sums (round 5)', 0), ('... no pixel taps', 16384), ('... no voxel taps', 32768), ('... no plane taps', 65536),
                       ('... no taps at all (sort + walk)', 16384 | 32768 | 65536)):
        lib.sherf_set_debug(bits)
        t = timed(lambda: _lib.call('sherf_gather_tokens_bwd_binned', *args, _lib.ptr(scratch), words.value, _lib.stream()))
        print(f'{name:52s} {t:8.3f} ms')
        if cnt is None:
            l0 = last['levels_struct'][0]
            nb = (l0.D + 4) * (l0.H + 4) * (l0.W + 4)                 # (round 5: the bins are the finest tapped level's cells)
            cnt = scratch[4:4 + nb].cpu().numpy()
            ne = cnt[cnt > 0]
            print(f'    bins {nb}, non-empty {ne.size}, samples per non-empty bin: mean {ne.mean():.1f} median {np.median(ne):.0f} p90 {np.percentile(ne, 90):.0f} max {ne.max()}')
    lib.sherf_set_debug(0)
    os.environ['SHERF_EXPERIMENT'] = '256'
    print(f"round 3's kernel (binned by the coarsest cell)      {timed(lambda: _lib.call('sherf_gather_tokens_bwd_binned', *args, _lib.ptr(scratch), words.value, _lib.stream())):8.3f} ms")
    os.environ['SHERF_EXPERIMENT'] = '0'


if __name__ == '__main__':
    main()
