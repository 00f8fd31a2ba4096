"""GPU diagnostic (round 3): layer-by-layer comparison of the sparse encoder's raw outputs between the f16x3 and the single-product
(SHERF_FRAME_ENCODER_SINGLE) convolutions on one frame -- where do they part?   python tools/enc_sp_diag.py [cfg]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_common as G


def main():
    global G
    args = [x for x in sys.argv[1:] if x != '--cpu']
    if '--cpu' in sys.argv:                      # the same comparison on the HOST build of the kernels (tools/cpu_shim.py)
        from tools import cpu_shim
        G = cpu_shim.enable()
    cfg = args[0] if args else 'tiny_ri'
    outs = {}
    for enc in ('f16x3', 'f16'):
        h = G.hip_render(cfg, precision='f16', options=dict(table_precision='f32', encoder_precision=enc))
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        pl = h['last']['plan']
        rows = [int(pl['L'][m['lev']]['n_rows'][0]) if m['lev'] else int(pl['L'][0]['n_total'][0]) for m in pl['meta']]
        nrows0 = int(pl['L'][0]['n_rows'][0])
        outs[enc] = [(m['wname'], (nrows0 if m['lev'] == 0 else rows[i]), m['out'][: (nrows0 if m['lev'] == 0 else rows[i])].clone()) for i, m in enumerate(pl['meta'])]
        outs[enc + '_fold'] = [r[:rows[[i for i, m in enumerate(pl['meta']) if m['lev'] == li + 1][-1]]].clone() for li, r in enumerate(pl['rows'])]   # (valid rows only)
        outs[enc + '_tok'] = G.untile_tokens(G.plain(h['last']['ws']['tokens']), int(h['last']['ws']['counters'][0])).clone()
        outs[enc + '_out'] = G.plain(h['last']['ws']['sample_out'][:int(h['last']['ws']['counters'][0])]).clone()
        outs[enc + '_rgb'] = h['rgb'].clone()
    for (name, n, a), (_, _, b) in zip(outs['f16x3'], outs['f16']):
        d = (a - b).abs()
        bad_rows = int((d.max(1)[0] > 1e-2 * a.abs().max()).sum())
        print(f'{name:12s} rows {n:6d} C {a.shape[1]:3d}  |x3| max {float(a.abs().max()):9.3e}  max diff {float(d.max()):9.3e}  rel {float(d.max() / a.abs().max()):8.2e}  rows off by > 1 % {bad_rows}'
              + (f'  first bad rows {torch.nonzero(d.max(1)[0] > 1e-2 * a.abs().max())[:8, 0].tolist()}' if bad_rows else ''))
    for i, (a, b) in enumerate(zip(outs['f16x3_fold'], outs['f16_fold'])):
        print(f'fold level {i}: rel {float((a - b).abs().max() / a.abs().max()):8.2e}')
    ta, tb = outs['f16x3_tok'], outs['f16_tok']
    print('tokens (3 slots x 32): max |x3|', float(ta.abs().max()), 'max diff', float((ta - tb).abs().max()), 'per slot', [float((ta[:, k] - tb[:, k]).abs().max()) for k in range(3)])
    oa, ob = outs['f16x3_out'], outs['f16_out']
    print('per-sample rgb max diff', float((oa[:, :3] - ob[:, :3]).abs().max()), ' sigma+ max diff', float((oa[:, 3].clamp(min=0) - ob[:, 3].clamp(min=0)).abs().max()),
          ' sigma+ max', float(oa[:, 3].clamp(min=0).max()))
    print('image rel diff', float((outs['f16x3_rgb'] - outs['f16_rgb']).abs().max()))


if __name__ == '__main__':
    main()
