"""GPU diagnostic (round 3): layer-by-layer comparison of the sparse encoder's raw outputs between the f16x3 and the single-product
(SHERF_FRAME_ENCODER_SINGLE) convolutions on one frame -- where do they part?   python tools/enc_sp_diag.py [cfg]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_common as G


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'tiny_ri'
    outs = {}
    for enc in ('f16x3', 'f16'):
        h = G.hip_render(cfg, precision='f16', options=dict(table_precision='f32', encoder_precision=enc))
        torch.cuda.synchronize()
        pl = h['last']['plan']
        rows = [int(pl['L'][m['lev']]['n_rows'][0]) if m['lev'] else int(pl['L'][0]['n_total'][0]) for m in pl['meta']]
        nrows0 = int(pl['L'][0]['n_rows'][0])
        outs[enc] = [(m['wname'], (nrows0 if m['lev'] == 0 else rows[i]), m['out'][: (nrows0 if m['lev'] == 0 else rows[i])].clone()) for i, m in enumerate(pl['meta'])]
        outs[enc + '_fold'] = [r[:].clone() for r in pl['rows']]
        outs[enc + '_rgb'] = h['rgb'].clone()
    for (name, n, a), (_, _, b) in zip(outs['f16x3'], outs['f16']):
        d = (a - b).abs()
        bad_rows = int((d.max(1)[0] > 1e-2 * a.abs().max()).sum())
        print(f'{name:12s} rows {n:6d} C {a.shape[1]:3d}  |x3| max {float(a.abs().max()):9.3e}  max diff {float(d.max()):9.3e}  rel {float(d.max() / a.abs().max()):8.2e}  rows off by > 1 % {bad_rows}'
              + (f'  first bad rows {torch.nonzero(d.max(1)[0] > 1e-2 * a.abs().max())[:8, 0].tolist()}' if bad_rows else ''))
    for i, (a, b) in enumerate(zip(outs['f16x3_fold'], outs['f16_fold'])):
        print(f'fold level {i}: rel {float((a - b).abs().max() / a.abs().max()):8.2e}')
    print('image rel diff', float((outs['f16x3_rgb'] - outs['f16_rgb']).abs().max()))


if __name__ == '__main__':
    main()
