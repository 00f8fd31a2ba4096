"""Diagnostic: per-sample difference of the fused gather+MLP launch against the split launches on the device (same frame)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_common as G

for cfg in ('tiny_nv', 'tiny', 'cfg1'):
    ref = G.hip_render(cfg, options=dict(split_gather=True))
    nv = int(ref['last']['ws']['counters'][0])
    so_ref = ref['last']['ws']['sample_out'][:nv].clone()
    tok_ref = ref['last']['ws']['tokens'].clone()
    for rep in range(2):
        got = G.hip_render(cfg)
        so = got['last']['ws']['sample_out'][:nv]
        d = (so - so_ref).abs()
        bad = (d.amax(1) > 0).nonzero().flatten().cpu().numpy()
        nt = (nv + 31) // 32
        tok_same = bool(torch.equal(got['last']['ws']['tokens'][:nt * 3072], tok_ref[:nt * 3072]))
        print(f'{cfg} rep {rep}: nv {nv} tiles {nt} tokens_equal {tok_same} differing samples {len(bad)} max diff rgb {float(d[:, :3].max()):.3e} sigma {float(d[:, 3].max()):.3e} '
              f'image equal {bool(torch.equal(got["rgb"], ref["rgb"]))}')
        if len(bad):
            tiles = np.unique(bad // 32)
            print('   tiles with differences:', tiles[:40].tolist(), '... waves (tile % 4):', np.bincount(tiles % 4, minlength=4).tolist(),
                  ' lanes j:', np.bincount(bad % 32, minlength=32).tolist())
            print('   first differing samples:', bad[:10].tolist(), ' ref', so_ref[bad[0]].tolist(), ' got', so[bad[0]].tolist())
