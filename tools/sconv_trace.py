"""In-kernel timeline of the sparse-convolution launches of one frame (profiling build, SHERF_SCONV_TRACE):

    bash tools/build_variants.sh sconvtrace
    gpurun -- 'SHERF_HIP_LIB=$PWD/sherf_amd/libsherf_hip_sconvtrace.so python tools/sconv_trace.py > gpurun_out/sconv_trace.log'

Every workgroup's wave 0 stamps s_memtime at its phase boundaries (csrc/svox.hip); this script renders the bench frame (cfg2), once
with the ray side running beside the encoder (the product schedule) and once with the ray side held back until the encoder is done
(`main_after_layer`), and prints per launch: workgroups, rows, the kernel's span (first start -> last end) and the mean cycles a
workgroup spends in each phase.
"""
import argparse
import ctypes as ct
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = ('nbr_table', 'bn_prologue', 'sync', 'tapmask+row0', 'first_tap', 'other_taps', 'lds_write', 'sync_all_waves', 'reduce+store+stats')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2')
    a0 = ap.parse_args()
    import bench
    from sherf_amd import _lib
    dev = torch.device('cuda', 0)
    a = argparse.Namespace(config=a0.config, precision='f16x3', bn_mode='train', exact_grids=False)
    w = bench.make_workload(a, 0.4, dev)
    lib = ct.CDLL(_lib.LIB_PATH)
    lib.sherf_sconv_set_trace.argtypes = [ct.c_void_p, ct.c_uint]
    cap = 1 << 16
    buf = torch.zeros(cap * 16, dtype=torch.int32, device=dev)
    for _ in range(3):
        bench.render_frame(w)
    torch.cuda.synchronize()
    for label, after in (('beside the ray side (product schedule)', -1), ('alone (ray side held back until the encoder is done)', 99)):
        w['opts']['main_after_layer'] = after
        bench.render_frame(w); torch.cuda.synchronize()
        buf.zero_(); torch.cuda.synchronize()
        assert lib.sherf_sconv_set_trace(buf.data_ptr(), cap) == 0
        bench.render_frame(w); torch.cuda.synchronize()
        assert lib.sherf_sconv_set_trace(None, 0) == 0
        r = buf.cpu().numpy().astype(np.int64).reshape(cap, 16) & 0xffffffff
        r = r[r[:, 4] != 0]
        print(f'== encoder {label}: {len(r)} workgroups traced')
        print(f'{"launch":>6} {"NCOT,NKB":>8} {"mode":>4} {"rows":>6} {"wgs":>5} {"span_us@2GHz":>12} {"wg_total":>9}  ' + '  '.join(f'{p:>12}' for p in PHASES) + '  taps(w0)')
        t_first = None
        for lid in np.unique(r[:, 0]):
            q = r[r[:, 0] == lid]
            st = q[:, 4:14]
            d = (st[:, 1:] - st[:, :-1]) & 0xffffffff
            tot = (st[:, 9] - st[:, 0]) & 0xffffffff
            s_min = st[:, 0].min(); e_max = st[:, 9].max()
            t_first = s_min if t_first is None else t_first
            tag = int(q[0, 3])
            print(f'{int(lid):6d} {(tag >> 8) & 0xff:>5},{tag & 0xff:<2} {(tag >> 16) & 0xff:>4} {int(q[0, 2]):6d} {len(q):5d} '
                  f'{((e_max - s_min) & 0xffffffff) / 2000.0:12.1f} {tot.mean():9.0f}  ' + '  '.join(f'{x:12.0f}' for x in d.mean(0)) + f'  {(q[:, 3] >> 24).mean():.1f}'
                  f'   start +{((s_min - t_first) & 0xffffffff) / 2000.0:.0f} us')


if __name__ == '__main__':
    main()
