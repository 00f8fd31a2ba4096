"""The dense GEMMs of the backward, one shape at a time: the streaming kernel (tall_stream_kernel) against the general one (SHERF_EXPERIMENT
bit 6), same operands, bitwise comparison, us per call and the two bounds (bytes at 8 TB/s, six bf16 products per term at 2.5 PFLOP/s).  GPU box only:

    python tools/bwd_gemm_ab.py [--rows 750000]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=750000)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    from sherf_amd import _lib
    from sherf_amd.backward_dense import HipOps, Mat
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    ops = HipOps()
    n = a.rows
    g = torch.Generator(device='cpu').manual_seed(0)

    def mat(r, c, ld=None, scale=1.0):
        ld = ld or c
        return Mat((torch.randn(r * ld, generator=g) * scale).to(dev), r, c, ld)
    # (label, tB, M, K, N, lda, ldc, bias+relu)
    shapes = [('fwd  128->128', 1, n, 128, 128, 128, 128, True), ('dgrad 128->128', 0, n, 128, 128, 128, 128, False), ('dgrad 128->71', 0, n, 128, 71, 128, 71, False),
              ('dgrad 128->199', 0, n, 128, 199, 128, 199, False), ('fwd  into cat5 (ldc 199)', 1, n, 128, 128, 128, 199, True), ('dgrad views 64->187', 0, n, 64, 187, 64, 187, False),
              ('qkv 32->144 (3n)', 1, 3 * n, 32, 144, 32, 144, False), ('to_out 48->32 (3n)', 1, 3 * n, 48, 32, 48, 32, True), ('ff 32->32 (3n)', 1, 3 * n, 32, 32, 32, 32, True),
              ('dgrad qkv 144->32 (3n)', 0, 3 * n, 144, 32, 144, 32, False), ('dgrad to_out 32->48 (3n)', 0, 3 * n, 32, 48, 32, 48, False)]
    def set_debug(v):
        os.environ['SHERF_EXPERIMENT'] = '64' if v else '0'
    for label, tB, M, K, N, lda, ldc, ba in shapes:
        A = mat(M, K, lda, 1e-3)
        B = mat(N, K) if tB else mat(K, N)
        bias = mat(1, N) if ba else None
        outs, times = [], []
        for dbg in (1, 0):
            set_debug(dbg)
            C = Mat(torch.full((M * ldc,), float('nan'), device=dev), M, N, ldc)
            run = (lambda: ops.gemm_bias_act(0, tB, A, B, C, bias, 1)) if ba else (lambda: ops.gemm(0, tB, A, B, C))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record(); torch.cuda.synchronize()
            times.append(1e3 * e0.elapsed_time(e1) / a.iters)
            outs.append(C.tensor().clone())
        set_debug(0)
        same = torch.equal(outs[0], outs[1])
        hbm = (M * K + M * N) * 4 / 8e12 * 1e6
        mfma = 2.0 * M * (-(-N // 32) * 32) * K * 6 / 2.5e15 * 1e6
        print(f'[gemm] {label:28s} M {M:8d} K {K:3d} N {N:3d}: general {times[0]:7.1f} us  streaming {times[1]:7.1f} us  ({times[0] / times[1]:.2f}x)  identical bits: {same}   '
              f'bounds: bytes {hbm:6.1f} us, MFMA {mfma:6.1f} us', flush=True)


if __name__ == '__main__':
    main()
