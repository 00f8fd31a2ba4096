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
    # weight gradients dW[M,N] = dy[rows,M]^T . x[rows,N]: the shared-B / solo kernels against round 2's (SHERF_EXPERIMENT bit 7)
    wshapes = [('dW 128x128', n, 128, 128), ('dW 128x71', n, 128, 71), ('dW 128x199', n, 128, 199), ('dW 64x187 (views)', n, 64, 187), ('dW 144x32 (qkv, 3n)', 3 * n, 144, 32),
               ('dW 32x32 (ff, 3n)', 3 * n, 32, 32), ('dW 32x48 (to_out, 3n)', 3 * n, 32, 48), ('dW 3x64 (rgb)', n, 3, 64), ('dW 1x128 (alpha)', n, 1, 128)]
    for label, rows, M, N in wshapes:
        dy, x = mat(rows, M, None, 1e-3), mat(rows, N)
        outs, times = [], []
        for old in (1, 0):
            os.environ['SHERF_EXPERIMENT'] = '128' if old else '0'
            C = Mat(torch.zeros(M * N, device=dev), M, N)
            run = lambda: ops.gemm(1, 0, dy, x, C)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record(); torch.cuda.synchronize()
            times.append(1e3 * e0.elapsed_time(e1) / a.iters)
            outs.append(C.tensor().double().cpu())
        os.environ['SHERF_EXPERIMENT'] = '0'
        ref = (dy.tensor()[:200000].double().t() @ x.tensor()[:200000].double()).cpu() if rows > 200000 else None
        rel = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        hbm = rows * (M + N) * 4 / 8e12 * 1e6
        print(f'[wgrad] {label:24s} rows {rows:8d}: round 2 {times[0]:7.1f} us  round 5 {times[1]:7.1f} us  ({times[0] / times[1]:.2f}x)  max rel difference {rel:.1e}   bound: bytes {hbm:6.1f} us', flush=True)
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
