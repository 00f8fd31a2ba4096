"""What does the board's power cap leave of the fp16 MFMA peak for a LIBRARY GEMM?  (GPU box only)

    python tools/gemm_ceiling.py

Times torch.matmul (hipBLASLt / rocBLAS underneath; not part of the product: a yardstick for DESIGN 5.1) in fp16 with fp32 accumulation on
  * a large square product (8192^3: the vendor's own best case), and
  * the network's shape: [1 268 000, 128] x [128, 128] (one decoder layer over the bench frame's valid samples),
each on standard-normal operands and on zeros (same instruction stream, no switching: the power-cap test of tools/mlp_ab.py --zero),
back to back for ~2 s so that the power management settles, and prints TFLOP/s and the fraction of the 2 516 TFLOP/s dense fp16 peak."""
import time

import torch

PEAK = 2516.0


def bench(a, b, secs=2.0):
    for _ in range(20):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            torch.matmul(a, b)
        e1.record(); torch.cuda.synchronize()
        n += 50
        last = e0.elapsed_time(e1) / 50
    return last


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    for name, (M, K, N) in (('square 8192^3', (8192, 8192, 8192)), ('square 4096^3', (4096, 4096, 4096)), ('network layer [1268000,128]x[128,128]', (1268000, 128, 128)),
                            ('network layer, 8 layers deep K (one [1268000,1024]x[1024,128])', (1268000, 1024, 128))):
        for data in ('normal', 'zeros'):
            a = (torch.randn(M, K, device=dev) if data == 'normal' else torch.zeros(M, K, device=dev)).half()
            b = (torch.randn(K, N, device=dev) if data == 'normal' else torch.zeros(K, N, device=dev)).half()
            ms = bench(a, b)
            tf = 2.0 * M * K * N / (ms * 1e-3) / 1e12
            hbm = (M * K + K * N + M * N) * 2 / (ms * 1e-3) / 1e9
            print(f'[gemm] {name:70s} {data:6s} {ms:8.4f} ms  {tf:7.1f} TFLOP/s = {tf / PEAK:.3f} of the fp16 peak   ({hbm:6.0f} GB/s of operands + result)', flush=True)
            del a, b


if __name__ == '__main__':
    main()
