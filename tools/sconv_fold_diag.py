"""Diagnostic (round 4) of the single-fp16-product sparse convolutions (SHERF_FRAME_ENCODER_SINGLE: right on the host build, wrong images on
the MI355X): sherf_svox_conv3 called directly -- the pointwise FOLD instances (mode 2: the launches tools/enc_sp_diag.py could not tell
apart) and a submanifold control -- three-product against single-product on the same random rows, per 32-column output tile:
max |diff|, non-finite counts, first bad rows / columns.      python tools/sconv_fold_diag.py [--cpu]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cpu = '--cpu' in sys.argv
    if cpu:
        from tools import cpu_shim
        cpu_shim.enable()
    from sherf_amd import _lib
    from sherf_amd.voxel import pack_conv_weights
    dev = torch.device('cpu' if cpu else 'cuda')
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    g = torch.Generator().manual_seed(0)
    stream = ctypes.c_void_p(0) if cpu else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (1000, 2689, 19245):
        for cin in (32, 64, 96):
            x = torch.randn(n, cin, generator=g).to(dev)
            W = (torch.randn(1, cin, 96, generator=g) / cin ** 0.5).to(dev)
            wp = pack_conv_weights(W)
            bn = torch.stack([1 + 0.1 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)])
            bn = torch.cat([bn, bn[1:].clamp(min=0)]).contiguous().to(dev)             # [3][Cin]: scale, shift, relu(shift)
            nrows = torch.tensor([n], dtype=torch.int32, device=dev)
            outs = {}
            for name, mode in (('x3', 2), ('sp', 2 | 1024), ('x3_half', 2 | 512), ('sp_half', 2 | 512 | 1024)):
                out = torch.full((n + 64, 96), float('nan'), device=dev)
                rc = _lib.lib().sherf_svox_conv3(None, P(nrows), 0, 0, 0, None, 0, 0, 0, P(x), cin, P(bn), None, P(wp), 96, mode, n, P(out), None, stream)
                assert rc == 0, (_lib.lib().sherf_last_error() if hasattr(_lib.lib(), 'sherf_last_error') else rc)
                if not cpu:
                    torch.cuda.synchronize()
                # (fp16 rows: [n][96] halves packed at the start of the buffer)
                outs[name] = out.view(torch.float16).reshape(-1)[:n * 96].view(n, 96).float() if 'half' in name else out[:n]
            ref = (torch.relu(x * bn[0] + bn[1]) @ W[0])
            for a, b in (('x3', 'sp'), ('x3_half', 'sp_half')):
                d = (outs[a] - outs[b]).abs()
                tiles = [float(d[:, 32 * c:32 * c + 32].nan_to_num(1e9).max()) for c in range(3)]
                nf = [int((~torch.isfinite(outs[b][:, 32 * c:32 * c + 32])).sum()) for c in range(3)]
                bad = torch.nonzero(~(d < 0.05 * ref.abs().max()))
                print(f'n {n:6d} Cin {cin:2d} {a} vs {b}: per-tile max diff {tiles}  non-finite in sp {nf}  |ref| max {float(ref.abs().max()):.3f}  '
                      f'x3 vs fp32 {float((outs["x3"] - ref).abs().max()):.2e}' + (f'  first bad (row, col) {bad[:6].tolist()}' if len(bad) else ''))


if __name__ == '__main__':
    main()
