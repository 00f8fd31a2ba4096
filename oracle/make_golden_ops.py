"""TEST INFRASTRUCTURE. Writes tests/golden/ops.npz by running the UNMODIFIED reference's `_bias_act_ref` (torch_utils/ops/
bias_act.py:93) and `_upfirdn2d_ref` (torch_utils/ops/upfirdn2d.py:139) -- the reference's own specification of its two custom CUDA
operators -- on the cases of oracle/ops_cases.py, together with their first and (bias_act) second derivatives obtained by autograd
through those functions.  Build container only (/root/reference);   python -m oracle.make_golden_ops"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/sherf'


def main():
    sys.path.insert(0, REF)
    from torch_utils.ops import bias_act as RB, upfirdn2d as RU      # the reference's files, untouched
    from oracle import ops_cases as C
    out = {}
    for case in C.BIAS_ACT_CASES:
        name, shape, dim, act, alpha, gain, clamp, with_b = case
        x, b, dy, ddx = C.bias_act_inputs(case)
        xt = torch.from_numpy(x).requires_grad_(True)
        bt = torch.from_numpy(b).requires_grad_(True) if b is not None else None
        dyt = torch.from_numpy(dy).requires_grad_(True)
        y = RB._bias_act_ref(xt, bt, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
        gs = torch.autograd.grad(y, [xt] + ([bt] if bt is not None else []), dyt, create_graph=True)
        out[name + '.y'] = y.detach().numpy()
        out[name + '.dx'] = gs[0].detach().numpy()
        if bt is not None:
            out[name + '.db'] = gs[1].detach().numpy()
        if gs[0].requires_grad:
            g2 = torch.autograd.grad(gs[0], [xt, dyt], torch.from_numpy(ddx), allow_unused=True)
            out[name + '.d2_x'] = (g2[0] if g2[0] is not None else torch.zeros_like(xt)).numpy()
            out[name + '.d2_dy'] = g2[1].numpy()
    for case in C.UPFIRDN_CASES:
        name, shape, fspec, up, down, pad, flip, gain = case
        x, f, dy_seed = C.upfirdn_inputs(case)
        xt = torch.from_numpy(x).requires_grad_(True)
        ft = torch.from_numpy(f) if f is not None else None
        y = RU._upfirdn2d_ref(xt, ft, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        dy = torch.from_numpy(np.random.RandomState(dy_seed).standard_normal(tuple(y.shape)).astype(np.float32))
        out[name + '.y'] = y.detach().numpy()
        out[name + '.dx'] = torch.autograd.grad(y, xt, dy)[0].numpy()
    # the convenience wrappers' padding rules (upfirdn2d.py:279-389), on the StyleGAN2 blur filter
    x = torch.from_numpy(C.rng('wrappers').standard_normal((1, 2, 8, 6)).astype(np.float32))
    f = RU.setup_filter([1, 3, 3, 1])
    out['wr.filter'] = f.numpy()
    out['wr.filter2d'] = RU.filter2d(x, f, impl='ref').numpy()
    out['wr.upsample2d'] = RU.upsample2d(x, f, up=2, impl='ref').numpy()
    out['wr.downsample2d'] = RU.downsample2d(x, f, down=2, impl='ref').numpy()
    out['wr.sep'] = RU.setup_filter(list(range(1, 9))).numpy()
    path = os.path.join(HERE, '..', 'tests', 'golden', 'ops.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, len(out), 'arrays', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
