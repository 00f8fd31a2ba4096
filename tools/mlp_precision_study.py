"""Offline study (CPU, oracle only): which MFMA operand formats keep the fused transformer + NeRF decoder inside the 1e-3 parity bar?
Emulates operand quantisation (single, hi/lo split on either side, 3-product split) per linear layer with fp32-grade
accumulation and reports the two figures tests/test_gpu_parity.py::test_per_sample_sigma_rgb gates on.  Result:
only the 3-product splits pass; bf16x1 reproduces the error measured on the MI355X (1.1e-2 / 2.0e-2), so the emulation is
representative.   python tools/mlp_precision_study.py tiny cfg1
"""
import sys, math, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import gpu_common as G
from oracle import sherf_oracle as O
torch.manual_seed(0)

def q(x, dt):
    return x.to(dt).to(torch.float32)
def split(x, dt):
    hi = q(x, dt); lo = q(x - hi, dt); return hi, lo

def mm(a, w, scheme):
    """a [.., K] @ w[N,K].t() with operand quantisation; accumulation in fp64 -> fp32."""
    wt = w.t()
    if scheme == 'fp32': return a @ wt
    kind, dt = scheme
    D = lambda x: x.double()
    if kind == 'x1': r = D(q(a, dt)) @ D(q(wt, dt))
    elif kind == 'x2w':   # act single, weight hi+lo
        wh, wl = split(wt, dt); r = D(q(a, dt)) @ (D(wh) + D(wl))
    elif kind == 'x2a':
        ah, al = split(a, dt); r = (D(ah) + D(al)) @ D(q(wt, dt))
    elif kind == 'x3':
        ah, al = split(a, dt); wh, wl = split(wt, dt); r = D(ah) @ D(wh) + D(ah) @ D(wl) + D(al) @ D(wh)
    elif kind[:3] == 'x3n':
        # main term hi.hi on the fp16 MFMA, the two cross terms (2^-11 of it) on the narrow-format scaled MFMA
        # (v_mfma_scale_f32_32x32x64_f8f6f4): BOTH operands of a cross product then carry m significant bits (fp8 e4m3 / fp6 e2m3: m = 4,
        # fp6 e3m2 / fp4: fewer), the block scale carrying the 2^-11.  Emulated optimistically: every element keeps its own exponent.
        m = int(kind[3:])
        ah, al = split(a, dt); wh, wl = split(wt, dt)
        r = D(ah) @ D(wh) + D(qbits(ah, m)) @ D(qbits(wl, m)) + D(qbits(al, m)) @ D(qbits(wh, m))
    return r.float()

def qbits(x, m):
    """x rounded to m significant bits (round to nearest even), exponent unbounded."""
    mant, ex = torch.frexp(x.double())
    return torch.ldexp(torch.round(mant * (1 << m)) / (1 << m), ex).float()

def run(state, tok, x_c, v_c, sch):
    """sch: dict layer-name -> scheme; default key '*'."""
    S = lambda name: sch.get(name, sch['*'])
    p = 'renderer.transformer.layers.0.'
    h = O._layer_norm(tok, state[p + '0.fn.norm.weight'], state[p + '0.fn.norm.bias'])
    qkv = mm(h, state[p + '0.fn.fn.to_qkv.weight'], S('qkv'))
    n = tok.shape[0]
    qq, k, v = [t.view(n, 3, 3, 16).permute(0, 2, 1, 3) for t in qkv.chunk(3, -1)]
    att = torch.softmax(torch.matmul(qq, k.transpose(-1, -2)) * (16 ** -0.5), -1)
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(n, 3, 48)
    y = mm(o, state[p + '0.fn.fn.to_out.0.weight'], S('out')) + state[p + '0.fn.fn.to_out.0.bias'] + tok
    h = O._layer_norm(y, state[p + '1.fn.norm.weight'], state[p + '1.fn.norm.bias'])
    h = mm(h, state[p + '1.fn.fn.net.0.weight'], S('ff0')) + state[p + '1.fn.fn.net.0.bias']
    h = 0.5 * h * (1 + torch.erf(h / math.sqrt(2.0)))
    z = mm(h, state[p + '1.fn.fn.net.3.weight'], S('ff1')) + state[p + '1.fn.fn.net.3.bias'] + y
    pe_x = O.positional_encoding(x_c, 6); pe_v = O.positional_encoding(v_c, 4)
    d = 'decoder.'
    x0 = torch.cat([pe_x, z[:, 0]], -1); h = x0
    for i in range(8):
        h = torch.relu(mm(h, state[f'{d}pts_linears.{i}.weight'], S(f'L{i}')) + state[f'{d}pts_linears.{i}.bias'])
        if i == 4: h = torch.cat([x0, h], -1)
    sigma = (mm(h, state[d + 'alpha_linear.weight'], S('alpha')) + state[d + 'alpha_linear.bias'])[:, 0]
    f = mm(h, state[d + 'feature_linear.weight'], S('feat')) + state[d + 'feature_linear.bias']
    g = torch.relu(mm(torch.cat([f, pe_v, z[:, 1]], -1), state[d + 'views_linear.weight'], S('views')) + state[d + 'views_linear.bias'])
    rgb = torch.sigmoid(mm(g, state[d + 'rgb_linear.weight'], S('rgb')) + state[d + 'rgb_linear.bias']) * 1.002 - 0.001
    return rgb, sigma

if __name__ == '__main__':
    bf, fp = torch.bfloat16, torch.float16
    for cfg in sys.argv[1:] or ['tiny']:
        state = G.state_for(cfg)                   # (the "_ri" configurations: the reference-init network)
        o = G.oracle_render(cfg)
        tok, x_c, v_c = o['tokens_in'], o['x_c'], o['v_c']
        rgb0, sig0 = run(state, tok, x_c, v_c, {'*': 'fp32'})
        print(cfg, tok.shape[0], 'self-check', float((rgb0 - o['sample_rgb']).abs().max()), float((sig0 - o['sample_sigma']).abs().max()))
        sr = torch.relu(o['sample_sigma'])
        def rep(name, sch):
            rgb, sig = run(state, tok, x_c, v_c, sch)
            print(f"  {name:34s} sigma+ rel-to-max {float((torch.relu(sig) - sr).abs().max() / sr.max()):.2e}  rgb max abs {float((rgb - o['sample_rgb']).abs().max()):.2e}")
        rep('bf16 x1', {'*': ('x1', bf)}); rep('bf16 x3', {'*': ('x3', bf)})
        rep('fp16 x1', {'*': ('x1', fp)}); rep('fp16 x2w', {'*': ('x2w', fp)}); rep('fp16 x2a', {'*': ('x2a', fp)}); rep('fp16 x3', {'*': ('x3', fp)})
        rep('bf16 x2a', {'*': ('x2a', bf)}); rep('bf16 x2w', {'*': ('x2w', bf)})
        for m in (4, 3, 2):
            rep(f'fp16 hi.hi + {m}-bit cross terms', {'*': (f'x3n{m}', fp)})
        names = ['qkv','out','ff0','ff1'] + [f'L{i}' for i in range(8)] + ['alpha','feat','views','rgb']
        print('  one layer fp16 x1, rest fp32:')
        for nm in names: rep('   ' + nm, {'*': 'fp32', nm: ('x1', fp)})
