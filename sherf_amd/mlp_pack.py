"""Host-side packing of the fusion-transformer + NeRFDecoder weights into the MFMA fragment stream that
`sherf_nerf_mlp` (sherf_amd/csrc/mlp.hip) streams through LDS.

A *chunk* is one 32-row output tile of one layer.  Its image is, per 16-wide K-block `kb`, the A operand of
v_mfma_f32_32x32x16_bf16 exactly as a wave reads it: lane l = (row i = l & 31, half h = l >> 5) holds 8 bf16,
element e being W[rows[i], col(kb, h, e)].  `col` encodes where the kernel keeps each input feature:

  'dlay' -- the input is the previous layer's fp32 accumulator tile, converted in registers:
            k-slot (kb, h, e) holds feature  base + 16*kb + (e & 3) + 8*(e >> 2) + 4*h
  'nat'  -- the input was generated lane-locally (positional encodings): feature base + 16*kb + 8*h + e

Every chunk stores a bf16 `hi` image followed by a `lo` image (W - hi, again rounded to bf16); prec 0 reads only
`hi`, prec 1 ("bf16x3") both.  Reference parameter names: renderer.py:271-276, triplane.py:277-283.
"""
import numpy as np

N_CHUNKS = 49


def _bf16_bits(x):
    """fp32 -> bf16 (round to nearest even) as uint16; matches torch .to(bfloat16) / v_cvt_pk_bf16_f32."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint16)


def _bf16_val(x):
    return (_bf16_bits(x).astype(np.uint32) << 16).view(np.float32)


def dlay_feature(kb, h, e):
    return 16 * kb + (e & 3) + 8 * (e >> 2) + 4 * h


def nat_feature(kb, h, e):
    return 16 * kb + 8 * h + e


def chunk_specs(sd, r='renderer.', d='decoder.'):
    """sd: name -> numpy array (fp32).  Returns the 49 chunk specs, in kernel order."""
    g = lambda n: np.asarray(sd[n], dtype=np.float32)
    ident = list(range(32))
    specs = []

    def add(W, rows, segs, bias=None):
        specs.append(dict(W=W, rows=list(rows) + [-1] * (32 - len(rows)), segs=segs, bias=bias))

    Wr = g(r + 'conv1d_reprojection.weight')[:, :, 0]
    add(Wr[:, 32:64], ident, [('nat', 0, 2, 32)])                                           # 0: rgb PE -> slot-2 token
    t = r + 'transformer.layers.0.'
    Wqkv = g(t + '0.fn.fn.to_qkv.weight')
    for rows in (range(0, 32), range(32, 48), range(48, 80), range(80, 112), range(112, 144)):   # 1..5
        add(Wqkv, rows, [('dlay', 0, 2, 32)])
    add(g(t + '0.fn.fn.to_out.0.weight'), ident, [('dlay', 0, 3, 48)], g(t + '0.fn.fn.to_out.0.bias'))   # 6
    add(g(t + '1.fn.fn.net.0.weight'), ident, [('dlay', 0, 2, 32)], g(t + '1.fn.fn.net.0.bias'))         # 7
    add(g(t + '1.fn.fn.net.3.weight'), ident, [('dlay', 0, 2, 32)], g(t + '1.fn.fn.net.3.bias'))         # 8
    x0 = [('nat', 0, 3, 39), ('dlay', 39, 2, 32)]
    for L in range(8):                                                                     # 9..40
        W, b = g(f'{d}pts_linears.{L}.weight'), g(f'{d}pts_linears.{L}.bias')
        segs = x0 if L == 0 else (x0 + [('dlay', 71, 8, 128)] if L == 5 else [('dlay', 0, 8, 128)])
        for T in range(4):
            add(W, range(32 * T, 32 * T + 32), segs, b)
    Wf, bf = g(d + 'feature_linear.weight'), g(d + 'feature_linear.bias')
    for T in range(4):                                                                     # 41..44
        add(Wf, range(32 * T, 32 * T + 32), [('dlay', 0, 8, 128)], bf)
    add(g(d + 'alpha_linear.weight'), [0], [('dlay', 0, 8, 128)], g(d + 'alpha_linear.bias'))   # 45
    Wv, bv = g(d + 'views_linear.weight'), g(d + 'views_linear.bias')
    for T in range(2):                                                                     # 46..47
        add(Wv, range(32 * T, 32 * T + 32), [('dlay', 0, 8, 128), ('nat', 128, 2, 27), ('dlay', 155, 2, 32)], bv)
    add(g(d + 'rgb_linear.weight'), [0, 1, 2], [('dlay', 0, 4, 64)], g(d + 'rgb_linear.bias'))   # 48
    assert len(specs) == N_CHUNKS
    return specs


def chunk_nkb(spec):
    return sum(s[2] for s in spec['segs'])


def chunk_image(spec):
    """fp32 A-operand image [nkb][64 lanes][8]."""
    W = spec['W']
    nkb = chunk_nkb(spec)
    img = np.zeros((nkb, 64, 8), np.float32)
    kb0 = 0
    for kind, base, n, real in spec['segs']:
        for kb in range(n):
            for l in range(64):
                i, h = l & 31, l >> 5
                row = spec['rows'][i]
                if row < 0:
                    continue
                for e in range(8):
                    f = dlay_feature(kb, h, e) if kind == 'dlay' else nat_feature(kb, h, e)
                    if f < real:
                        img[kb0 + kb, l, e] = W[row, base + f]
        kb0 += n
    return img


def bias_table(vec, rows):
    """[2][16] fp32 in accumulator (D) layout: reg r of half h <-> tile row (r&3) + 8*(r>>2) + 4*h."""
    out = np.zeros((2, 16), np.float32)
    if vec is None:
        return out
    for h in range(2):
        for rr in range(16):
            row = rows[(rr & 3) + 8 * (rr >> 2) + 4 * h]
            if row >= 0:
                out[h, rr] = vec[row]
    return out


def pack(sd, r='renderer.', d='decoder.'):
    """-> (stream uint8 [bytes], wbias float32 [(49+4)*32], nkb list)."""
    specs = chunk_specs(sd, r, d)
    parts, bias, nkbs = [], [], []
    for sp in specs:
        img = chunk_image(sp)
        hi = _bf16_bits(img)
        lo = _bf16_bits(img - _bf16_val(img))
        parts.append(hi.tobytes()); parts.append(lo.tobytes())
        bias.append(bias_table(sp['bias'], sp['rows']))
        nkbs.append(img.shape[0])
    t = r + 'transformer.layers.0.'
    for n in ('0.fn.norm.weight', '0.fn.norm.bias', '1.fn.norm.weight', '1.fn.norm.bias'):
        bias.append(bias_table(np.asarray(sd[t + n], np.float32), list(range(32))))
    stream = np.frombuffer(b''.join(parts), dtype=np.uint8).copy()
    return stream, np.stack(bias).reshape(-1).astype(np.float32), nkbs
