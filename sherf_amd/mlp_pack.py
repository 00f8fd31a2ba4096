"""Host-side packing of the fusion-transformer + NeRFDecoder weights into the MFMA fragment stream that
`sherf_nerf_mlp` (sherf_amd/csrc/mlp.hip) streams through LDS.

A *chunk* is one 32-row output tile of one layer.  Its image is, per 16-wide K-block `kb`, the A operand of
v_mfma_f32_32x32x16_bf16 exactly as a wave reads it: lane l = (row i = l & 31, half h = l >> 5) holds 8 bf16,
element e being W[rows[i], col(kb, h, e)].  `col` encodes where the kernel keeps each input feature:

  'dlay' -- the input is the previous layer's fp32 accumulator tile, converted in registers:
            k-slot (kb, h, e) holds feature  base + 16*kb + (e & 3) + 8*(e >> 2) + 4*h
  'nat'  -- the input was generated lane-locally (positional encodings): feature base + 16*kb + 8*h + e

The stream is cut into STEPS (what the kernel's 3-slot LDS ring holds at a time): a step is a list of (chunk, K-block) UNITS in
the order the kernel consumes them -- `step_unit` below restates csrc/mlp.hip's table (tests/test_boundary.py compares it with
the library's own export, sherf_mlp_stream_layout) -- each unit being its 1 KiB `hi` fragment followed, for prec 1, by its `lo`
fragment; every step is zero-padded to a multiple of 4 pieces (one DMA round of the workgroup's four waves).
  prec 1 ("f16x3"): hi = fp16(W), lo = fp16(W - hi): 22 significant bits, three MFMAs per product in the kernel.
  prec 0 ("bf16") : hi = bf16(W) only.
Reference parameter names: renderer.py:271-276, triplane.py:277-283.
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


def chunk_specs(sd, r='renderer.', d='decoder.', dtype=np.float32):
    """sd: name -> numpy array.  Returns the 49 chunk specs, in kernel order (dtype int64: `sd` holds element INDICES, see
    stream_index)."""
    g = lambda n: np.asarray(sd[n], dtype=dtype)
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


def chunk_image(spec, fill=0):
    """A-operand image [nkb][64 lanes][8] in the dtype of the spec's matrix (fp32 values, or int64 indices with fill = -1)."""
    W = spec['W']
    nkb = chunk_nkb(spec)
    img = np.full((nkb, 64, 8), fill, W.dtype)
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


def bias_table(vec, rows, fill=0):
    """[2][16] in accumulator (D) layout: reg r of half h <-> tile row (r&3) + 8*(r>>2) + 4*h (fp32 values, or int64 indices with
    fill = -1)."""
    out = np.full((2, 16), fill, np.float32 if vec is None else np.asarray(vec).dtype)
    if vec is None:
        return out
    for h in range(2):
        for rr in range(16):
            row = rows[(rr & 3) + 8 * (rr >> 2) + 4 * h]
            if row >= 0:
                out[h, rr] = vec[row]
    return out


N_STEPS = 43
PRECISIONS = {'bf16': 0, 'f16x3': 1, 'f16': 2}
N_PIECE = {0: 1, 1: 2, 2: 1}          # 1 KiB fragments per unit: hi [, lo]


def step_units(s):
    """units in step s (csrc/mlp.hip: step_units)."""
    return 10 if s < 4 else 8 if s < 20 else (10 if (s - 20) % 3 == 0 else 8) if s < 26 else 8 if s < 42 else 4


def step_unit(s, u):
    """unit u of step s -> (chunk, K-block) or None for padding (csrc/mlp.hip: step_unit)."""
    if s == 0:
        return (0, u) if u < 2 else (1 + (u - 2) // 2, (u - 2) % 2)
    if s == 1:
        return (5, u) if u < 2 else (6, u - 2) if u < 5 else (7, u - 5) if u < 7 else (8, u - 7) if u < 9 else None
    if s < 4:
        return (9 + 2 * (s - 2) + (u & 1), u // 2)
    if s < 20:
        q = s - 4
        return (13 + 4 * (q // 4) + 2 * ((q % 4) // 2) + (u & 1), 4 * (q % 2) + u // 2)
    if s < 26:
        q = s - 20
        return (29 + 2 * (q // 3) + (u & 1), (0, 5, 9)[q % 3] + u // 2)
    if s < 34:
        q = s - 26
        return (33 + 4 * (q // 4) + 2 * ((q % 4) // 2) + (u & 1), 4 * (q % 2) + u // 2)
    if s < 38:
        q = s - 34
        return (41 + 2 * (q // 2) + (u & 1), 4 * (q % 2) + u // 2)
    if s == 38:
        return (45, u)
    if s < 42:
        return (46 + (u & 1), 4 * (s - 39) + u // 2)
    return (48, u)


def step_pieces(s, prec):
    return (step_units(s) * N_PIECE[prec] + 3) // 4 * 4


def _f16_bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)


F16_MAX = 65504.0


def check_f16_range(sd, r='renderer.', d='decoder.', prec=1):
    """The fp16 modes (prec 1, 2) hold weights AND activations as fp16 (hi [+ lo]): a value beyond 65504 would turn into inf - inf =
    NaN silently.  Weights are checked here (non-finite or out of range -> ValueError; pack with prec 0 = bf16, which has fp32's
    range, or rescale).  Activations: the kernel's inputs are encodings in [-1, 1] and O(1) tokens; a layer's output is bounded by
    |x|_inf * max_row |W|_1 + |b|, so the product of those row norms bounds every activation -- it must stay below the fp16 range
    too (a trained 8 x 128 ReLU decoder sits many orders below: ~1e1 per layer would be needed to reach it)."""
    if prec == 0:
        return
    bound = 64.0                    # generous bound on |input|_inf: PE in [-1, 1], LayerNorm'd tokens, |x_c| of a few metres
    for name, v in sd.items():
        if not (name.startswith(d) or name.startswith(r + 'transformer') or name.startswith(r + 'conv1d_reprojection')):
            continue
        a = np.asarray(v, dtype=np.float32)
        if not np.isfinite(a).all():
            raise ValueError(f'{name}: non-finite values; the MLP weight stream cannot be packed')
        if a.size and float(np.abs(a).max()) > F16_MAX:
            raise ValueError(f'{name}: |w| up to {float(np.abs(a).max()):.3g} exceeds the fp16 range of mlp_precision f16x3 / f16; use bf16')
    # a-priori bound of the activations: loose by orders of magnitude on real weights (it assumes every sign aligned), so it is only
    # used to REJECT the absurd -- per-layer gains that let it reach 1e30 mean activations can really leave the fp16 range
    for lname in [f'{d}pts_linears.{i}' for i in range(8)] + [d + 'feature_linear', d + 'views_linear']:
        W = np.abs(np.asarray(sd[lname + '.weight'], np.float64)).sum(1).max()
        bound = bound * W + np.abs(np.asarray(sd[lname + '.bias'], np.float64)).max()
    if bound > 1e30:
        raise ValueError(f'decoder weight norms allow activations up to ~{bound:.1e}: outside the fp16 range of mlp_precision f16x3 / f16; use bf16')


def raise_for_flags(flag, bound, prec):
    """The device pack's checks (sherf_mlp_pack_stream's flag word, the a-priori activation bound) -> check_f16_range's errors."""
    if flag & 1:
        raise ValueError('non-finite values in the transformer / decoder weights; the MLP weight stream cannot be packed')
    if prec != 0 and (flag & 2):
        raise ValueError('a transformer / decoder weight exceeds the fp16 range of mlp_precision f16x3 / f16; use bf16')
    if prec != 0 and bound > 1e30:
        raise ValueError(f'decoder weight norms allow activations up to ~{bound:.1e}: outside the fp16 range of mlp_precision f16x3 / f16; use bf16')


def pack(sd, r='renderer.', d='decoder.', prec=1):
    """-> (stream uint8 [bytes], wbias float32 [(49+4)*32], nkb list) for the kernel's `prec` (1 = f16x3, 0 = bf16, 2 = f16)."""
    assert prec in (0, 1, 2)
    check_f16_range(sd, r, d, prec)
    specs = chunk_specs(sd, r, d)
    frag, bias, nkbs = [], [], []
    for sp in specs:
        img = chunk_image(sp)
        if prec == 1:
            hi = _f16_bits(img)
            lo = _f16_bits(img - hi.view(np.float16).astype(np.float32))
        elif prec == 2:
            hi, lo = _f16_bits(img), None
        else:
            hi, lo = _bf16_bits(img), None
        frag.append((hi, lo))
        bias.append(bias_table(sp['bias'], sp['rows']))
        nkbs.append(img.shape[0])
    parts, used = [], set()
    for s_ in range(N_STEPS):
        n = 0
        for u in range(step_units(s_)):
            cu = step_unit(s_, u)
            if cu is None:
                parts.append(bytes(1024 * N_PIECE[prec]))
            else:
                c, kb = cu
                assert kb < nkbs[c] and cu not in used, (s_, u, cu)
                used.add(cu)
                parts.append(frag[c][0][kb].tobytes())
                if prec == 1:
                    parts.append(frag[c][1][kb].tobytes())
            n += N_PIECE[prec]
        parts.append(bytes(1024 * (step_pieces(s_, prec) - n)))
    assert len(used) == sum(nkbs), 'every (chunk, K-block) unit must be streamed exactly once'
    t = r + 'transformer.layers.0.'
    for n in ('0.fn.norm.weight', '0.fn.norm.bias', '1.fn.norm.weight', '1.fn.norm.bias'):
        bias.append(bias_table(np.asarray(sd[t + n], np.float32), list(range(32))))
    stream = np.frombuffer(b''.join(parts), dtype=np.uint8).copy()
    return stream, np.stack(bias).reshape(-1).astype(np.float32), nkbs


# ---- the same stream, packed ON THE DEVICE -----------------------------------------------------------------------------------------
# The stream is a pure gather of weight elements (plus rounding), so its layout can be computed once per architecture as an index map
# and applied by a kernel (csrc/mlp.hip: sherf_mlp_pack_stream) every time the weights change: a training step repacks after every
# optimiser update, and the Python loops of chunk_image above cost ~45 ms per call (round 2's training step: 48 ms of its 132 ms).
_NORM_NAMES = ('0.fn.norm.weight', '0.fn.norm.bias', '1.fn.norm.weight', '1.fn.norm.bias')


def packed_names(r='renderer.', d='decoder.'):
    """Parameters the stream and the bias table read, in the order of the flat vector the device pack takes."""
    t = r + 'transformer.layers.0.'
    names = [r + 'conv1d_reprojection.weight', t + '0.fn.fn.to_qkv.weight', t + '0.fn.fn.to_out.0.weight', t + '0.fn.fn.to_out.0.bias',
             t + '1.fn.fn.net.0.weight', t + '1.fn.fn.net.0.bias', t + '1.fn.fn.net.3.weight', t + '1.fn.fn.net.3.bias']
    for L in range(8):
        names += [f'{d}pts_linears.{L}.weight', f'{d}pts_linears.{L}.bias']
    for n in ('feature_linear', 'alpha_linear', 'views_linear', 'rgb_linear'):
        names += [d + n + '.weight', d + n + '.bias']
    return names + [t + n for n in _NORM_NAMES]


_index_cache = {}


def stream_index(shapes, r='renderer.', d='decoder.', prec=1):
    """shapes: name -> shape of every parameter of packed_names().  -> (src int32 [2-byte slots of the stream], bias_src int32
    [(49+4)*32]): slot i of the stream holds piece (src[i] & 1) -- 0 = hi, 1 = lo -- of element src[i] >> 1 of the flat vector
    (the parameters of packed_names() concatenated), or zero padding where src[i] < 0; bias_src likewise (fp32, no pieces).
    Built by running the host packer above on element indices instead of values: one statement of the layout."""
    names = packed_names(r, d)
    key = (prec, r, d, tuple(tuple(shapes[n]) for n in names))
    if key in _index_cache:
        return _index_cache[key]
    sd, off = {}, 0
    for n in names:
        k = int(np.prod(shapes[n]))
        sd[n] = np.arange(off, off + k, dtype=np.int64).reshape(tuple(shapes[n]))
        off += k
    specs = chunk_specs(sd, r, d, dtype=np.int64)
    imgs = [chunk_image(sp, fill=-1) for sp in specs]                 # [nkb][64][8] element indices
    npc = N_PIECE[prec]
    pad = np.full(512, -1, np.int64)
    parts = []
    for s_ in range(N_STEPS):
        n = 0
        for u in range(step_units(s_)):
            cu = step_unit(s_, u)
            for q in range(npc):
                if cu is None:
                    parts.append(pad)
                else:
                    e = imgs[cu[0]][cu[1]].reshape(-1)
                    parts.append(np.where(e >= 0, 2 * e + q, -1))
            n += npc
        parts += [pad] * (step_pieces(s_, prec) - n)
    src = np.concatenate(parts).astype(np.int32)
    bias = [bias_table(sp['bias'], sp['rows'], fill=-1).astype(np.int64) if sp['bias'] is not None else np.full((2, 16), -1, np.int64)
            for sp in specs]
    t = r + 'transformer.layers.0.'
    bias += [bias_table(sd[t + n], list(range(32)), fill=-1) for n in _NORM_NAMES]
    out = (src, np.stack(bias).reshape(-1).astype(np.int32), off)
    _index_cache[key] = out
    return out


def pack_from_index(flat, src, bias_src, prec):
    """The index map applied on the host (numpy): what sherf_mlp_pack_stream computes on the device.  tests: == pack()."""
    flat = np.asarray(flat, np.float32)
    v = np.where(src >= 0, flat[np.maximum(src, 0) >> 1], np.float32(0)).astype(np.float32)
    if prec == 0:
        bits = _bf16_bits(v)
    else:
        hi = v.astype(np.float16)
        bits = np.where((src & 1) == 1, (v - hi.astype(np.float32)).astype(np.float16).view(np.uint16), hi.view(np.uint16))
        bits = np.where(src >= 0, bits, 0).astype(np.uint16)
    wb = np.where(bias_src >= 0, flat[np.maximum(bias_src, 0)], np.float32(0)).astype(np.float32)
    return bits.view(np.uint8), wb
