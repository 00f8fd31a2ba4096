"""CPU checks of the host logic around the MFMA kernel: the packed weight stream (sherf_amd/mlp_pack.py) is
consumed by a numpy emulation of the kernel's register dataflow (hardware C/D layout of v_mfma_f32_32x32x16:
lane = (col j, half h), reg r <-> row (r&3) + 8*(r>>2) + 4*h) and must reproduce the oracle's
transformer + decoder outputs."""
import json
import os
import numpy as np
import pytest
import torch

from oracle import fixtures, sherf_oracle as O
from sherf_amd import mlp_pack


def _state(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    return {n: fixtures.seeded_param(n, s) for n, s in shapes.items() if fixtures.seeded_param(n, s) is not None}


def _row_of(h, r):
    return (r & 3) + 8 * (r >> 2) + 4 * h


class Emu:
    """Emulates nerf_mlp_kernel for n samples; activations are [n, 32] tiles in natural row order."""

    def __init__(self, stream, wbias, nkbs, use_lo=True, prec=1):
        """Reads the STEPPED stream back (mlp_pack.step_unit order: hi [, lo] piece per unit, steps padded to 4 pieces)."""
        self.bias = wbias.reshape(-1, 2, 16)
        self.img = [np.zeros((nkb, 64, 8)) for nkb in nkbs]
        raw = np.frombuffer(stream.tobytes(), np.uint16).reshape(-1, 512)            # 1 KiB pieces of 64 lanes x 8 elements
        dec = (lambda p: p.view(np.float16).astype(np.float64)) if prec == 1 else (lambda p: (p.astype(np.uint32) << 16).view(np.float32).astype(np.float64))
        piece = 0
        for s_ in range(mlp_pack.N_STEPS):
            for u in range(mlp_pack.step_units(s_)):
                cu = mlp_pack.step_unit(s_, u)
                if cu is not None:
                    w = dec(raw[piece])
                    if prec == 1 and use_lo:
                        w = w + dec(raw[piece + 1])
                    self.img[cu[0]][cu[1]] = w.reshape(64, 8)
                piece += prec + 1
            piece = sum(mlp_pack.step_pieces(i, prec) for i in range(s_ + 1))
        assert piece * 1024 == stream.nbytes

    @staticmethod
    def tile_frags(tile):
        """accumulator tile [n,32] -> two K-blocks [n, 2(h), 8(e)] (split_tile in the kernel)."""
        out = []
        for kbl in range(2):
            b = np.zeros((tile.shape[0], 2, 8))
            for h in range(2):
                for e in range(8):
                    b[:, h, e] = tile[:, _row_of(h, 8 * kbl + e)]
            out.append(b)
        return out

    @staticmethod
    def half_frag(v16):
        """16 values of one head (natural dim order) -> one K-block, as kept in 8 regs of a half tile."""
        b = np.zeros((v16.shape[0], 2, 8))
        for h in range(2):
            for e in range(8):
                b[:, h, e] = v16[:, _row_of(h, e)]
        return b

    @staticmethod
    def nat_frags(feats, nkb):
        f = np.zeros((feats.shape[0], nkb * 16)); f[:, :feats.shape[1]] = feats
        return [f[:, 16 * kb:16 * kb + 16].reshape(-1, 2, 8) for kb in range(nkb)]

    def mma(self, c, frags):
        img = self.img[c]
        assert len(frags) == img.shape[0], (c, len(frags), img.shape)
        out = np.zeros((frags[0].shape[0], 32))
        for h in range(2):
            for r in range(16):
                out[:, _row_of(h, r)] = self.bias[c, h, r]
        for kb, b in enumerate(frags):
            for h in range(2):
                out += b[:, h, :] @ img[kb, 32 * h:32 * h + 32, :].T
        return out

    def ln(self, x, idx):
        g = np.zeros(32); bt = np.zeros(32)
        for h in range(2):
            for r in range(16):
                g[_row_of(h, r)] = self.bias[mlp_pack.N_CHUNKS + 2 * idx, h, r]
                bt[_row_of(h, r)] = self.bias[mlp_pack.N_CHUNKS + 2 * idx + 1, h, r]
        mu = x.mean(1, keepdims=True); var = ((x - mu) ** 2).mean(1, keepdims=True)
        return (x - mu) / np.sqrt(var + 1e-5) * g + bt

    def run(self, tok, rgb, x_c, v_c):
        pe = lambda x, F: O.positional_encoding(torch.from_numpy(x).float(), F).numpy().astype(np.float64)
        tok = tok.astype(np.float64).copy()
        tok[:, 2] += self.mma(0, self.nat_frags(pe(rgb, 5)[:, :32], 2))
        ln = [self.tile_frags(self.ln(tok[:, t], 0)) for t in range(3)]
        q01 = [self.mma(1, ln[i]) for i in range(2)]
        q2 = [self.mma(2, ln[i]) for i in range(2)]
        k01 = [self.mma(3, ln[t]) for t in range(3)]
        k2v0 = [self.mma(4, ln[t]) for t in range(3)]
        v12 = [self.mma(5, ln[t]) for t in range(3)]
        q = [[q01[i][:, :16], q01[i][:, 16:], q2[i][:, :16]] for i in range(2)]
        k = [[k01[t][:, :16], k01[t][:, 16:], k2v0[t][:, :16]] for t in range(3)]
        v = [[k2v0[t][:, 16:], v12[t][:, :16], v12[t][:, 16:]] for t in range(3)]
        y = []
        for i in range(2):
            frs = []
            for hd in range(3):
                d = np.stack([(q[i][hd] * k[t][hd]).sum(1) for t in range(3)], 1) * 0.25
                p = np.exp(d - d.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
                frs.append(self.half_frag(sum(p[:, t:t + 1] * v[t][hd] for t in range(3))))
            y.append(self.mma(6, frs) + tok[:, i])
        z = []
        for i in range(2):
            a = self.mma(7, self.tile_frags(self.ln(y[i], 1)))
            a = 0.5 * a * (1 + np.vectorize(__import__('math').erf)(a / np.sqrt(2.0)))
            z.append(self.mma(8, self.tile_frags(a)) + y[i])
        x0 = self.nat_frags(pe(x_c, 6), 3) + self.tile_frags(z[0])
        hcur = [np.maximum(self.mma(9 + T, x0), 0) for T in range(4)]
        c = 13
        for L in range(1, 8):
            fr = sum((self.tile_frags(t) for t in hcur), [])
            if L == 5:
                fr = x0 + fr
            hcur = [np.maximum(self.mma(c + T, fr), 0) for T in range(4)]
            c += 4
        fr = sum((self.tile_frags(t) for t in hcur), [])
        feat = [self.mma(41 + T, fr) for T in range(4)]
        sigma = self.mma(45, fr)[:, 0]
        vin = sum((self.tile_frags(t) for t in feat), []) + self.nat_frags(pe(v_c, 4), 2) + self.tile_frags(z[1])
        g = [np.maximum(self.mma(46 + T, vin), 0) for T in range(2)]
        o = self.mma(48, sum((self.tile_frags(t) for t in g), []))[:, :3]
        return 1 / (1 + np.exp(-o)) * 1.002 - 0.001, sigma


@pytest.fixture(scope='module')
def packed(golden_dir):
    sd = _state(golden_dir)
    return sd, mlp_pack.pack(sd)


def test_stream_layout(packed):
    sd, (stream, wbias, nkbs) = packed
    assert len(nkbs) == 49 and sum(nkbs) == 351
    assert stream.nbytes == 704 * 1024 and sum(mlp_pack.step_pieces(s, 1) for s in range(mlp_pack.N_STEPS)) == 704
    assert max(mlp_pack.step_pieces(s, 1) for s in range(mlp_pack.N_STEPS)) == 20          # the kernel's LDS ring slot: 20 KiB
    assert all(mlp_pack.step_pieces(s, p) % 4 == 0 for s in range(mlp_pack.N_STEPS) for p in (0, 1))   # whole DMA rounds of 4 waves
    units = [mlp_pack.step_unit(s, u) for s in range(mlp_pack.N_STEPS) for u in range(mlp_pack.step_units(s))]
    real = [u for u in units if u is not None]
    assert len(real) == 351 and len(set(real)) == 351 and units.count(None) == 1         # every (chunk, K-block) exactly once + one pad
    s0, _, _ = mlp_pack.pack(sd, prec=0)
    assert s0.nbytes == 1024 * sum(mlp_pack.step_pieces(s, 0) for s in range(mlp_pack.N_STEPS))
    assert wbias.size == (49 + 4) * 32
    assert nkbs[:9] == [2, 2, 2, 2, 2, 2, 3, 2, 2] and nkbs[9] == 5 and nkbs[29] == 13 and nkbs[46] == 12 and nkbs[48] == 4


@pytest.mark.parametrize('prec', [0, 1, 2])
def test_index_map_reproduces_the_host_packer(packed, prec):
    """mlp_pack.stream_index (the layout as an element-index map, applied by sherf_mlp_pack_stream on the device) == pack()."""
    sd, _ = packed
    stream, wbias, _ = mlp_pack.pack(sd, prec=prec)
    names = mlp_pack.packed_names()
    src, bsrc, n_flat = mlp_pack.stream_index({n: sd[n].shape for n in names}, prec=prec)
    flat = np.concatenate([np.asarray(sd[n], np.float32).reshape(-1) for n in names])
    assert flat.size == n_flat and src.size * 2 == stream.nbytes and bsrc.size == wbias.size
    s2, wb2 = mlp_pack.pack_from_index(flat, src, bsrc, prec)
    assert np.array_equal(s2, stream) and np.array_equal(wb2, wbias)
    used = np.unique(src[src >= 0] >> 1)
    assert used.size > 0.95 * (n_flat - 4 * 32 - sum(sd[n].size for n in names if n.endswith('.bias')))   # (weights: all but a few unused columns)


def test_stream_reproduces_oracle_network(packed):
    sd, (stream, wbias, nkbs) = packed
    state = {k: torch.from_numpy(v) for k, v in sd.items()}
    fx = fixtures.renderer_inputs('tiny')
    r = O.render_from_fixture(fx, state, training=True)
    Wb = sd['renderer.conv1d_reprojection.weight'][:, 32:64, 0]
    tok = r['tokens_in'].numpy().copy()
    tok[:, 2] -= O.positional_encoding(r['tap_rgb'], 5)[:, :32].numpy() @ Wb.T
    emu = Emu(stream, wbias, nkbs, use_lo=True)
    rgb, sigma = emu.run(tok, r['tap_rgb'].numpy(), r['x_c'].numpy(), r['v_c'].numpy())
    ref_rgb, ref_sig = r['sample_rgb'].numpy(), r['sample_sigma'].numpy()
    assert np.abs(sigma - ref_sig).max() / np.abs(ref_sig).max() < 2e-5          # hi + lo fp16 weights: 22 bits
    assert np.abs(rgb - ref_rgb).max() < 2e-5
    # hi-only streams == plain fp16 / bf16 weights: larger but bounded error
    rgb0, sigma0 = Emu(stream, wbias, nkbs, use_lo=False).run(tok, r['tap_rgb'].numpy(), r['x_c'].numpy(), r['v_c'].numpy())
    assert 2e-5 < np.abs(sigma0 - ref_sig).max() / np.abs(ref_sig).max() < 5e-3
    s0, wb0, _ = mlp_pack.pack(sd, prec=0)
    rgb1, sigma1 = Emu(s0, wb0, nkbs, prec=0).run(tok, r['tap_rgb'].numpy(), r['x_c'].numpy(), r['v_c'].numpy())
    assert np.abs(sigma1 - ref_sig).max() / np.abs(ref_sig).max() < 3e-2


def test_fast_erf_formula_accuracy():
    """The optional erf of the MLP kernel's GELU (csrc/mlp.hip: SHERF_MLP_FAST_ERF, Abramowitz-Stegun 7.1.26) evaluated in
    float32 exactly as the kernel writes it: |error| stays below 6e-7 over the whole range, odd symmetry, saturation."""
    import math
    f = np.float32
    x = np.concatenate([np.linspace(-6, 6, 20001), [0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0]]).astype(f)
    ax = np.abs(x)
    t = f(1.0) / (f(0.3275911) * ax + f(1.0))
    poly = t * (t * (t * (t * (t * f(1.061405429) + f(-1.453152027)) + f(1.421413741)) + f(-0.284496736)) + f(0.254829592))
    y = np.copysign(f(1.0) - poly * np.exp(-ax * ax, dtype=f), x)
    ref = np.array([math.erf(float(v)) for v in x])
    assert np.abs(y.astype(np.float64) - ref).max() < 6e-7        # 1.5e-7 (formula) + float32 rounding
    assert y[-2] == 1.0 and y[-1] == -1.0
