"""The backward kernels' REAL source executed on the CPU (tests/hipcpu: HIP-on-CPU shim, every GPU thread a cooperative fiber)
against the torch emulation that specifies them (tests/bwd_emulator.py), op by op and through the whole dense-stage
orchestration.  This is what verifies csrc/bwd_dense.hip and csrc/bwd_encoder.hip between GPU sessions."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import sherf_oracle as O
from synthdata import fixtures
from sherf_amd import _lib
from sherf_amd.backward_dense import HipOps, Mat, dense_backward
from tests.bwd_emulator import EmuOps
from tests.hipcpu import build_cpu


@pytest.fixture(scope='module')
def cpu_lib(tmp_path_factory):
    path = build_cpu.build('sherf_hipcpu_bwd', ['bwd_dense.hip', 'bwd_gemm.hip', 'bwd_encoder.hip'], str(tmp_path_factory.mktemp('hipcpu')), compiler=build_cpu.CLANG)
    lib = ctypes.CDLL(path)
    for name, (ret, args) in _lib.parse_header(_lib.HEADER_BWD).items():
        fn = getattr(lib, name)
        fn.restype = ret
        fn.argtypes = [a[0] for a in args]
    return lib


class CpuKernelOps(HipOps):
    """HipOps whose entry points are the CPU build of the same kernel sources (CPU tensors, synchronous)."""

    def __init__(self, lib, monkeypatch):
        self.st = None
        self.lib = lib

        def call(name, *a):
            rc = getattr(lib, name)(*a)
            assert rc == 0, (name, lib.sherf_bwd_last_error())
        monkeypatch.setattr(_lib, 'call_bwd', call)
        monkeypatch.setattr(_lib, 'ptr', lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr()))

    @staticmethod
    def _p(m):
        return ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off)


def _pair(rows, cols, ld=None, seed=0, off=3):
    ld = ld or cols
    buf = torch.randn(rows * ld + 7, generator=torch.Generator().manual_seed(seed))
    return Mat(buf.clone(), rows, cols, ld, off), Mat(buf.clone(), rows, cols, ld, off)


def _same(a, b, tol=1e-5):
    d = float((a.tensor().double() - b.tensor().double()).abs().max())
    assert d <= tol * float(a.tensor().abs().max()) + 1e-12, d


def test_streaming_tall_gemm_is_the_general_kernel_bit_for_bit(cpu_lib, monkeypatch):
    """Round 5: every (column tiles, K-blocks) shape of the decoder / transformer backward with beta == 0 and 16-byte aligned rows takes
    tall_stream_kernel (whole half-tiles of A prefetched, no per-element bounds checks): same six products in the same order -> the same bits
    as the general kernel (SHERF_EXPERIMENT bit 6 selects it), ragged last tile, ragged last column tile, bias + ReLU epilogue, strided C."""
    h = CpuKernelOps(cpu_lib, monkeypatch)
    cpu_lib.sherf_bwd_gemm_last_path.restype = ctypes.c_int
    g = torch.Generator().manual_seed(5)

    def mat(r, c, ld):
        return Mat(torch.randn(r * ld, generator=g) * 1e-2, r, c, ld)
    #          K    N   tB  rows
    cases = [(128, 128, 1, 301), (128, 128, 0, 64), (128, 71, 0, 95), (128, 199, 0, 70), (128, 64, 1, 33), (64, 187, 0, 129), (32, 144, 1, 200), (48, 32, 1, 31),
             (32, 32, 1, 97), (32, 48, 0, 66), (144, 32, 0, 130), (128, 3, 1, 40),
             (71, 128, 1, 77), (199, 128, 1, 45), (187, 64, 1, 100)]          # K not a multiple of 16: rows padded to whole blocks, NaN in the padding
    for K, N, tB, rows in cases:
        A = mat(rows, K, K + 4 if K % 16 == 0 else (K + 15) // 16 * 16)
        if K % 16:
            torch.as_strided(A.buf, (rows, A.ld - K), (A.ld, 1), K).fill_(float('nan'))
        B = mat(N, K, K + 1) if tB else mat(K, N, N + 2)
        bias = mat(1, N, N)
        for act, with_bias in ((1, True), (0, False)):
            outs = []
            for general in (1, 0):
                monkeypatch.setenv('SHERF_EXPERIMENT', '64' if general else '0')
                C = Mat(torch.full((rows * (N + 5),), 7.0), rows, N, N + 5)
                h.gemm_bias_act(0, tB, A, B, C, bias if with_bias else None, act)
                assert cpu_lib.sherf_bwd_gemm_last_path() == (1 if general else 3), (K, N, tB, general)
                outs.append(C.buf.clone())
            assert torch.equal(outs[0], outs[1]), (K, N, tB, rows, act)
            ref = A.tensor().double() @ (B.tensor().double().t() if tB else B.tensor().double())
            if with_bias:
                ref = ref + bias.tensor().double()
            if act:
                ref = ref.clamp(min=0)
            got = torch.as_strided(outs[1], (rows, N), (N + 5, 1)).double()
            assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-12
            assert bool((torch.as_strided(outs[1], (rows, 5), (N + 5, 1), N) == 7.0).all())          # nothing written past the N columns
    monkeypatch.setenv('SHERF_EXPERIMENT', '0')


def test_residual_joins_in_the_store(cpu_lib, monkeypatch):
    """sherf_bwd_gemm_bias_act_add: C = act(A . B^T + bias) + addend on the streaming kernel's fused store (N <= 32, K = 32 / 48, aligned rows: last_path 5) and
    as product + add pass (anything else); sherf_bwd_ln_bwd_add = sherf_bwd_ln_bwd + addend.  Against float64 / the plain entry point."""
    h = CpuKernelOps(cpu_lib, monkeypatch)
    cpu_lib.sherf_bwd_gemm_last_path.restype = ctypes.c_int
    g = torch.Generator().manual_seed(31)

    def mat(r, c, ld, scale=1.0):
        return Mat(torch.randn(r * ld, generator=g) * scale, r, c, ld)
    for rows, K, N, lda, fused in ((301, 48, 32, 48, True), (97, 32, 32, 36, True), (64, 32, 20, 32, True), (90, 32, 32, 33, False), (70, 64, 32, 64, False), (50, 48, 40, 48, False)):
        A, B, bias, R = mat(rows, K, lda), mat(N, K, K + 1), mat(1, N, N), mat(rows, N, N + 2)
        for act, with_bias in ((0, True), (1, True), (0, False)):
            C = Mat(torch.full((rows * (N + 3),), 9.0), rows, N, N + 3)
            h.gemm_bias_act_add(1, A, B, C, bias if with_bias else None, act, R)
            assert (cpu_lib.sherf_bwd_gemm_last_path() == 5) == fused, (rows, K, N, lda)
            ref = A.tensor().double() @ B.tensor().double().t() + (bias.tensor().double() if with_bias else 0.0)
            ref = (ref.clamp(min=0) if act else ref) + R.tensor().double()
            assert float((C.tensor().double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max()), (rows, K, N, act)
            assert bool((torch.as_strided(C.buf, (rows, 3), (N + 3, 1), N) == 9.0).all())
    rows = 777
    dy, xh, w, inv, add = mat(rows, 32, 32), mat(rows, 32, 32), mat(1, 32, 32), mat(rows, 1, 1), mat(rows, 32, 32)
    outs = []
    for use_add in (0, 1):
        dx, dw, db = Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(32), 1, 32), Mat(torch.zeros(32), 1, 32)
        h.ln_bwd(dy, w, xh, inv, dx, dw, db, addend=add if use_add else None)
        outs.append((dx.buf.clone(), dw.buf.clone(), db.buf.clone()))
    assert torch.equal(outs[1][0], outs[0][0] + add.buf)
    assert torch.allclose(outs[1][1], outs[0][1], rtol=1e-5, atol=1e-5) and torch.allclose(outs[1][2], outs[0][2], rtol=1e-5, atol=1e-5)      # (atomics: order-dependent rounding)


def test_fused_data_gradient_store(cpu_lib, monkeypatch):
    """sherf_bwd_gemm_dgrad_fused: C = A . B (+ rank-one term) masked by the layer below's activations, column sums accumulated -- the one-kernel
    path (N, K = 128, aligned rows: last_path 5) and the composition of separate kernels (anything else) against float64, every optional part on / off,
    ragged last tile, strided operands, colsum accumulating into what it held."""
    h = CpuKernelOps(cpu_lib, monkeypatch)
    cpu_lib.sherf_bwd_gemm_last_path.restype = ctypes.c_int
    g = torch.Generator().manual_seed(21)

    def mat(r, c, ld, scale=1.0):
        return Mat(torch.randn(r * ld, generator=g) * scale, r, c, ld)
    for (rows, K, N, lda, fused) in ((301, 128, 128, 132, True), (64, 128, 128, 128, True), (95, 120, 100, 128, True), (77, 128, 128, 131, False), (130, 64, 128, 64, False),
                                     (50, 128, 199, 128, False)):
        A, B = mat(rows, K, lda, 1e-2), mat(K, N, N + 3)
        s1, w1 = mat(rows, 1, 4), mat(1, N, N)
        H = mat(rows, N, N + 8)
        for use_r1, use_mask, use_sum in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (0, 0, 0), (0, 1, 0)):
            C = Mat(torch.full((rows * (N + 5),), 3.0), rows, N, N + 5)
            cs = Mat(torch.full((N,), 0.5), 1, N)
            h.gemm_dgrad_fused(A, B, C, s1 if use_r1 else None, w1 if use_r1 else None, H if use_mask else None, cs if use_sum else None)
            assert (cpu_lib.sherf_bwd_gemm_last_path() == 5) == fused, (rows, K, N, lda)
            ref = A.tensor().double() @ B.tensor().double()
            if use_r1:
                ref = ref + s1.tensor().double() * w1.tensor().double()
            if use_mask:
                ref = ref * (H.tensor() > 0).double()
            scale = float(ref.abs().max())
            assert float((C.tensor().double() - ref).abs().max()) <= 3e-6 * scale + 1e-12, (rows, K, N, use_r1, use_mask)
            assert bool((torch.as_strided(C.buf, (rows, 5), (N + 5, 1), N) == 3.0).all())
            if use_sum:
                assert float((cs.tensor().double() - 0.5 - ref.sum(0, keepdim=True)).abs().max()) <= 1e-5 * scale * rows ** 0.5 + 1e-9
            else:
                assert bool((cs.tensor() == 0.5).all())


def test_weight_gradient_gemm_kernels_of_round_5(cpu_lib, monkeypatch):
    """dW = dy^T . x on wgrad_shared_kernel (B split once per workgroup into LDS, row tiles owned by waves) and wgrad_solo_kernel (one row tile: the
    waves on different steps) for every (row tiles, column tiles) pair of the path: against float64, several slabs, ragged last step, ragged last
    row / column tile, strided operands, accumulation into C (beta 1) and overwrite (beta 0); SHERF_EXPERIMENT bit 7 = round 2's kernel, same result to
    fp32 rounding."""
    h = CpuKernelOps(cpu_lib, monkeypatch)
    cpu_lib.sherf_bwd_gemm_last_path.restype = ctypes.c_int
    g = torch.Generator().manual_seed(11)

    def mat(r, c, ld, scale=1.0):
        return Mat(torch.randn(r * ld, generator=g) * scale, r, c, ld)
    #          M    N   rows
    cases = [(128, 128, 1500), (128, 71, 530), (128, 199, 700), (64, 187, 1030), (144, 32, 1111), (32, 32, 2049), (32, 48, 600), (3, 64, 515), (1, 128, 1000)]
    for M, N, rows in cases:
        dy, x = mat(rows, M, M + 3, 1e-3), mat(rows, N, N + 1)
        ref = dy.tensor().double().t() @ x.tensor().double()
        for beta in (0.0, 1.0):
            outs = []
            for old in (0, 1):
                monkeypatch.setenv('SHERF_EXPERIMENT', '128' if old else '0')
                C = Mat(torch.full((M * (N + 2),), 0.25), M, N, N + 2)
                h.gemm(1, 0, dy, x, C, beta)
                assert cpu_lib.sherf_bwd_gemm_last_path() == (2 if old else 4), (M, N, old)
                outs.append(C.buf.clone())
            want = ref + (0.25 if beta else 0.0)
            for o in outs:
                got = torch.as_strided(o, (M, N), (N + 2, 1)).double()
                assert float((got - want).abs().max()) <= 3e-6 * float(ref.abs().max()) + 1e-9, (M, N, rows, beta)
                assert bool((torch.as_strided(o, (M, 2), (N + 2, 1), N) == 0.25).all())          # nothing written past the N columns
    monkeypatch.setenv('SHERF_EXPERIMENT', '0')


def test_dense_entry_points_match_their_specification(cpu_lib, monkeypatch):
    e, h = EmuOps(), CpuKernelOps(cpu_lib, monkeypatch)
    n = 70
    for tA in (0, 1):
        for tB in (0, 1):
            a_c, a_g = _pair(*((19, 13) if tA else (13, 19)), ld=23, seed=1)
            b_c, b_g = _pair(*((11, 19) if tB else (19, 11)), ld=25, seed=2)
            c_c, c_g = _pair(13, 11, ld=17, seed=3)
            e.gemm(tA, tB, a_c, b_c, c_c, 0.5); h.gemm(tA, tB, a_g, b_g, c_g, 0.5); _same(c_c, c_g)
            # gradient-sized operands (1e-7): an fp16 operand split underflows here (seen on the hardware in round 2); the bf16 parts must not
            for m in (a_c, a_g):
                m.tensor().mul_(1e-7)
            e.gemm(tA, tB, a_c, b_c, c_c, 0.0); h.gemm(tA, tB, a_g, b_g, c_g, 0.0); _same(c_c, c_g)
    # every GEMM shape of the decoder / transformer backward stays on the MFMA kernels -- also the skip layer's data gradient
    # dx = dy[n,128] . W[128,199], whose fragment image (8 K-blocks x 7 column tiles) exceeds the LDS and is cut into column slices
    # (it used to fall through to plain_gemm_kernel, 6.9 ms per call on the hardware)
    cpu_lib.sherf_bwd_gemm_last_path.restype = ctypes.c_int
    rows = 300
    shapes = [(128, 71), (128, 128), (128, 199), (64, 187), (3, 64), (1, 128), (144, 32), (32, 48), (32, 32), (32, 96)]     # (out, in) of every Linear
    for out_f, in_f in shapes:
        for tA, tB, (M, N, K) in ((0, 0, (rows, in_f, out_f)),          # dx = dy . W
                                  (0, 1, (rows, out_f, in_f)),          # forward recompute y = x . W^T
                                  (1, 0, (out_f, in_f, rows))):         # dW = dy^T . x
            a_c, a_g = _pair(*((K, M) if tA else (M, K)), seed=7)
            b_c, b_g = _pair(*((N, K) if tB else (K, N)), seed=8)
            c_c, c_g = _pair(M, N, ld=N + 3, seed=9)
            e.gemm(tA, tB, a_c, b_c, c_c, 1.0); h.gemm(tA, tB, a_g, b_g, c_g, 1.0); _same(c_c, c_g)
            assert cpu_lib.sherf_bwd_gemm_last_path() in (1, 2, 4), (out_f, in_f, tA, tB)
    y_c, y_g = _pair(n, 40, 45, 4); b_c, b_g = _pair(1, 40, seed=5)
    for act in (0, 1):
        e.bias_act(y_c, b_c, act); h.bias_act(y_g, b_g, act); _same(y_c, y_g)
    d_c, d_g = _pair(n, 40, 41, 6)
    e.relu_mask(d_c, y_c); h.relu_mask(d_g, y_g); _same(d_c, d_g)
    s_c, s_g = Mat(torch.zeros(40), 1, 40), Mat(torch.zeros(40), 1, 40)
    e.colsum(d_c, s_c); h.colsum(d_g, s_g); _same(s_c, s_g)
    e.copy2d(y_c.colslice(3, 20), d_c.colslice(0, 17), add=True); h.copy2d(y_g.colslice(3, 20), d_g.colslice(0, 17), add=True); _same(y_c, y_g)
    # the fused forms: ReLU mask + column sums in one pass (C divides 256), bias + ReLU in the product's store (tall path and the others)
    for C in (64, 128):
        m_c, m_g = _pair(1100, C, C + 5, 21); hh_c, hh_g = _pair(1100, C, C + 2, 22)
        o_c, o_g = Mat(torch.zeros(C), 1, C), Mat(torch.zeros(C), 1, C)
        e.relu_mask_colsum(m_c, hh_c, o_c); h.relu_mask_colsum(m_g, hh_g, o_g); _same(m_c, m_g); _same(o_c, o_g, 1e-4)
    for tA, tB, (M, N, K) in ((0, 1, (700, 128, 71)), (0, 1, (700, 3, 64)), (1, 0, (40, 24, 300)), (1, 1, (9, 7, 5))):
        for act, with_bias in ((1, True), (0, True), (1, False)):
            a_c, a_g = _pair(*((K, M) if tA else (M, K)), seed=23)
            bb_c, bb_g = _pair(*((N, K) if tB else (K, N)), seed=24)
            c_c, c_g = _pair(M, N, ld=N + 3, seed=25)
            bi_c, bi_g = _pair(1, N, seed=26)
            e.gemm_bias_act(tA, tB, a_c, bb_c, c_c, bi_c if with_bias else None, act, 0.5)
            h.gemm_bias_act(tA, tB, a_g, bb_g, c_g, bi_g if with_bias else None, act, 0.5)
            _same(c_c, c_g, 1e-4)
    x_c, x_g = _pair(n, 3, 12, 7)
    for NF in (4, 5, 6):
        o_c, o_g = _pair(n, 3 + 6 * NF, 45, 8)
        e.pe(x_c, NF, o_c); h.pe(x_g, NF, o_g); _same(o_c, o_g, 2e-5)
    rows = 3 * n
    x_c, x_g = _pair(rows, 32, seed=9); w_c, w_g = _pair(1, 32, seed=10); bb_c, bb_g = _pair(1, 32, seed=11)
    mk = lambda: [Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(rows), rows, 1)]
    oc, og = mk(), mk()
    e.ln_fwd(x_c, w_c, bb_c, *oc); h.ln_fwd(x_g, w_g, bb_g, *og)
    for a, b in zip(oc, og):
        _same(a, b)
    dy_c, dy_g = _pair(rows, 32, seed=12)
    mk2 = lambda: [Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(32), 1, 32), Mat(torch.zeros(32), 1, 32)]
    rc, rg = mk2(), mk2()
    e.ln_bwd(dy_c, w_c, oc[1], oc[2], *rc); h.ln_bwd(dy_g, w_g, og[1], og[2], *rg)
    for a, b in zip(rc, rg):
        _same(a, b, 1e-4)
    q_c, q_g = _pair(n, 432, seed=13, off=4)           # (the attention kernels take 16-byte aligned operands)
    a_c, a_g = Mat(torch.zeros(n * 27), n, 27), Mat(torch.zeros(n * 27), n, 27)
    o_c, o_g = Mat(torch.zeros(n * 144), n, 144), Mat(torch.zeros(n * 144), n, 144)
    e.attn_fwd(q_c, a_c, o_c); h.attn_fwd(q_g, a_g, o_g); _same(a_c, a_g); _same(o_c, o_g)
    go_c, go_g = _pair(n, 144, seed=14, off=0)
    dq_c, dq_g = Mat(torch.zeros(n * 432), n, 432), Mat(torch.zeros(n * 432), n, 432)
    e.attn_bwd(q_c, a_c, go_c, dq_c); h.attn_bwd(q_g, a_g, go_g, dq_g); _same(dq_c, dq_g, 1e-4)
    u_c, u_g = _pair(n, 32, seed=15)
    g_c, g_g = Mat(torch.zeros(n * 32), n, 32), Mat(torch.zeros(n * 32), n, 32)
    e.gelu_fwd(u_c, g_c); h.gelu_fwd(u_g, g_g); _same(g_c, g_g)
    d_c, d_g = _pair(n, 32, seed=16)
    e.gelu_bwd(d_c, u_c); h.gelu_bwd(d_g, u_g); _same(d_c, d_g)
    l_c, l_g = _pair(n, 3, seed=17)
    e.rgb_fwd(l_c); h.rgb_fwd(l_g); _same(l_c, l_g)
    d_c, d_g = _pair(n, 3, seed=18)
    e.rgb_bwd(d_c, l_c); h.rgb_bwd(d_g, l_g); _same(d_c, d_g)
    # tile / untile round trip and layout
    tok = torch.randn(n, 96, generator=torch.Generator().manual_seed(19))
    tiles = (n + 31) // 32
    t_c, t_g = torch.zeros(tiles * 3072), torch.zeros(tiles * 3072)
    e.tile_tokens(Mat(tok.reshape(-1).clone(), n, 96), n, t_c); h.tile_tokens(Mat(tok.reshape(-1).clone(), n, 96), n, t_g)
    assert torch.equal(t_c, t_g)
    back, ext = Mat(torch.zeros(n * 96), n, 96), Mat(torch.zeros(n * 12), n, 12)
    h.untile(t_g, torch.arange(tiles * 384, dtype=torch.float32), n, back, ext)
    assert torch.equal(back.tensor(), tok)
    assert torch.equal(ext.tensor(), torch.arange(tiles * 384, dtype=torch.float32).view(tiles, 12, 32).permute(0, 2, 1).reshape(tiles * 32, 12)[:n])
    # unfold32 / bn_relu_apply
    HW, groups = 70, 3
    gen = torch.Generator().manual_seed(20)
    d_f = torch.randn(groups * HW * 32, generator=gen); Wm = torch.randn(1024, generator=gen); inp = torch.randn(groups * 32 * HW, generator=gen)
    mk3 = lambda: (Mat(torch.zeros(groups * 32 * HW), groups * 32, HW), Mat(torch.zeros(1024), 32, 32))
    (di_c, dw_c), (di_g, dw_g) = mk3(), mk3()
    e.unfold32(Mat(d_f.clone(), groups * HW, 32), Mat(Wm.clone(), 32, 32), Mat(inp.clone(), groups * 32, HW), HW, groups, 32, HW * 32, di_c, dw_c)
    h.unfold32(Mat(d_f.clone(), groups * HW, 32), Mat(Wm.clone(), 32, 32), Mat(inp.clone(), groups * 32, HW), HW, groups, 32, HW * 32, di_g, dw_g)
    _same(di_c, di_g, 1e-5); _same(dw_c, dw_g, 1e-4)


def test_dense_backward_through_the_real_kernels(cpu_lib, monkeypatch, golden_dir):
    """sherf_amd/backward_dense.py: dense_backward with every entry point served by the CPU build of csrc/bwd_dense.hip, on a
    48-sample slice of `tiny_nv`, against autograd through the oracle."""
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    state = {k: torch.from_numpy(fixtures.seeded_param(k, s)) for k, s in shapes.items() if fixtures.seeded_param(k, s) is not None}
    fx = fixtures.renderer_inputs('tiny_nv')
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
    n = 48
    pe_x, pe_v = O.positional_encoding(r['x_c'][:n], 6), O.positional_encoding(r['v_c'][:n], 4)
    tin = r['tokens_in'][:n].clone().requires_grad_(True)
    st = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in state.items()}
    z = O.transformer(st, tin)
    rgb, sig = O.nerf_decoder(st, pe_x, z, pe_v)
    gen = torch.Generator().manual_seed(1)
    d_rgb, d_sig = torch.randn(n, 3, generator=gen), torch.randn(n, generator=gen)
    ((rgb * d_rgb).sum() + (sig * d_sig).sum()).backward()
    Wb = state['renderer.conv1d_reprojection.weight'][:, 32:64, 0]
    tok = r['tokens_in'][:n].clone()
    tok[:, 2] -= O.positional_encoding(r['tap_rgb'][:n], 5)[:, :32] @ Wb.t()
    ext = torch.zeros(n, 12)
    ext[:, 0:3], ext[:, 3:6], ext[:, 6:9] = r['x_c'][:n], r['v_c'][:n], r['tap_rgb'][:n]
    d_sample = torch.cat([d_rgb, d_sig[:, None]], 1).contiguous()
    ops = CpuKernelOps(cpu_lib, monkeypatch)
    d_tin, grads, dWb_pe = dense_backward(ops, state, Mat(tok.reshape(-1).clone(), n, 96), Mat(ext.reshape(-1).clone(), n, 12),
                                          Mat(d_sample.reshape(-1).clone(), n, 4))
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    assert rel(d_tin.tensor().view(n, 3, 32), tin.grad) < 1e-3
    for k, v in grads.items():
        assert st[k].grad is not None and rel(v, st[k].grad) < 2e-3, (k, rel(v, st[k].grad))


@pytest.mark.parametrize('mode,Cin,Cout', [(0, 32, 64), (1, 32, 64), (0, 32, 32), (1, 64, 64), (0, 64, 96), (1, 96, 96)])
def test_sparse_conv_gradient_kernels_on_cpu(cpu_lib, monkeypatch, mode, Cin, Cout):
    """sherf_bwd_conv_wgrad (the compile-time channel pairs of the fast kernel and the generic one: 96 -> 96) and the fp32 form of the input
    gradient against their specification."""
    from tests.bwd_emulator import make_level
    e, h = EmuOps(), CpuKernelOps(cpu_lib, monkeypatch)
    g = torch.Generator().manual_seed(1)
    fine_e, fine_k = make_level(torch.unique(torch.randint(0, 8 * 10 * 12, (260,), generator=g)), (8, 10, 12))
    if mode == 0:
        out_e, out_k = fine_e, fine_k
    else:
        out_e, out_k = make_level(torch.unique(torch.randint(0, 4 * 5 * 6, (70,), generator=g)), (4, 5, 6))
    in_raw = torch.randn(fine_e['cap'] * Cin, generator=g); d_raw = torch.randn(out_e['cap'] * Cout, generator=g)
    bn = torch.randn(3 * Cin, generator=g); mult = torch.randint(1, 3, (fine_e['cap'],), generator=g)
    W = torch.randn(Cout * 27 * Cin, generator=g)
    for use_bn in (False, True):
        dW_c, dW_g = Mat(torch.zeros(Cout * 27 * Cin), Cout, 27 * Cin), Mat(torch.zeros(Cout * 27 * Cin), Cout, 27 * Cin)
        e.conv_wgrad(out_e, fine_e, Mat(in_raw.clone(), fine_e['cap'], Cin), Cin, Mat(bn.clone(), 1, 3 * Cin) if use_bn else None,
                     mult if use_bn else None, Mat(d_raw.clone(), out_e['cap'], Cout), Cout, mode, dW_c)
        h.conv_wgrad(out_k, fine_k, Mat(in_raw.clone(), fine_e['cap'], Cin), Cin, Mat(bn.clone(), 1, 3 * Cin) if use_bn else None,
                     mult.to(torch.int32) if use_bn else None, Mat(d_raw.clone(), out_e['cap'], Cout), Cout, mode, dW_g)
        _same(dW_c, dW_g, 1e-4)
    di_c, di_g = Mat(torch.zeros(fine_e['cap'] * Cin), fine_e['cap'], Cin), Mat(torch.zeros(fine_e['cap'] * Cin), fine_e['cap'], Cin)
    e.conv_dgrad(fine_e, out_e, Mat(d_raw.clone(), out_e['cap'], Cout), Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_c)
    h.conv_dgrad_valu(fine_k, out_k, Mat(d_raw.clone(), out_e['cap'], Cout), Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_g)
    _same(di_c, di_g, 1e-4)


@pytest.fixture(scope='module')
def svox_lib(tmp_path_factory):
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the MFMA kernels')
    path = build_cpu.build('sherf_hipcpu_svox', ['svox.hip'], str(tmp_path_factory.mktemp('hipcpu_svox')),
                           extra_src='char g_sherf_err[256] = ""; int g_sherf_debug = 0;\n', compiler=build_cpu.CLANG)
    lib = ctypes.CDLL(path)
    protos = _lib.parse_header()
    for fn in ('sherf_svox_conv3_dgrad',):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = protos[fn][0], [a[0] for a in protos[fn][1]]
    return lib


@pytest.mark.parametrize('mode,Cin,Cout', [(0, 32, 32), (1, 32, 32), (1, 32, 64), (0, 64, 64), (1, 64, 96), (0, 96, 96)])
def test_mfma_input_gradient_convolution_on_cpu(cpu_lib, svox_lib, monkeypatch, mode, Cin, Cout):
    """HipOps.conv_dgrad as the training step runs it -- the forward's MFMA sparse-convolution kernel with mirrored taps, exchanged channel
    roles, and the rows scaled into the fp16 split's range by the maximum sherf_bwd_bn_relu left behind -- against the specification
    (tests/bwd_emulator.py) for every layer shape of the encoder, on gradient-sized values (1e-7: unscaled, the split underflows)."""
    from tests.bwd_emulator import make_level
    e, h = EmuOps(), CpuKernelOps(cpu_lib, monkeypatch)

    def call(name, *a):
        rc = getattr(svox_lib, name)(*a)
        assert rc == 0, name
    monkeypatch.setattr(_lib, 'call', call)
    g = torch.Generator().manual_seed(3 + Cin + Cout)
    fine_e, fine_k = make_level(torch.unique(torch.randint(0, 8 * 10 * 12, (260,), generator=g)), (8, 10, 12))
    out_e, out_k = (fine_e, fine_k) if mode == 0 else make_level(torch.unique(torch.randint(0, 4 * 5 * 6, (70,), generator=g)), (4, 5, 6))
    d_raw = torch.randn(out_e['cap'] * Cout, generator=g) * 1e-7
    d_raw[::7] *= 1e-4                                             # elements far below the maximum
    W = torch.randn(Cout * 27 * Cin, generator=g) * 0.1
    di_c, di_g = Mat(torch.zeros(fine_e['cap'] * Cin), fine_e['cap'], Cin), Mat(torch.full((fine_e['cap'] * Cin,), float('nan')), fine_e['cap'], Cin)
    e.conv_dgrad(fine_e, out_e, Mat(d_raw.clone(), out_e['cap'], Cout), Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_c)
    dm = Mat(d_raw.clone(), out_e['cap'], Cout)
    n_o = int(out_e['n_rows'])
    dm.amax = Mat(d_raw.view(-1, Cout)[:n_o].abs().max().reshape(1).clone(), 1, 1)             # (as sherf_bwd_bn_relu leaves it: bits of max |d_raw|)
    h.conv_dgrad(fine_k, out_k, dm, Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_g)
    n_i = int(fine_e['n_rows'])
    a, b = di_c.tensor()[:n_i].double(), di_g.tensor()[:n_i].double()
    assert torch.isfinite(b).all()
    err = float((a - b).abs().max()) / float(a.abs().max())
    print(f'input-gradient convolution mode {mode} {Cin} -> {Cout}: max error {err:.2e} of the maximum')
    assert err <= 4e-7                       # round 6: both operands' lo halves carried at 2^11 (rounds 3-5: 2e-6)
    with pytest.raises(RuntimeError):
        h.conv_dgrad(fine_k, out_k, Mat(d_raw.clone(), out_e['cap'], Cout), Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_g)   # no amax


def test_batchnorm_backward_and_row_kernels_on_cpu(cpu_lib, monkeypatch):
    from tests.bwd_emulator import make_level
    e, h = EmuOps(), CpuKernelOps(cpu_lib, monkeypatch)
    gen = torch.Generator().manual_seed(6)
    lev_e, lev_k = make_level(torch.unique(torch.randint(0, 512, (150,), generator=gen)), (8, 8, 8))
    n, cap, C = int(lev_e['n_rows']), lev_e['cap'], 64
    raw = torch.randn(cap * C, generator=gen); d_out = torch.randn(cap * C, generator=gen)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    mult = torch.randint(1, 4, (cap,), generator=gen)
    N = int(mult[:n].sum())
    x = raw.view(cap, C)[:n]
    mean = x.sum(0) / N
    var = (((x - mean) ** 2).sum(0) + (N - n) * mean ** 2) / N
    inv = 1 / torch.sqrt(var + 1e-3)
    scale, shift = gamma * inv, beta - mean * gamma * inv
    bn, st = torch.cat([scale, shift, torch.relu(shift)]), torch.cat([mean, var])
    mk = lambda: [Mat(torch.zeros(cap * C), cap, C), Mat(torch.zeros(C), 1, C), Mat(torch.zeros(C), 1, C)]
    oc, og = mk(), mk()
    for use_mult in (True, False):
        e.bn_relu_bwd(Mat(d_out.clone(), cap, C), Mat(raw.clone(), cap, C), Mat(bn.clone(), 1, 3 * C), Mat(st.clone(), 1, 2 * C), Mat(gamma.clone(), 1, C),
                      mult if use_mult else None, torch.tensor(N), torch.tensor(n), *oc)
        h.bn_relu_bwd(Mat(d_out.clone(), cap, C), Mat(raw.clone(), cap, C), Mat(bn.clone(), 1, 3 * C), Mat(st.clone(), 1, 2 * C), Mat(gamma.clone(), 1, C),
                      mult.to(torch.int32) if use_mult else None, torch.tensor([N], dtype=torch.int32), lev_k['n_rows'], *og)
        for a, b in zip(oc, og):
            _same(a, b, 1e-4)
        assert float(og[0].amax.tensor().view(-1)[0]) == float(og[0].tensor()[:n].abs().max())      # the scale of the MFMA input gradient
    a_c, a_g = Mat(torch.zeros(cap * C), cap, C), Mat(torch.zeros(cap * C), cap, C)
    e.bn_relu_apply(Mat(raw.clone(), cap, C), Mat(bn.clone(), 1, 3 * C), torch.tensor(n), a_c)
    h.bn_relu_apply(Mat(raw.clone(), cap, C), Mat(bn.clone(), 1, 3 * C), lev_k['n_rows'], a_g)
    _same(a_c, a_g, 1e-6)
    D, H, W = lev_e['dims']
    k = lev_e['keys'][:n]
    coord = torch.stack([torch.zeros_like(k), k // (H * W), (k // W) % H, k % W], 1)
    coord = torch.cat([coord, coord[:7]])
    f_c, f_g = Mat(torch.zeros(coord.shape[0] * 32), coord.shape[0], 32), Mat(torch.zeros(coord.shape[0] * 32), coord.shape[0], 32)
    dg = torch.randn(cap * 32, generator=gen)
    e.gather_rows(coord, coord.shape[0], lev_e, Mat(dg.clone(), cap, 32), 32, f_c)
    h.gather_rows(coord.to(torch.int32).contiguous(), coord.shape[0], lev_k, Mat(dg.clone(), cap, 32), 32, f_g)
    _same(f_c, f_g, 1e-6)


# ---- kernels of the FORWARD library that need no gfx950 intrinsic: compositing (fwd + bwd), gather (fwd + bwd), table folds ----
@pytest.fixture(scope='module')
def fwd_lib(tmp_path_factory):
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build (_Float16 tables)')
    path = build_cpu.build('sherf_hipcpu_fwd', ['composite.hip', 'gather.hip', 'fold.hip'], str(tmp_path_factory.mktemp('hipcpu_fwd')),
                           extra_src='char g_sherf_err[256] = ""; int g_sherf_debug = 0;\n', compiler=build_cpu.CLANG)
    lib = ctypes.CDLL(path)
    protos = _lib.parse_header()
    for name in ('sherf_composite_compact', 'sherf_composite_compact_bwd', 'sherf_gather_tokens', 'sherf_gather_tokens_pe', 'sherf_gather_tokens_bwd', 'sherf_fold_tables',
                 'sherf_img_to_hwc4', 'sherf_gather_tokens_bwd_binned', 'sherf_gather_bwd_scratch_words'):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = protos[name][0], [a[0] for a in protos[name][1]]
    return lib


@pytest.fixture(scope='module')
def frame(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    state = {k: torch.from_numpy(fixtures.seeded_param(k, s)) for k, s in shapes.items() if fixtures.seeded_param(k, s) is not None}
    fx = fixtures.renderer_inputs('tiny_nv')
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
    return fx, state, r, g


def _P(t):
    return ctypes.c_void_p(t.data_ptr())


def _f2ord(x):                                              # csrc/common.h: f2ord (order-preserving float -> int)
    i = np.float32(x).view(np.int32)
    return int(i if i >= 0 else i ^ 0x7FFFFFFF)


def test_compositing_kernels_on_cpu(fwd_lib, frame):
    fx, state, r, g = frame
    R, S = r['t'].shape
    valid = r['valid']
    nv = valid.numel()
    ray = valid // S
    cnt = torch.bincount(ray, minlength=R).to(torch.int32)
    base = (torch.cumsum(cnt, 0) - cnt).to(torch.int32)
    cs_idx = valid.to(torch.int32).contiguous()
    sample_out = torch.cat([r['sample_rgb'], r['sample_sigma'][:, None]], 1).contiguous()
    d = fixtures.to_torch(fx['input_data'])
    rd, nr, fr = d['ray_d_all'][0, 0].contiguous(), d['near_all'][0, 0, :, 0].contiguous(), d['far_all'][0, 0, :, 0].contiguous()
    counters = torch.tensor([nv, _f2ord(float(r['t'].min())), _f2ord(float(r['t'].max())), 0], dtype=torch.int32)
    rgb, dep, acc = torch.zeros(R, 3), torch.zeros(R), torch.zeros(R)
    assert fwd_lib.sherf_composite_compact(_P(counters), _P(base), _P(cnt), _P(cs_idx), _P(sample_out), _P(rd), _P(nr), _P(fr), R, S, 0, _P(rgb),
                                           _P(dep), _P(acc), None) == 0
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    assert rel(rgb, r['rgb']) < 1e-5 and rel(acc, r['acc']) < 1e-5 and torch.allclose(dep, r['depth'], rtol=1e-5, atol=1e-6)
    rs = np.random.RandomState(11)
    t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
    t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
    d_rgb = (2.0 * (r['rgb'] - t_rgb) / (R * 3)).contiguous(); d_acc = (2.0 * (r['acc'] - t_acc) / R).contiguous()
    d_out = torch.zeros(nv, 4)
    assert fwd_lib.sherf_composite_compact_bwd(_P(base), _P(cnt), _P(cs_idx), _P(sample_out), _P(rd), _P(nr), _P(fr), R, S, 0, _P(d_rgb), _P(d_acc),
                                               _P(d_out), None) == 0
    assert rel(d_out[:, :3], g['stage.sample_rgb']) < 1e-5 and rel(d_out[:, 3], g['stage.sample_sigma']) < 2e-4


def test_gather_kernels_on_cpu(fwd_lib, frame):
    """sherf_fold_tables + sherf_gather_tokens (forward) against the oracle's tokens, and sherf_gather_tokens_bwd against the
    oracle's tap stencils -- the kernels' real source on folded tables built the way sherf_amd/renderer.py builds them."""
    from oracle import backward_explicit as BX
    from tests.bwd_emulator import make_level
    fx, state, r, g = frame
    n = r['x_c'].shape[0]
    planes = torch.from_numpy(fx['planes'])[0].contiguous(); obs_feat = torch.from_numpy(fx['obs_feat'])[0].contiguous()
    obs_img = torch.from_numpy(fx['input_data']['obs_img_all'])[0, 0].contiguous()
    P, (Hf, Wf), (H, W) = planes.shape[-1], obs_feat.shape[-2:], obs_img.shape[-2:]
    Wr = state['renderer.conv1d_reprojection.weight'][:, :, 0]; br = state['renderer.conv1d_reprojection.bias']
    Wp = state['renderer.conv1d_projection.weight'][:, :, 0]; bp = state['renderer.conv1d_projection.bias']
    Wa, Wb, Wc = Wr[:, 0:32], Wr[:, 32:64], Wr[:, 64:96]
    planes_f, feat_f, img4 = torch.zeros(3, P, P, 32), torch.zeros(Hf, Wf, 64), torch.zeros(H, W, 4)
    assert fwd_lib.sherf_fold_tables(_P(planes), _P(Wa.t().contiguous()), _P(planes_f), P * P, 3, 32, P * P * 32, 0, None) == 0
    assert fwd_lib.sherf_fold_tables(_P(obs_feat), _P(Wb.t().contiguous()), _P(feat_f), Hf * Wf, 2, 64, 32, 0, None) == 0
    assert fwd_lib.sherf_img_to_hwc4(_P(obs_img), _P(img4), H * W, None) == 0
    assert torch.allclose(planes_f, torch.einsum('oi,pihw->phwo', Wa, planes), atol=1e-5)
    # voxel levels: (bits, prefix) records + folded rows  F_l = cat_s Wc Wp[32s:32s+32, cols_l]
    levels = (_lib.VoxLevel * 3)()
    keep, kers = [], []
    for i, ((keys, act, shape), (c0, c1)) in enumerate(zip(r['taps'], ((0, 32), (32, 96), (96, 192)))):
        emu, ker = make_level(keys, shape)
        Fcat = torch.cat([Wc @ Wp[32 * s:32 * s + 32, c0:c1] for s in range(3)], 0)          # [96, C]
        rows = torch.zeros(ker['cap'], 96); rows[:act.shape[0]] = act @ Fcat.t()
        keep += [ker['wp'], rows]; kers.append(ker)
        levels[i].wp, levels[i].rows = ker['wp'].data_ptr(), rows.data_ptr()
        levels[i].D, levels[i].H, levels[i].W = shape
    tok_bias = torch.cat([br + Wc @ bp[32 * s:32 * s + 32] for s in range(3)]).contiguous()
    geom = torch.zeros(n, 8); geom[:, 0:3], geom[:, 3:6], geom[:, 6:8] = r['x_c'], r['v_c'], r['uv']
    tiles = (n + 31) // 32
    tokens, extras = torch.zeros(tiles * 3072), torch.zeros(tiles * 384)
    counters = torch.tensor([n, 0, 0, 0], dtype=torch.int32)
    bounds = torch.from_numpy(fx['input_data']['t_world_bounds']).reshape(6).contiguous()
    vox_min = r['sp_input']['bounds'][0].contiguous()
    vox_sh = (ctypes.c_int32 * 3)(*[int(v) for v in r['sp_input']['out_sh']])
    assert fwd_lib.sherf_gather_tokens(_P(counters), _P(geom), _P(planes_f), P, _P(feat_f), Hf, Wf, _P(img4), H, W, levels, _P(tok_bias), _P(bounds),
                                       _P(vox_min), vox_sh, 0, n, _P(tokens), _P(extras), None) == 0
    tok = tokens.view(tiles, 3, 8, 32, 4).permute(0, 3, 1, 2, 4).reshape(tiles * 32, 3, 32)[:n]
    ref = r['tokens_in'].clone()
    ref[:, 2] -= O.positional_encoding(r['tap_rgb'], 5)[:, :32] @ Wb.t()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    assert rel(tok, ref) < 1e-4
    ex = extras.view(tiles, 12, 32).permute(0, 2, 1).reshape(tiles * 32, 12)[:n]
    assert rel(ex[:, 0:3], r['x_c']) < 1e-6 and rel(ex[:, 6:9], r['tap_rgb']) < 1e-5
    # ---- the fp16-table mode (mode | 16): tables folded to half by the same kernel, rows given as half; the taps, their weights and
    #      the sums are unchanged, so the tokens differ from the fp32-table ones by the rounding of the table entries only (2^-11) ----
    planes_h, feat_h = torch.zeros(3, P, P, 32), torch.zeros(Hf, Wf, 64)                      # fp32-sized buffers, first half used
    assert fwd_lib.sherf_fold_tables(_P(planes), _P(Wa.t().contiguous()), _P(planes_h), P * P, 3, 32, P * P * 32, 1, None) == 0
    assert fwd_lib.sherf_fold_tables(_P(obs_feat), _P(Wb.t().contiguous()), _P(feat_h), Hf * Wf, 2, 64, 32, 1, None) == 0
    got_h = planes_h.view(-1).view(torch.float16)[:planes_f.numel()].view_as(planes_f)
    assert torch.equal(got_h, planes_f.half())                                                  # round to nearest even of the fp32 fold
    levels_h = (_lib.VoxLevel * 3)()
    for i in range(3):
        rows_h = keep[2 * i + 1].half().contiguous()
        keep.append(rows_h)
        levels_h[i].wp, levels_h[i].rows = levels[i].wp, rows_h.data_ptr()
        levels_h[i].D, levels_h[i].H, levels_h[i].W = levels[i].D, levels[i].H, levels[i].W
    tokens_h, extras_h = torch.zeros(tiles * 3072), torch.zeros(tiles * 384)
    assert fwd_lib.sherf_gather_tokens(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels_h, _P(tok_bias), _P(bounds),
                                       _P(vox_min), vox_sh, 16, n, _P(tokens_h), _P(extras_h), None) == 0
    tok_h = tokens_h.view(tiles, 3, 8, 32, 4).permute(0, 3, 1, 2, 4).reshape(tiles * 32, 3, 32)[:n]
    assert 1e-6 < rel(tok_h, tok) < 1e-3 and torch.equal(extras_h, extras)
    # round 6: the voxel rows of the next corner requested ahead (the default; absent corners read row 0 with weight 0) and round 5's loop (SHERF_EXPERIMENT bit 9): the same bits
    # (the default on fp16 tables is round 6's sixteen-channel kernel -- two lanes per sample; SHERF_EXPERIMENT bit 10: the eight-channel kernel with the next
    #  corner's rows requested ahead; bits 10 + 9: round 5's eight-channel loop)
    for word in ('1024', '1536'):
        os.environ['SHERF_EXPERIMENT'] = word
        tokens_pf, extras_pf = torch.full((tiles * 3072,), float('nan')), torch.full((tiles * 384,), float('nan'))
        assert fwd_lib.sherf_gather_tokens(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels_h, _P(tok_bias), _P(bounds),
                                           _P(vox_min), vox_sh, 16, n, _P(tokens_pf), _P(extras_pf), None) == 0
        os.environ['SHERF_EXPERIMENT'] = '0'
        assert torch.equal(tokens_pf, tokens_h) and torch.equal(extras_pf, extras_h), word
    # ---- backward: scatter of d_tokens ----
    dt = g['stage.tokens_in']
    pad = torch.zeros(tiles * 32, 96); pad[:n] = dt.reshape(n, 96)
    d_tiled = pad.view(tiles, 32, 3, 8, 4).permute(0, 2, 3, 1, 4).reshape(-1).contiguous()
    bnd = bounds.view(2, 3)
    words = ctypes.c_int64(0)
    assert fwd_lib.sherf_gather_bwd_scratch_words(levels, n, ctypes.byref(words)) == 0 and words.value > 2 * n
    scratch = torch.full((words.value,), -7, dtype=torch.int32)                 # (not zeroed by the caller)
    dbg = ctypes.c_int.in_dll(fwd_lib, 'g_sherf_debug')
    nb0 = (levels[0].D + 4) * (levels[0].H + 4) * (levels[0].W + 4)          # round 5's run order bins by the finest tapped level's cells
    assert words.value == 4 + 4 * nb0 + 2 * n + (nb0 + 1023) // 1024 + 4
    for binned in ('runs', True, 8192, False):         # round 5's form (sorted by finest cell, every level's sums in registers); round 3's (binned by coarsest
      # cell: SHERF_EXPERIMENT bit 8); the same with the coarsest level through memory; the direct one
      os.environ['SHERF_EXPERIMENT'] = '0' if binned == 'runs' else '256'
      dbg.value = binned if binned not in (True, 'runs') and binned else 0
      d_planes_f, d_feat_f, d_bias = torch.zeros(3 * P * P, 32), torch.zeros(Hf * Wf, 64), torch.zeros(96)
      d_rows = [torch.zeros(k['cap'], 96) for k in kers]
      if binned:
        assert fwd_lib.sherf_gather_tokens_bwd_binned(_P(counters), _P(geom), _P(d_tiled), P, Hf, Wf, H, W, levels, _P(bounds), _P(vox_min), vox_sh, n,
                                                      _P(d_planes_f), _P(d_feat_f), _P(d_rows[0]), _P(d_rows[1]), _P(d_rows[2]), _P(d_bias), _P(scratch),
                                                      words.value, None) == 0
        nb = nb0 if binned == 'runs' else (levels[2].D + 4) * (levels[2].H + 4) * (levels[2].W + 4)
        cnt = scratch[4:4 + nb]
        assert int(cnt.sum()) == n and int(scratch[0]) == int((cnt > 0).sum())                  # every sample binned once; the list of non-empty bins
        order = scratch[4 + 4 * nb + n:4 + 4 * nb + 2 * n]
        assert sorted(order.tolist()) == list(range(n))                                          # the sorted order is a permutation of the samples
        if binned == 'runs':                                                                      # ... in ascending order of the samples' finest cell
            keys = scratch[4 + 4 * nb:4 + 4 * nb + n][order.long()]
            assert bool((keys[1:] >= keys[:-1]).all()) and len(set(keys.tolist())) > 10
        assert fwd_lib.sherf_gather_tokens_bwd_binned(_P(counters), _P(geom), _P(d_tiled), P, Hf, Wf, H, W, levels, _P(bounds), _P(vox_min), vox_sh, n,
                                                      _P(d_planes_f), _P(d_feat_f), _P(d_rows[0]), _P(d_rows[1]), _P(d_rows[2]), _P(d_bias), _P(scratch),
                                                      words.value - 1, None) != 0               # scratch too small
      else:
        assert fwd_lib.sherf_gather_tokens_bwd(_P(counters), _P(geom), _P(d_tiled), P, Hf, Wf, H, W, levels, _P(bounds), _P(vox_min), vox_sh, n,
                                               _P(d_planes_f), _P(d_feat_f), _P(d_rows[0]), _P(d_rows[1]), _P(d_rows[2]), _P(d_bias), None) == 0
      _check_scatter(BX, r, dt, n, P, Hf, Wf, H, W, bnd, d_planes_f, d_feat_f, d_rows, d_bias, rel)
    os.environ['SHERF_EXPERIMENT'] = '0'


def _check_scatter(BX, r, dt, n, P, Hf, Wf, H, W, bnd, d_planes_f, d_feat_f, d_rows, d_bias, rel):
    ref_pf = BX.triplane_bwd((3, 32, P, P), r['x_c'], bnd, dt.permute(1, 0, 2)).permute(0, 2, 3, 1).reshape(3 * P * P, 32)
    assert rel(d_planes_f, ref_pf) < 1e-4
    gg = 2.0 * r['uv'] / torch.tensor([W, H], dtype=torch.float32) - 1.0
    ref_ff = BX._grid_sample_2d_bwd((64, Hf, Wf), gg[:, 0], gg[:, 1], True, dt[:, :2].reshape(n, 64)).permute(1, 2, 0).reshape(Hf * Wf, 64)
    assert rel(d_feat_f, ref_ff) < 1e-4
    for (keys, act, shape), dr in zip(r['taps'], d_rows):
        assert rel(dr[:act.shape[0]], BX.trilinear_sparse_bwd(keys, act.shape[0], shape, r['grid'], dt.reshape(n, 96))) < 1e-4
    assert rel(d_bias.view(3, 32), dt.sum(0)) < 1e-4


@pytest.mark.skipif(not os.environ.get('SHERF_SLOW'), reason='~20 s and covered end to end by tests/test_hipcpu_frame.py: SHERF_SLOW=1 to run the stage in isolation')
def test_encoder_backward_through_the_real_kernels(cpu_lib, monkeypatch, golden_dir):
    """sherf_amd/backward_encoder.py: encoder_backward with every entry point served by the CPU build of csrc/bwd_encoder.hip on the
    full 13-layer encoder of `tiny_nv`, against autograd through the oracle.  (Per-kernel checks + the emulated orchestration run
    in the default suite; this is their conjunction.)"""
    from oracle import backward_explicit as BX
    from sherf_amd.backward_encoder import encoder_backward
    from tests.bwd_emulator import make_level
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    state = {k: torch.from_numpy(fixtures.seeded_param(k, s)) for k, s in shapes.items() if fixtures.seeded_param(k, s) is not None}
    fx = fixtures.renderer_inputs('tiny_nv')
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
        coord = r['sp_input']['coord']
        N = coord.shape[0]
        taps, cache = BX.encoder_forward_cached(state, torch.from_numpy(fx['vertex_feat']), coord, r['sp_input']['out_sh'])
    _, inv0, uk0, sh0, g0 = cache[0]
    convs = [e for e in cache if e[0] == 'conv']
    lev_keys, lev_dims = [uk0], [tuple(sh0)]
    for e in convs:
        if e[6]['down']:
            lev_keys.append(e[6]['keys_out']); lev_dims.append(tuple(e[6]['sh_out']))
    levels = [make_level(k, d, pad=3)[1] for k, d in zip(lev_keys, lev_dims)]
    mult = torch.cat([torch.bincount(inv0, minlength=uk0.numel()), torch.ones(3, dtype=torch.long)]).to(torch.int32)
    layers, lev = [], 0
    for i, e in enumerate(convs):
        _, wname, bname, pairs, g_in, bnc, meta = e
        xh, inv, y, xh0, y0, m_, n_rows = bnc
        lev_out = lev + 1 if meta['down'] else lev
        gamma, beta = state[bname + '.weight'], state[bname + '.bias']
        mean, C = -(xh0 / inv), xh.shape[1]
        scale = gamma * inv
        shift = beta - mean * scale
        cap = levels[lev_out]['cap']
        rawp = torch.zeros(cap, C); rawp[:meta['raw'].shape[0]] = meta['raw']
        layers.append(dict(wname=wname, bname=bname, cin=g_in.shape[1], cout=C, down=meta['down'], tap=i in (4, 8, 12), lev_in=lev, lev_out=lev_out,
                           raw=Mat(rawp.reshape(-1), cap, C), bnparam=Mat(torch.cat([scale, shift, torch.relu(shift)]).clone(), 1, 3 * C),
                           stats=Mat(torch.cat([mean, 1.0 / inv ** 2 - 1e-3]).clone(), 1, 2 * C)))
        lev = lev_out
    g0p = torch.zeros(levels[0]['cap'], 32); g0p[:g0.shape[0]] = g0
    ctx = dict(levels=levels, mult=mult, n_total=torch.tensor([N], dtype=torch.int32), coord=coord.to(torch.int32).contiguous(), N=N,
               g0=Mat(g0p.reshape(-1), levels[0]['cap'], 32), layers=layers)
    d_levels = []
    for i, (keys, feats, shape) in enumerate(taps):
        cap = levels[i + 1]['cap']
        d = torch.zeros(cap, feats.shape[1]); d[:feats.shape[0]] = g[f'stage.level{i}']
        d_levels.append(Mat(d.reshape(-1), cap, feats.shape[1]))
    d_feat, grads = encoder_backward(CpuKernelOps(cpu_lib, monkeypatch), state, ctx, d_levels)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    assert rel(d_feat.tensor(), g['input.vertex_feat']) < 2e-3
    for k, v in grads.items():
        assert rel(v, g[k]) < 5e-3, (k, rel(v, g[k]))


# ---- the MFMA kernels: real source, host build with clang (bf16 / ext_vector_type), wave64 collectives and v_mfma emulated by
# the shim; the LDS-DMA / waitcnt / barrier inline asm replaced by their functional equivalents (tests/hipcpu/build_cpu.py) ----
@pytest.fixture(scope='module')
def mlp_lib(tmp_path_factory):
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the bf16 kernels')
    path = build_cpu.build('sherf_hipcpu_mlp', ['mlp.hip'], str(tmp_path_factory.mktemp('hipcpu_mlp')),
                           extra_src='char g_sherf_err[256] = ""; int g_sherf_debug = 0;\n', compiler=build_cpu.CLANG)
    lib = ctypes.CDLL(path)
    protos = _lib.parse_header()
    for fn in ('sherf_nerf_mlp', 'sherf_nerf_mlp2', 'sherf_nerf_mlp3', 'sherf_nerf_mlp3_pe', 'sherf_nerf_mlp_split', 'sherf_mlp_pack_stream'):
        getattr(lib, fn).restype, getattr(lib, fn).argtypes = protos[fn][0], [a[0] for a in protos[fn][1]]
    return lib


@pytest.mark.parametrize('prec', [0, 1, 2])
def test_device_pack_kernel_source_on_cpu(mlp_lib, frame, prec):
    """sherf_mlp_pack_stream (the weight stream packed by a kernel from the live parameters: what a training step does after every
    optimiser update) == the host packer mlp_pack.pack, bit for bit; its flag word reports non-finite / out-of-fp16-range weights."""
    from sherf_amd import mlp_pack
    fx, state, r, g = frame
    sd = {k: v.numpy() for k, v in state.items() if not k.startswith('renderer.encoder_3d.')}
    stream, wbias, _ = mlp_pack.pack(sd, prec=prec)
    names = mlp_pack.packed_names()
    src, bsrc, n_flat = mlp_pack.stream_index({n: sd[n].shape for n in names}, prec=prec)
    flat = torch.cat([state[n].float().reshape(-1) for n in names]).contiguous()
    src_t, bsrc_t = torch.from_numpy(src), torch.from_numpy(bsrc)

    def run(flat):
        out, wb, flag = torch.zeros(2 * src.size, dtype=torch.uint8), torch.zeros(bsrc.size), torch.full((1,), 7, dtype=torch.int32)
        assert mlp_lib.sherf_mlp_pack_stream(_P(flat), _P(src_t), src.size, prec, _P(out), _P(bsrc_t), bsrc.size, _P(wb), _P(flag), None) == 0
        return out.numpy(), wb.numpy(), int(flag)
    out, wb, flag = run(flat)
    assert flag == 0 and np.array_equal(out, stream) and np.array_equal(wb, wbias)
    big = flat.clone(); big[int(src[src >= 0][0]) >> 1] = 1e5
    assert run(big)[2] == (0 if prec == 0 else 2)
    nan = flat.clone(); nan[int(src[src >= 0][5]) >> 1] = float('nan')
    assert run(nan)[2] & 1
    with pytest.raises(ValueError):
        mlp_pack.raise_for_flags(2, 0.0, 1)
    mlp_pack.raise_for_flags(2, 0.0, 0)
    assert mlp_lib.sherf_mlp_pack_stream(_P(flat), _P(src_t), src.size, 3, _P(torch.zeros(4)), _P(bsrc_t), bsrc.size, _P(torch.zeros(4)), _P(torch.zeros(1, dtype=torch.int32)), None) != 0


@pytest.mark.parametrize('prec,tol_sig,tol_rgb', [(1, 2e-5, 2e-5), (0, 5e-2, 5e-2), (2, 6e-3, 6e-3)])
def test_mlp_kernel_source_on_cpu(mlp_lib, frame, prec, tol_sig, tol_rgb):
    """sherf_nerf_mlp: the fused transformer + decoder MFMA kernel executed from its real source on the CPU, against the oracle's
    per-sample rgb / sigma (adversarial seeded weights): f16x3 (prec 1) to fp32 grade -- rel-to-max AND the true per-sample relative error with the floors of
    oracle/parity.py -- and the single-product modes (prec 0 bf16, prec 2 fp16) to their own classes."""
    from oracle import parity
    from sherf_amd import mlp_pack
    fx, state, r, g = frame
    n = r['x_c'].shape[0]
    stream, wbias, _ = mlp_pack.pack({k: v.numpy() for k, v in state.items() if not k.startswith('renderer.encoder_3d.')}, prec=prec)
    stream_t, wbias_t = torch.from_numpy(stream), torch.from_numpy(wbias)
    Wb = state['renderer.conv1d_reprojection.weight'][:, 32:64, 0]
    tok = r['tokens_in'].clone()
    tok[:, 2] -= O.positional_encoding(r['tap_rgb'], 5)[:, :32] @ Wb.t()
    tiles = (n + 31) // 32
    pad = torch.zeros(tiles * 32, 96); pad[:n] = tok.reshape(n, 96)
    tokens = pad.view(tiles, 32, 3, 8, 4).permute(0, 2, 3, 1, 4).reshape(-1).contiguous()
    ext = torch.zeros(tiles * 32, 12)
    ext[:n, 0:3], ext[:n, 3:6], ext[:n, 6:9] = r['x_c'], r['v_c'], r['tap_rgb']
    extras = ext.view(tiles, 32, 12).permute(0, 2, 1).reshape(-1).contiguous()
    counters = torch.tensor([n, 0, 0, 0], dtype=torch.int32)
    out = torch.zeros(tiles * 32, 4)
    assert mlp_lib.sherf_nerf_mlp(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(out), None) == 0
    sig_h, sig_ref = torch.relu(out[:n, 3]), torch.relu(r['sample_sigma'])
    e_sig = float((sig_h - sig_ref).abs().max() / sig_ref.max())
    e_rgb = float((out[:n, :3] - r['sample_rgb']).abs().max())
    r_sig = float(((sig_h - sig_ref).abs() / sig_ref.clamp(min=parity.FLOOR_SIGMA)).max())
    r_rgb = float(((out[:n, :3] - r['sample_rgb']).abs() / r['sample_rgb'].abs().clamp(min=parity.FLOOR_RGB)).max())
    print(f'prec {prec}: sigma+ rel-to-max {e_sig:.2e} per-sample rel {r_sig:.2e}; rgb abs {e_rgb:.2e} per-sample rel {r_rgb:.2e}')
    assert e_sig < tol_sig and e_rgb < tol_rgb, (e_sig, e_rgb)
    if prec == 1:
        assert r_sig < 1e-3 and r_rgb < 1e-3, (r_sig, r_rgb)
    assert mlp_lib.sherf_nerf_mlp(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), 3, n, _P(out), None) != 0        # unknown precision
    # the two-launch form (tokens kernel with resident weights + decoder kernel): the same bits; the non-finite flag word is left alone
    zfrag = torch.zeros(tiles * (2048 if prec == 1 else 1024), dtype=torch.int32)
    for flag in (0, 1):
        counters[3] = flag
        out2 = torch.full((tiles * 32, 4), float('nan'))
        assert mlp_lib.sherf_nerf_mlp_split(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(zfrag), _P(out2), None) == 0
        assert torch.equal(out2[:n], out[:n]) and int(counters[3]) == flag
    # two tiles per wave (round 5, single-product precisions): the same bits for every sample count modulo the 8-tile workgroup (dead tiles
    # and half-empty waves take part in every barrier), and the f16x3 precision is refused
    if prec == 1:
        assert mlp_lib.sherf_nerf_mlp2(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(out), None) != 0
        assert mlp_lib.sherf_nerf_mlp3(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(out), None) != 0
    else:
        for m in sorted({n, n - 32, n - 40, 33, 1}):
            if m < 1:
                continue
            counters[0], counters[3] = m, 0
            ref = torch.full((tiles * 32, 4), float('nan'))
            assert mlp_lib.sherf_nerf_mlp(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(ref), None) == 0
            out3 = torch.full((tiles * 32, 4), float('nan'))
            assert mlp_lib.sherf_nerf_mlp2(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(out3), None) == 0
            assert torch.equal(out3[:m], ref[:m]) and torch.isnan(out3[m:]).all() and int(counters[3]) == 0, m
            # the decoder with its epilogues inside the MFMA stream (sherf_nerf_mlp3)
            out4 = torch.full((tiles * 32, 4), float('nan'))
            assert mlp_lib.sherf_nerf_mlp3(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), prec, n, _P(out4), None) == 0
            assert torch.equal(out4[:m], ref[:m]) and torch.isnan(out4[m:]).all() and int(counters[3]) == 0, m


def test_encodings_from_the_gather_feed_the_network_the_same_bits(fwd_lib, mlp_lib, frame):
    """Round 6: sherf_gather_tokens_pe writes PE6(x_c) / PE4(v_c) / PE5(rgb) as fp16 MFMA operand fragments and
    sherf_nerf_mlp3_pe reads them instead of evaluating the encodings -- real sources on the host build: (a) tokens / extras equal the plain
    fp16-table gather's bit for bit; (b) the fragments hold the oracle's encodings (renderer.py:875-916) in natural feature order, rounded to fp16,
    zero padded; (c) the network's outputs equal sherf_nerf_mlp3's on the same tokens / extras BIT FOR BIT, for every sample count modulo the tile."""
    from sherf_amd import mlp_pack
    from tests.bwd_emulator import make_level
    fx, state, r, g = frame
    n = r['x_c'].shape[0]
    planes = torch.from_numpy(fx['planes'])[0].contiguous(); obs_feat = torch.from_numpy(fx['obs_feat'])[0].contiguous()
    obs_img = torch.from_numpy(fx['input_data']['obs_img_all'])[0, 0].contiguous()
    P, (Hf, Wf), (H, W) = planes.shape[-1], obs_feat.shape[-2:], obs_img.shape[-2:]
    Wr = state['renderer.conv1d_reprojection.weight'][:, :, 0]; br = state['renderer.conv1d_reprojection.bias']
    Wp = state['renderer.conv1d_projection.weight'][:, :, 0]; bp = state['renderer.conv1d_projection.bias']
    Wa, Wb, Wc = Wr[:, 0:32], Wr[:, 32:64], Wr[:, 64:96]
    planes_h, feat_h, img4 = torch.zeros(3, P, P, 32), torch.zeros(Hf, Wf, 64), torch.zeros(H, W, 4)
    assert fwd_lib.sherf_fold_tables(_P(planes), _P(Wa.t().contiguous()), _P(planes_h), P * P, 3, 32, P * P * 32, 1, None) == 0
    assert fwd_lib.sherf_fold_tables(_P(obs_feat), _P(Wb.t().contiguous()), _P(feat_h), Hf * Wf, 2, 64, 32, 1, None) == 0
    assert fwd_lib.sherf_img_to_hwc4(_P(obs_img), _P(img4), H * W, None) == 0
    levels = (_lib.VoxLevel * 3)()
    keep = []
    for i, ((keys, act, shape), (c0, c1)) in enumerate(zip(r['taps'], ((0, 32), (32, 96), (96, 192)))):
        emu, ker = make_level(keys, shape)
        Fcat = torch.cat([Wc @ Wp[32 * s:32 * s + 32, c0:c1] for s in range(3)], 0)
        rows = torch.zeros(ker['cap'], 96); rows[:act.shape[0]] = act @ Fcat.t()
        rows_h = rows.half().contiguous()
        keep += [ker['wp'], rows_h]
        levels[i].wp, levels[i].rows = ker['wp'].data_ptr(), rows_h.data_ptr()
        levels[i].D, levels[i].H, levels[i].W = shape
    tok_bias = torch.cat([br + Wc @ bp[32 * s:32 * s + 32] for s in range(3)]).contiguous()
    geom = torch.zeros(n, 8); geom[:, 0:3], geom[:, 3:6], geom[:, 6:8] = r['x_c'], r['v_c'], r['uv']
    tiles = (n + 31) // 32
    counters = torch.tensor([n, 0, 0, 0], dtype=torch.int32)
    bounds = torch.from_numpy(fx['input_data']['t_world_bounds']).reshape(6).contiguous()
    vox_min = r['sp_input']['bounds'][0].contiguous()
    vox_sh = (ctypes.c_int32 * 3)(*[int(v) for v in r['sp_input']['out_sh']])
    tokens, extras = torch.zeros(tiles * 3072), torch.zeros(tiles * 384)
    assert fwd_lib.sherf_gather_tokens(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels, _P(tok_bias), _P(bounds),
                                       _P(vox_min), vox_sh, 16, n, _P(tokens), _P(extras), None) == 0
    tokens_p, extras_p = torch.zeros(tiles * 3072), torch.zeros(tiles * 384)
    pefrag = torch.full((tiles * 7 * 64 * 8,), float('nan'), dtype=torch.float16)
    assert fwd_lib.sherf_gather_tokens_pe(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels, _P(tok_bias), _P(bounds),
                                          _P(vox_min), vox_sh, 16, n, _P(tokens_p), _P(extras_p), _P(pefrag), None) == 0
    assert torch.equal(tokens_p, tokens) and torch.equal(extras_p, extras)                                  # (a)
    # fp32 tables / the voxel-only pass / no buffer are refused
    assert fwd_lib.sherf_gather_tokens_pe(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels, _P(tok_bias), _P(bounds),
                                          _P(vox_min), vox_sh, 0, n, _P(tokens_p), _P(extras_p), _P(pefrag), None) != 0
    assert fwd_lib.sherf_gather_tokens_pe(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels, _P(tok_bias), _P(bounds),
                                          _P(vox_min), vox_sh, 18, n, _P(tokens_p), _P(extras_p), _P(pefrag), None) != 0
    assert fwd_lib.sherf_gather_tokens_pe(_P(counters), _P(geom), _P(planes_h), P, _P(feat_h), Hf, Wf, _P(img4), H, W, levels, _P(tok_bias), _P(bounds),
                                          _P(vox_min), vox_sh, 16, n, _P(tokens_p), _P(extras_p), None, None) != 0
    # (b) pefrag[tile][q][h][j][8]: feature 16 kb + 8 h + e of the natural order [v, sin(2^0 v), cos(2^0 v), ...]
    ex = extras.view(tiles, 12, 32).permute(0, 2, 1).reshape(tiles * 32, 12)
    pf = pefrag.view(tiles, 7, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(tiles * 32, 7 * 16).float()          # [sample][q * 16 + 8 h + e]
    assert torch.isfinite(pf).all()
    for lo, hi, cols, L, width in ((0, 48, slice(0, 3), 6, 39), (48, 80, slice(3, 6), 4, 27), (80, 112, slice(6, 9), 5, 32)):
        want = O.positional_encoding(ex[:n, cols], L)[:, :width]
        got = pf[:n, lo:hi]
        assert float((got[:, :width] - want).abs().max()) < 1.2e-3, (lo, float((got[:, :width] - want).abs().max()))   # fp16 rounding of values in [-1, 1] (x_c up to ~1.3)
        assert bool((got[:, width:] == 0).all())
    assert bool((pf[n:] [:, 3:48:3] >= 0).all())                                                           # (padding columns: PE(0): finite)
    # (c) the network on the same tokens / extras
    stream, wbias, _ = mlp_pack.pack({k: v.numpy() for k, v in state.items() if not k.startswith('renderer.encoder_3d.')}, prec=2)
    stream_t, wbias_t = torch.from_numpy(stream), torch.from_numpy(wbias)
    for m in sorted({n, n - 32, n - 40, 33, 1}):
        if m < 1:
            continue
        for notrans in (0, 256):
            counters[0], counters[3] = m, 0
            ref = torch.full((tiles * 32, 4), float('nan'))
            assert mlp_lib.sherf_nerf_mlp3(_P(counters), _P(tokens), _P(extras), _P(stream_t), _P(wbias_t), 2 | notrans, n, _P(ref), None) == 0
            out = torch.full((tiles * 32, 4), float('nan'))
            assert mlp_lib.sherf_nerf_mlp3_pe(_P(counters), _P(tokens), _P(extras), _P(pefrag), _P(stream_t), _P(wbias_t), 2 | notrans, n, _P(out), None) == 0
            assert torch.equal(out[:m], ref[:m]) and torch.isnan(out[m:]).all() and int(counters[3]) == 0, (m, notrans)
    counters[0] = n
    for bad in (0, 1):                                                                                      # only the single-fp16-product stream has this form
        assert mlp_lib.sherf_nerf_mlp3_pe(_P(counters), _P(tokens), _P(extras), _P(pefrag), _P(stream_t), _P(wbias_t), bad, n, _P(out), None) != 0
    assert mlp_lib.sherf_nerf_mlp3_pe(_P(counters), _P(tokens), _P(extras), None, _P(stream_t), _P(wbias_t), 2, n, _P(out), None) != 0
