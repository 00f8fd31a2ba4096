"""GPU tests of the backward kernels (BASELINE config 5) -- under `-m gpu` since round 2, when they first ran on an MI355X (green after
two fixes of the test harness).  Each kernel is checked against the oracle's stage-boundary gradients, the whole chain against the
fingerprints of the UNMODIFIED reference's gradients (tests/golden/grad_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import sherf_oracle as O
from tests import gpu_common as G

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]


@pytest.mark.parametrize('cfg', ['tiny', 'tiny_nv'])
def test_composite_backward_kernel(cfg):
    from sherf_amd.backward import composite_backward
    fx = G.fixture(cfg)
    loss, g = O.gradients_from_fixture(fx, G.seeded_state(), stages=True)
    h = G.hip_render(cfg)
    R = h['rgb'].shape[0]
    rs = np.random.RandomState(11)
    t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
    t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
    d_rgb = (2.0 * (h['rgb'] - t_rgb) / (R * 3)).cuda()
    d_acc = (2.0 * (h['acc'] - t_acc) / R).cuda()
    d = G.to_cuda(fx['input_data'])
    # (1) the whole chain: gradients at OUR forward point (f16x3 decoder, fixed-point BatchNorm) against the oracle's -- the forward's
    #     ~3e-5 differences in sigma are amplified by the exp(-sigma * delta) chain (first hardware run: 2.7e-3 on tiny_nv)
    out = composite_backward(h['rend'], d_rgb, d_acc, d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0]).cpu()
    assert G.rel(out[:, :3], g['stage.sample_rgb']) < 1e-3
    assert G.rel(out[:, 3], g['stage.sample_sigma']) < 5e-3
    # (2) the kernel alone: the oracle's own per-sample (rgb, sigma) and image gradients in the workspace (the compact order is
    #     bit-identical, tests/test_gpu_parity.py) -> only the backward arithmetic differs
    o = G.oracle_render(cfg)
    nv = o['valid'].numel()
    ws = h['last']['ws']
    ws['sample_out'][:nv] = torch.cat([o['sample_rgb'], o['sample_sigma'].view(-1, 1)], 1).cuda()
    d_rgb = (2.0 * (o['rgb'] - t_rgb) / (R * 3)).cuda()
    d_acc = (2.0 * (o['acc'] - t_acc) / R).cuda()
    out = composite_backward(h['rend'], d_rgb, d_acc, d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0]).cpu()
    assert G.rel(out[:, :3], g['stage.sample_rgb']) < 1e-5
    assert G.rel(out[:, 3], g['stage.sample_sigma']) < 2e-4


# ---- every entry point of include/sherf_hip_bwd.h against its torch emulation (tests/bwd_emulator.py) on random data ----
def _pair(rows, cols, ld=None, seed=0, off=3):
    """The same random matrix as a CPU Mat and a GPU Mat (with padding columns when ld > cols)."""
    from sherf_amd.backward_dense import Mat
    ld = ld or cols
    g = torch.Generator().manual_seed(seed)
    buf = torch.randn(rows * ld + 7, generator=g)
    return Mat(buf.clone(), rows, cols, ld, off), Mat(buf.cuda(), rows, cols, ld, off)


def _same(a, b, tol=1e-5):
    assert G.rel(b.tensor().cpu(), a.tensor()) < tol


def test_bwd_gemm_all_transpositions():
    from sherf_amd.backward_dense import HipOps
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    M, N, K = 37, 29, 53
    for tA in (0, 1):
        for tB in (0, 1):
            a_c, a_g = _pair(*((K, M) if tA else (M, K)), ld=61, seed=1)
            b_c, b_g = _pair(*((N, K) if tB else (K, N)), ld=67, seed=2)
            c_c, c_g = _pair(M, N, ld=31, seed=3)
            e.gemm(tA, tB, a_c, b_c, c_c, 0.5); h.gemm(tA, tB, a_g, b_g, c_g, 0.5)
            torch.cuda.synchronize()
            _same(c_c, c_g, 1e-4)
            # gradient-sized operands (1e-7): an fp16 operand split underflows here; the bf16 split must not
            a_c.tensor().mul_(1e-7); a_g.tensor().mul_(1e-7)
            e.gemm(tA, tB, a_c, b_c, c_c, 0.0); h.gemm(tA, tB, a_g, b_g, c_g, 0.0)
            torch.cuda.synchronize()
            _same(c_c, c_g, 1e-4)


def test_round5_gemm_kernels_on_device(monkeypatch):
    """The backward's round-5 GEMM kernels on the MI355X at the decoder / transformer shapes: tall_stream_kernel = the general kernel BIT FOR BIT
    (SHERF_EXPERIMENT bit 6 selects the general one), incl. the padded K = 71 / 199 operands with NaN in the padding; wgrad_shared_kernel /
    wgrad_solo_kernel = round 2's kernel (bit 7) to fp32 rounding (atomics: order-dependent) and both against float64."""
    from sherf_amd.backward_dense import HipOps, Mat
    from sherf_amd import _lib
    h = HipOps()
    g = torch.Generator().manual_seed(51)

    def mat(r, c, ld, scale=1.0):
        return Mat((torch.randn(r * ld, generator=g) * scale).cuda(), r, c, ld)
    last = _lib.lib_bwd().sherf_bwd_gemm_last_path
    rows = 20011
    for K, N, tB in ((128, 128, 1), (128, 128, 0), (128, 71, 0), (128, 199, 0), (64, 187, 0), (32, 144, 1), (48, 32, 1), (32, 32, 1), (144, 32, 0), (71, 128, 1), (199, 128, 1), (187, 64, 1)):
        lda = K + 4 if K % 16 == 0 else (K + 15) // 16 * 16
        A, B, bias = mat(rows, K, lda, 1e-2), (mat(N, K, K + 1) if tB else mat(K, N, N + 2)), mat(1, N, N)
        if K % 16:
            torch.as_strided(A.buf, (rows, lda - K), (lda, 1), K).fill_(float('nan'))
        outs = []
        for general in (1, 0):
            monkeypatch.setenv('SHERF_EXPERIMENT', '64' if general else '0')
            C = Mat(torch.full((rows * (N + 5),), 7.0).cuda(), rows, N, N + 5)
            h.gemm_bias_act(0, tB, A, B, C, bias, 1)
            torch.cuda.synchronize()
            assert last() == (1 if general else 3), (K, N, tB)
            outs.append(C.buf.clone())
        assert torch.equal(outs[0], outs[1]), (K, N, tB)
        a64 = torch.as_strided(A.buf, (rows, K), (lda, 1)).double()
        ref = (a64 @ (B.tensor().double().t() if tB else B.tensor().double()) + bias.tensor().double()).clamp(min=0)
        assert float((torch.as_strided(outs[1], (rows, N), (N + 5, 1)).double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    rows = 70001
    for M, N in ((128, 128), (128, 71), (128, 199), (64, 187), (144, 32), (32, 32), (32, 48), (3, 64), (1, 128)):
        dy, x = mat(rows, M, M + 3, 1e-3), mat(rows, N, N + 1)
        ref = dy.tensor().double().t() @ x.tensor().double()
        outs = []
        for old in (1, 0):
            monkeypatch.setenv('SHERF_EXPERIMENT', '128' if old else '0')
            C = Mat(torch.zeros(M * N).cuda(), M, N)
            h.gemm(1, 0, dy, x, C)
            torch.cuda.synchronize()
            assert last() == (2 if old else 4), (M, N)
            outs.append(C.tensor().double())
        for o in outs:
            assert float((o - ref).abs().max()) <= 3e-6 * float(ref.abs().max()) + 1e-12, (M, N)
    monkeypatch.setenv('SHERF_EXPERIMENT', '0')


def test_fused_data_gradient_store_on_device():
    """sherf_bwd_gemm_dgrad_fused on the MI355X: the one-kernel path (N, K = 128, aligned rows) and the composition of separate kernels against the
    emulator, with and without the rank-one term / mask / column sums; 40 000 rows so that every workgroup's column-sum reduction takes part."""
    from sherf_amd.backward_dense import HipOps, Mat
    from sherf_amd import _lib
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    for rows, K, N, lda, fused in ((40001, 128, 128, 132, True), (3000, 128, 128, 131, False), (5000, 64, 128, 64, False)):
        a_c, a_g = _pair(rows, K, ld=lda, seed=31, off=0)
        b_c, b_g = _pair(K, N, ld=N + 3, seed=32)
        s_c, s_g = _pair(rows, 1, ld=4, seed=33)
        w_c, w_g = _pair(1, N, seed=34)
        m_c, m_g = _pair(rows, N, ld=N + 8, seed=35)
        for use_r1, use_mask, use_sum in ((1, 1, 1), (0, 1, 1), (0, 0, 0)):
            c_c, c_g = _pair(rows, N, ld=N + 5, seed=36)
            o_c, o_g = Mat(torch.full((N,), 0.5), 1, N), Mat(torch.full((N,), 0.5).cuda(), 1, N)
            args = lambda s1, w1, m, o: (s1 if use_r1 else None, w1 if use_r1 else None, m if use_mask else None, o if use_sum else None)
            e.gemm_dgrad_fused(a_c, b_c, c_c, *args(s_c, w_c, m_c, o_c)); h.gemm_dgrad_fused(a_g, b_g, c_g, *args(s_g, w_g, m_g, o_g))
            torch.cuda.synchronize()
            assert (_lib.lib_bwd().sherf_bwd_gemm_last_path() == 5) == fused
            _same(c_c, c_g, 1e-5); _same(o_c, o_g, 1e-4)


def test_residual_joins_in_the_store_on_device():
    """sherf_bwd_gemm_bias_act_add (the transformer's to_out / net.3 with their residual) on the fused path (N = 32, K = 32 / 48, aligned rows) and the
    fallback, and sherf_bwd_ln_bwd_add, against the emulator."""
    from sherf_amd.backward_dense import HipOps, Mat
    from sherf_amd import _lib
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    for rows, K, N, lda, fused in ((30001, 48, 32, 48, True), (7001, 32, 32, 36, True), (900, 32, 32, 33, False), (700, 64, 32, 64, False)):
        a_c, a_g = _pair(rows, K, ld=lda, seed=41, off=0)
        b_c, b_g = _pair(N, K, ld=K + 1, seed=42)
        bi_c, bi_g = _pair(1, N, seed=43)
        r_c, r_g = _pair(rows, N, ld=N, seed=44, off=0)
        for act in (0, 1):
            c_c, c_g = _pair(rows, N, ld=N + 4, seed=45)
            e.gemm_bias_act_add(1, a_c, b_c, c_c, bi_c, act, r_c); h.gemm_bias_act_add(1, a_g, b_g, c_g, bi_g, act, r_g)
            torch.cuda.synchronize()
            assert (_lib.lib_bwd().sherf_bwd_gemm_last_path() == 5) == fused
            _same(c_c, c_g, 1e-5)
    rows = 5003
    dy_c, dy_g = _pair(rows, 32, seed=46, off=0)
    x = torch.randn(rows, 32, generator=torch.Generator().manual_seed(47))
    w_c, w_g = _pair(1, 32, seed=48, off=0)
    xh = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    iv = 1.0 / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    ad_c, ad_g = _pair(rows, 32, seed=49, off=0)
    mk = lambda t, dev: Mat(t.contiguous().view(-1).to(dev), t.shape[0], t.shape[1])
    outs = []
    for ops, dev, dy, w, ad in ((e, 'cpu', dy_c, w_c, ad_c), (h, 'cuda', dy_g, w_g, ad_g)):
        dx, dw, db = Mat(torch.zeros(rows * 32, device=dev), rows, 32), Mat(torch.zeros(32, device=dev), 1, 32), Mat(torch.zeros(32, device=dev), 1, 32)
        ops.ln_bwd(dy, w, mk(xh, dev), mk(iv, dev), dx, dw, db, addend=ad)
        outs.append((dx, dw, db))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        _same(a, b, 1e-4)


def test_bwd_elementwise_kernels():
    from sherf_amd.backward_dense import HipOps, Mat
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    n = 301
    # bias_act / relu_mask / colsum / copy2d on strided views
    y_c, y_g = _pair(n, 40, 45, 4); b_c, b_g = _pair(1, 40, seed=5)
    for act in (0, 1):
        e.bias_act(y_c, b_c, act); h.bias_act(y_g, b_g, act); _same(y_c, y_g)
    d_c, d_g = _pair(n, 40, 41, 6)
    e.relu_mask(d_c, y_c); h.relu_mask(d_g, y_g); _same(d_c, d_g)
    s_c, s_g = Mat(torch.zeros(40), 1, 40), Mat(torch.zeros(40).cuda(), 1, 40)
    e.colsum(d_c, s_c); h.colsum(d_g, s_g); _same(s_c, s_g, 1e-4)
    e.copy2d(y_c.colslice(3, 20), d_c.colslice(0, 17), add=True); h.copy2d(y_g.colslice(3, 20), d_g.colslice(0, 17), add=True); _same(y_c, y_g)
    # the fused forms: ReLU mask + column sums in one pass (C divides 256), bias + ReLU in the product's store (tall path and the others)
    for C in (64, 128):
        m_c, m_g = _pair(1100, C, C + 5, 21); hh_c, hh_g = _pair(1100, C, C + 2, 22)
        o_c, o_g = Mat(torch.zeros(C), 1, C), Mat(torch.zeros(C).cuda(), 1, C)
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
    # positional encoding
    x_c, x_g = _pair(n, 3, 12, 7)
    for NF in (4, 5, 6):
        o_c, o_g = _pair(n, 3 + 6 * NF, 45, 8)
        e.pe(x_c, NF, o_c); h.pe(x_g, NF, o_g); _same(o_c, o_g, 2e-5)
    # LayerNorm forward / backward
    rows = 3 * n
    x_c, x_g = _pair(rows, 32, seed=9); w_c, w_g = _pair(1, 32, seed=10); bb_c, bb_g = _pair(1, 32, seed=11)
    outs_c = [Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(rows), rows, 1)]
    outs_g = [Mat(torch.zeros(rows * 32).cuda(), rows, 32), Mat(torch.zeros(rows * 32).cuda(), rows, 32), Mat(torch.zeros(rows).cuda(), rows, 1)]
    e.ln_fwd(x_c, w_c, bb_c, *outs_c); h.ln_fwd(x_g, w_g, bb_g, *outs_g)
    for a, b in zip(outs_c, outs_g):
        _same(a, b, 1e-5)
    dy_c, dy_g = _pair(rows, 32, seed=12)
    res_c = [Mat(torch.zeros(rows * 32), rows, 32), Mat(torch.zeros(32), 1, 32), Mat(torch.zeros(32), 1, 32)]
    res_g = [Mat(torch.zeros(rows * 32).cuda(), rows, 32), Mat(torch.zeros(32).cuda(), 1, 32), Mat(torch.zeros(32).cuda(), 1, 32)]
    e.ln_bwd(dy_c, w_c, outs_c[1], outs_c[2], *res_c); h.ln_bwd(dy_g, w_g, outs_g[1], outs_g[2], *res_g)
    for a, b in zip(res_c, res_g):
        _same(a, b, 1e-4)
    # attention core forward / backward
    q_c, q_g = _pair(n, 432, seed=13, off=4)           # (the attention kernels take 16-byte aligned operands)
    a_c, a_g = Mat(torch.zeros(n * 27), n, 27), Mat(torch.zeros(n * 27).cuda(), n, 27)
    o_c, o_g = Mat(torch.zeros(n * 144), n, 144), Mat(torch.zeros(n * 144).cuda(), n, 144)
    e.attn_fwd(q_c, a_c, o_c); h.attn_fwd(q_g, a_g, o_g); _same(a_c, a_g); _same(o_c, o_g)
    go_c, go_g = _pair(n, 144, seed=14, off=0)
    dq_c, dq_g = Mat(torch.zeros(n * 432), n, 432), Mat(torch.zeros(n * 432).cuda(), n, 432)
    e.attn_bwd(q_c, a_c, go_c, dq_c); h.attn_bwd(q_g, a_g, go_g, dq_g); _same(dq_c, dq_g, 1e-4)
    # GELU / rgb head
    u_c, u_g = _pair(n, 32, seed=15)
    g_c, g_g = Mat(torch.zeros(n * 32), n, 32), Mat(torch.zeros(n * 32).cuda(), n, 32)
    e.gelu_fwd(u_c, g_c); h.gelu_fwd(u_g, g_g); _same(g_c, g_g)
    d_c, d_g = _pair(n, 32, seed=16)
    e.gelu_bwd(d_c, u_c); h.gelu_bwd(d_g, u_g); _same(d_c, d_g)
    l_c, l_g = _pair(n, 3, seed=17)
    e.rgb_fwd(l_c); h.rgb_fwd(l_g); _same(l_c, l_g)
    d_c, d_g = _pair(n, 3, seed=18)
    e.rgb_bwd(d_c, l_c); h.rgb_bwd(d_g, l_g); _same(d_c, d_g)
    torch.cuda.synchronize()


def test_dense_backward_on_the_gpu():
    """The whole dense-stage backward through the HIP entry points against autograd through the oracle."""
    from sherf_amd.backward_dense import HipOps, Mat, dense_backward
    cfg = 'tiny_nv'
    fx = G.fixture(cfg)
    state = G.seeded_state()
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    h = G.hip_render(cfg)
    ws = h['last']['ws']
    n = int(ws['counters'][0])
    ops = HipOps()
    tok, ext = Mat.zeros(n, 96, 'cuda'), Mat.zeros(n, 12, 'cuda')
    ops.untile(ws['tokens'], ws['extras'], n, tok, ext)
    d_sample = torch.cat([g['stage.sample_rgb'], g['stage.sample_sigma'][:, None]], 1).contiguous().cuda()
    st = {k: v.cuda() for k, v in state.items()}
    d_tin, grads, dWb_pe = dense_backward(ops, st, tok, ext, Mat(d_sample.view(-1), n, 4))
    torch.cuda.synchronize()
    assert G.rel(d_tin.tensor().cpu().view(n, 3, 32), g['stage.tokens_in']) < 2e-3
    for k, v in grads.items():
        assert G.rel(v.cpu(), g[k]) < 5e-3, k


@pytest.mark.parametrize('use_trans', [True, False])
def test_full_backward_against_reference_gradients(use_trans):
    """forward + backward through the HIP pipeline under the stub loss of BASELINE config 5, against the fingerprints of the
    UNMODIFIED reference's gradients (tests/golden/grad_tiny_nv.npz) and the oracle's input gradients.  use_trans = False (round 6): a
    renderer built without its transformer against the reference built the same way (tests/golden/grad_tiny_nv_notrans.npz)."""
    import os
    from sherf_amd.backward import render_backward
    cfg = 'tiny_nv'
    fx = G.fixture(cfg)
    ref = np.load(os.path.join(G.GOLDEN, f'grad_{cfg}.npz' if use_trans else f'grad_{cfg}_notrans.npz'))
    h = G.hip_render(cfg, use_trans=use_trans)              # training-mode forward (batch statistics)
    assert (h['rend'].transformer is not None) == use_trans
    rend, dec = h['rend'], h['dec']                         # (G.hip_modules() with no argument is a DIFFERENT cache entry)
    R = h['rgb'].shape[0]
    rs = np.random.RandomState(11)
    t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
    t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
    d_rgb = (2.0 * (h['rgb'] - t_rgb) / (R * 3)).cuda()
    d_acc = (2.0 * (h['acc'] - t_acc) / R).cuda()
    out = render_backward(rend, dec, d_rgb, d_acc)
    torch.cuda.synchronize()
    grads = dict(out['params'])
    grads.update({'input.planes': out['planes'], 'input.obs_feat': out['obs_feat'], 'input.vertex_feat': out['vertex_feat']})
    names = [k for k in ref.files if k not in ('loss', 'ref_cpu_seconds')]
    assert set(names) == set(grads), set(names) ^ set(grads)
    # encoder entries: rounds 2-5 bounded them by 0.15 ("ill-conditioned on this fixture": a 3e-5 perturbation of the forward moved the
    # reference's own encoder gradients by 5-6 %) -- the perturbation was OUR forward's: three-product sparse convolutions whose lo halves lost
    # their low bits (2e-6 per layer).  With the lo halves carried at 2^11 (round 6, csrc/svox.hip) they meet the bounds of every other gradient.
    enc = lambda k: 'encoder_3d' in k or k == 'input.vertex_feat'
    worst_enc = [0.0, 0.0]
    for k in names:
        ours, r = O.grad_fingerprint(grads[k].float().cpu()), ref[k]
        tn, tv = (ENC_TN, ENC_TV) if enc(k) else (1e-2, 5e-2)
        en, evv = abs(ours[2] - r[2]) / (r[2] + 1e-30), np.linalg.norm(ours[3:] - r[3:]) / (np.linalg.norm(r[3:]) + 1e-30)
        if enc(k):
            worst_enc = [max(worst_enc[0], en), max(worst_enc[1], evv)]
        assert en < tn, (k, ours[2], r[2])
        assert evv < tv, (k, evv)
    print(f'encoder gradients vs the reference golden: worst norm error {worst_enc[0]:.3e}, worst fingerprint error {worst_enc[1]:.3e}')


ENC_TN, ENC_TV = 1e-2, 5e-2          # round 6: the encoder's entries under the bounds of every other gradient (rounds 2-5: 0.15; measured 3.7e-4 / 8.5e-4)


def _full_size_backward(cfg, device, n_expected=None):
    """All gradients of the HIP backward against autograd through the oracle evaluated on `device` ('cuda': stock ATen ops on the same
    GPU, fp32; 'cpu': this function's own plumbing on the host build, tests/test_hipcpu_frame.py); the HIP backward runs TWICE on fresh
    modules to bound the spread its float atomics' order causes."""
    from sherf_amd.backward import render_backward
    fx = G.fixture(cfg)
    info = {}
    O.NN_CHUNK, chunk0 = (32768 if device != 'cpu' else O.NN_CHUNK), O.NN_CHUNK
    try:
        loss_o, g_o = O.gradients_from_fixture(fx, G.state_for(cfg), device=torch.device(device), info=info)
        # round 5: the float64 truth of every gradient on the fp32 run's discrete branches (oracle: gradients_truth64_from_fixture)
        if device != 'cpu':
            torch.cuda.empty_cache()
        _, g_t = O.gradients_truth64_from_fixture(fx, G.state_for(cfg), info, device=None if device == 'cpu' else torch.device(device))
    finally:
        O.NN_CHUNK = chunk0
    g_o = {k: v.detach().float().cpu() for k, v in g_o.items()}
    g_t = {k: v.detach().double().cpu() for k, v in g_t.items()}
    info.pop('decisions', None)
    spi = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in info['sp_input'].items()}
    if device != 'cpu':
        torch.cuda.empty_cache()
    runs = []
    for _ in range(2):
        G.hip_modules.cache_clear()                         # fresh modules: nothing of the previous run's workspace
        h = G.hip_render(cfg, sp_input=spi)                 # training-mode forward, f16x3 (the autograd configuration)
        R = h['rgb'].shape[0]
        rs = np.random.RandomState(11)
        t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1, R, 3)).astype(np.float32))[0]
        t_acc = torch.from_numpy(rs.uniform(0, 1, (1, R, 1)).astype(np.float32))[0, :, 0]
        loss_h = float(((h['rgb'] - t_rgb) ** 2).mean() + ((h['acc'] - t_acc) ** 2).mean())
        out = render_backward(h['rend'], h['dec'], G.dev_tensor(2.0 * (h['rgb'] - t_rgb) / (R * 3)), G.dev_tensor(2.0 * (h['acc'] - t_acc) / R))
        torch.cuda.synchronize()
        grads = {k: G.plain(v.detach().float()) for k, v in out['params'].items()}
        grads.update({'input.planes': G.plain(out['planes']), 'input.obs_feat': G.plain(out['obs_feat']), 'input.vertex_feat': G.plain(out['vertex_feat'])})
        runs.append(grads)
    G.hip_modules.cache_clear()
    assert abs(loss_h - loss_o) < 1e-4 * abs(loss_o), (loss_h, loss_o)
    names = sorted(k for k in g_o if not k.startswith('stage.'))
    assert set(names) == set(runs[0]), set(names) ^ set(runs[0])
    assert n_expected is None or len(names) == n_expected, len(names)
    assert set(g_t) == set(names), set(g_t) ^ set(names)
    worst, spread, ours_t, ref_t = {}, {}, {}, {}
    for k in names:
        ref = g_o[k].reshape(-1).double()
        tru = g_t[k].reshape(-1)
        a, b = runs[0][k].reshape(-1).double(), runs[1][k].reshape(-1).double()
        nrm = float(tru.norm()) + 1e-30
        worst[k] = float((a - ref).norm()) / nrm                # ours vs the fp32 oracle (what rounds 3-4 bounded by 1e-2 / 0.15)
        ours_t[k] = float((a - tru).norm()) / nrm               # ours vs the float64 truth
        ref_t[k] = float((ref - tru).norm()) / nrm              # the fp32 oracle (== the reference's backward) vs the float64 truth
        spread[k] = float((a - b).norm()) / nrm
    top = sorted(ours_t.items(), key=lambda kv: -kv[1])[:8]
    print(f'{cfg}: {len(names)} gradients, loss {loss_h:.6f} vs {loss_o:.6f}; largest run-to-run spread {max(spread.values()):.2e} ({max(spread, key=spread.get)})')
    print('   norm-relative distance from the float64 truth        ours      fp32 reference   (ours vs fp32 reference)')
    for k, v in top:
        print(f'   {k:50s} {v:.3e}   {ref_t[k]:.3e}   ({worst[k]:.3e})')
    nenc = [k for k in names if 'encoder_3d' not in k and k != 'input.vertex_feat']
    print(f'   worst outside the encoder: ours {max(ours_t[k] for k in nenc):.3e}  fp32 reference {max(ref_t[k] for k in nenc):.3e}')
    for k in names:
        # Everything OUTSIDE the sparse encoder: the forward protocol's rule for extreme values (oracle/parity.py) -- our
        # distance from the float64 truth may not exceed TWICE the fp32 reference's own distance from it plus 1e-3 of the gradient's norm
        # (rounds 3-4: a flat 1e-2).  MI355X, 512 x 512 x 64 (profiles/r05_call_f_*): ours 4.5e-4, the fp32 reference 4.1e-4.
        # Round 6: the SAME rule for the gradients behind the sparse encoder (its parameters, the vertex features).  Rounds 3-5 bounded them by
        # 0.15 / 2e-2 of the norm: the input-gradient convolutions multiplied fp16 hi + lo operands whose lo halves lost their low bits to
        # fp16's 2^-24 floor (2e-6 of the specification per layer) and every BatchNorm backward behind them amplifies rounding 10^3-10^4
        # times (it subtracts the mean of a gradient whose common-mode part dominates).  With both lo halves carried at 2^11 (csrc/svox.hip,
        # IN_BN 3: 2-3e-7 of the maximum, fp32's own noise) ours sits where the fp32 reference sits: MI355X, cfg2_ri, worst entry
        # (down0.1.bias) ours 1.017e-2 vs the reference's 1.015e-2 of the norm from the float64 truth (profiles/r06_call_c_*).
        assert ours_t[k] <= 2.0 * ref_t[k] + 1e-3, (k, ours_t[k], ref_t[k])
        assert spread[k] < 1e-3, (k, spread[k])
    return ours_t, ref_t


@pytest.mark.parametrize('cfg', ['cfg2_ri', 'cfg2_dense_ri'])
def test_full_size_backward_against_oracle_autograd(cfg):
    """BASELINE config 5 at its OWN size: 512 x 512 rays x 64 samples, ~7 x 10^5 valid samples -- where the
    binned tap scatter (float atomics: order dependent) and the eight-wave tall GEMMs actually run.  Every gradient (78 parameters + the three
    feature inputs) against autograd through the oracle as stock ATen ops on the same GPU."""
    _full_size_backward(cfg, 'cuda')


def test_training_repack_reports_bad_weights_one_step_late():
    """Round 5: under autograd the weight stream's flag word (non-finite / out-of-fp16-range weights) is no longer read back at the repack that set it
    (a host wait behind the previous step's kernels) but at the NEXT repack: a weight that goes bad raises the same ValueError one step later, and
    good steps never wait."""
    from sherf_amd.voxel import SparseConvTensor
    fx = G.fixture('tiny_nv')
    rend, dec = G.hip_modules.__wrapped__('f16x3', 'seeded')
    rend.enable_autograd = True
    spi = G.oracle_render('tiny_nv')['sp_input']
    d = G.to_cuda(fx['input_data'])
    opts = dict(fx['options']); opts['mlp_precision'] = 'f16x3'

    def step():
        planes = G.to_cuda(fx['planes']).requires_grad_(True)
        sp = SparseConvTensor(G.to_cuda(fx['vertex_feat']), G.dev_tensor(spi['coord']), spi['out_sh'], 1)
        si = dict(coord=G.dev_tensor(spi['coord']), out_sh=spi['out_sh'], batch_size=1, bounds=G.dev_tensor(spi['bounds'])[None])
        rgb, depth, acc = rend(planes, d['obs_img_all'][:, 0], G.to_cuda(fx['obs_feat']), sp, None, si, dec, d['ray_o_all'][:, 0], d['ray_d_all'][:, 0],
                               d['near_all'][:, 0], d['far_all'][:, 0], d, opts)
        (rgb.square().mean() + acc.square().mean()).backward()
    step()
    assert rend.__dict__.get('_pack_flag_pending') is not None            # the first repack's word is in flight, nothing was waited for
    with torch.no_grad():
        dec.pts_linears[3].weight[5, 7] = 1e6                             # beyond the fp16 range: the next repack flags it ...
    step()                                                                # ... and is itself not read yet
    with torch.no_grad():
        dec.pts_linears[3].weight[5, 7] = 0.5
    with pytest.raises(ValueError, match='fp16 range'):
        step()                                                            # ... but the repack after it reads the word of the bad one


def _random_level(dims, n, seed):
    """A random sparse level: sorted unique keys + the (bits, prefix) records the kernels use, as CPU and GPU dicts."""
    D, H, W = dims
    g = torch.Generator().manual_seed(seed)
    keys = torch.unique(torch.randint(0, D * H * W, (n,), generator=g))
    nwords = (D * H * W + 31) // 32
    bits = torch.zeros(nwords, dtype=torch.int64)
    bits.index_put_((keys // 32,), (1 << (keys % 32)), accumulate=True)
    pop = torch.tensor([bin(int(b)).count('1') for b in bits.tolist()], dtype=torch.int64)
    prefix = torch.cumsum(pop, 0) - pop
    wp = torch.stack([bits, prefix], 1)
    wp32 = wp.to(torch.int64).view(-1)
    wp32 = torch.where(wp32 >= 2 ** 31, wp32 - 2 ** 32, wp32).to(torch.int32).view(nwords, 2)
    cap = keys.numel() + 5
    kp = torch.cat([keys, torch.zeros(5, dtype=keys.dtype)])
    cpu = dict(keys=kp, n_rows=torch.tensor(keys.numel()), dims=dims, cap=cap)
    gpu = dict(keys=kp.to(torch.int32).cuda(), wp=wp32.cuda(), n_rows=torch.tensor([keys.numel()], dtype=torch.int32).cuda(), dims=dims, cap=cap)
    return cpu, gpu


@pytest.mark.parametrize('mode', [0, 1])
def test_sparse_conv_gradient_kernels(mode):
    """sherf_bwd_conv_wgrad / sherf_bwd_conv_dgrad against their emulation on a random sparse level pair."""
    from sherf_amd.backward_dense import HipOps, Mat
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    fine_c, fine_g = _random_level((12, 16, 20), 900, 1)
    if mode == 0:
        out_c, out_g = fine_c, fine_g
    else:
        out_c, out_g = _random_level((6, 8, 10), 300, 2)
    Cin, Cout = 32, 64
    gen = torch.Generator().manual_seed(3)
    in_raw = torch.randn(fine_c['cap'] * Cin, generator=gen); d_raw = torch.randn(out_c['cap'] * Cout, generator=gen)
    bn = torch.randn(3 * Cin, generator=gen); mult = torch.randint(1, 3, (fine_c['cap'],), generator=gen)
    W = torch.randn(Cout * 27 * Cin, generator=gen)
    for use_bn in (False, True):
        dW_c, dW_g = Mat(torch.zeros(Cout * 27 * Cin), Cout, 27 * Cin), Mat(torch.zeros(Cout * 27 * Cin).cuda(), Cout, 27 * Cin)
        e.conv_wgrad(out_c, fine_c, Mat(in_raw.clone(), fine_c['cap'], Cin), Cin, Mat(bn.clone(), 1, 3 * Cin) if use_bn else None,
                     mult if use_bn else None, Mat(d_raw.clone(), out_c['cap'], Cout), Cout, mode, dW_c)
        h.conv_wgrad(out_g, fine_g, Mat(in_raw.cuda(), fine_c['cap'], Cin), Cin, Mat(bn.cuda(), 1, 3 * Cin) if use_bn else None,
                     mult.to(torch.int32).cuda() if use_bn else None, Mat(d_raw.cuda(), out_c['cap'], Cout), Cout, mode, dW_g)
        torch.cuda.synchronize()
        assert G.rel(dW_g.tensor().cpu(), dW_c.tensor()) < 1e-4
    di_c, di_g = Mat(torch.zeros(fine_c['cap'] * Cin), fine_c['cap'], Cin), Mat(torch.zeros(fine_c['cap'] * Cin).cuda(), fine_c['cap'], Cin)
    e.conv_dgrad(fine_c, out_c, Mat(d_raw.clone(), out_c['cap'], Cout), Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_c)
    h.conv_dgrad_valu(fine_g, out_g, Mat(d_raw.cuda(), out_c['cap'], Cout), Cout, Mat(W.cuda(), Cout, 27 * Cin), Cin, mode, di_g)
    torch.cuda.synchronize()
    assert G.rel(di_g.tensor().cpu(), di_c.tensor()) < 1e-4


@pytest.mark.parametrize('mode,Cin,Cout', [(0, 32, 32), (1, 32, 32), (1, 32, 64), (0, 64, 64), (1, 64, 96), (0, 96, 96)])
def test_mfma_input_gradient_convolution(mode, Cin, Cout):
    """HipOps.conv_dgrad as the training step runs it (sherf_svox_conv3_dgrad: the forward's MFMA sparse convolution with mirrored taps,
    exchanged channel roles and the rows scaled into the fp16 split's range by the layer's max |d_raw|) against the specification, for
    every layer shape of the encoder, on gradient-sized values (1e-7) with elements far below the maximum."""
    from sherf_amd.backward_dense import HipOps, Mat
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    fine_c, fine_g = _random_level((12, 16, 20), 900, 1)
    out_c, out_g = (fine_c, fine_g) if mode == 0 else _random_level((6, 8, 10), 300, 2)
    gen = torch.Generator().manual_seed(3 + Cin + Cout)
    d_raw = torch.randn(out_c['cap'] * Cout, generator=gen) * 1e-7
    d_raw[::7] *= 1e-4
    W = torch.randn(Cout * 27 * Cin, generator=gen) * 0.1
    di_c = Mat(torch.zeros(fine_c['cap'] * Cin), fine_c['cap'], Cin)
    di_g = Mat(torch.full((fine_c['cap'] * Cin,), float('nan')).cuda(), fine_c['cap'], Cin)
    e.conv_dgrad(fine_c, out_c, Mat(d_raw.clone(), out_c['cap'], Cout), Cout, Mat(W.clone(), Cout, 27 * Cin), Cin, mode, di_c)
    dm = Mat(d_raw.cuda(), out_c['cap'], Cout)
    n_o, n_i = int(out_c['n_rows']), int(fine_c['n_rows'])
    dm.amax = Mat(d_raw.view(-1, Cout)[:n_o].abs().max().reshape(1).cuda(), 1, 1)
    h.conv_dgrad(fine_g, out_g, dm, Cout, Mat(W.cuda(), Cout, 27 * Cin), Cin, mode, di_g)
    torch.cuda.synchronize()
    a, b = di_c.tensor()[:n_i].double(), di_g.tensor()[:n_i].cpu().double()
    assert torch.isfinite(b).all()
    err = float((a - b).abs().max()) / float(a.abs().max())
    print(f'input-gradient convolution mode {mode} {Cin} -> {Cout}: max error {err:.2e} of the maximum')
    assert err <= 4e-7                       # round 6: both operands' lo halves carried at 2^11 (rounds 3-5: 2e-6)


def test_batchnorm_backward_and_small_encoder_kernels():
    from sherf_amd.backward_dense import HipOps, Mat
    from tests.bwd_emulator import EmuOps
    e, h = EmuOps(), HipOps()
    lev_c, lev_g = _random_level((8, 8, 8), 200, 5)
    n, cap, C = int(lev_c['n_rows']), lev_c['cap'], 64
    gen = torch.Generator().manual_seed(6)
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
    outs_c = [Mat(torch.zeros(cap * C), cap, C), Mat(torch.zeros(C), 1, C), Mat(torch.zeros(C), 1, C)]
    outs_g = [Mat(torch.zeros(cap * C).cuda(), cap, C), Mat(torch.zeros(C).cuda(), 1, C), Mat(torch.zeros(C).cuda(), 1, C)]
    e.bn_relu_bwd(Mat(d_out.clone(), cap, C), Mat(raw.clone(), cap, C), Mat(bn.clone(), 1, 3 * C), Mat(st.clone(), 1, 2 * C), Mat(gamma.clone(), 1, C),
                  mult, torch.tensor(N), torch.tensor(n), *outs_c)
    h.bn_relu_bwd(Mat(d_out.cuda(), cap, C), Mat(raw.cuda(), cap, C), Mat(bn.cuda(), 1, 3 * C), Mat(st.cuda(), 1, 2 * C), Mat(gamma.cuda(), 1, C),
                  mult.to(torch.int32).cuda(), torch.tensor([N], dtype=torch.int32).cuda(), lev_g['n_rows'], *outs_g)
    torch.cuda.synchronize()
    for a, b in zip(outs_c, outs_g):
        assert G.rel(b.tensor().cpu(), a.tensor()) < 1e-4
    assert float(outs_g[0].amax.tensor().view(-1)[0]) == float(outs_g[0].tensor()[:n].abs().max())      # the scale of the MFMA input gradient
    # bn_relu_apply, gather_rows, unfold32
    a_c, a_g = Mat(torch.zeros(cap * C), cap, C), Mat(torch.zeros(cap * C).cuda(), cap, C)
    e.bn_relu_apply(Mat(raw.clone(), cap, C), Mat(bn.clone(), 1, 3 * C), torch.tensor(n), a_c)
    h.bn_relu_apply(Mat(raw.cuda(), cap, C), Mat(bn.cuda(), 1, 3 * C), lev_g['n_rows'], a_g)
    assert G.rel(a_g.tensor().cpu(), a_c.tensor()) < 1e-6
    D, H, W = lev_c['dims']
    k = lev_c['keys'][:n]
    coord = torch.stack([torch.zeros_like(k), k // (H * W), (k // W) % H, k % W], 1)
    coord = torch.cat([coord, coord[:7]])                                  # duplicate rows share a voxel
    f_c, f_g = Mat(torch.zeros(coord.shape[0] * 32), coord.shape[0], 32), Mat(torch.zeros(coord.shape[0] * 32).cuda(), coord.shape[0], 32)
    dg = torch.randn(cap * 32, generator=gen)
    e.gather_rows(coord, coord.shape[0], lev_c, Mat(dg.clone(), cap, 32), 32, f_c)
    h.gather_rows(coord.to(torch.int32).cuda(), coord.shape[0], lev_g, Mat(dg.cuda(), cap, 32), 32, f_g)
    assert G.rel(f_g.tensor().cpu(), f_c.tensor()) < 1e-6
    HW, groups = 300, 3
    d_f = torch.randn(groups * HW * 32, generator=gen); Wm = torch.randn(32 * 32, generator=gen); inp = torch.randn(groups * 32 * HW, generator=gen)
    di_c, dw_c = Mat(torch.zeros(groups * 32 * HW), groups * 32, HW), Mat(torch.zeros(1024), 32, 32)
    di_g, dw_g = Mat(torch.zeros(groups * 32 * HW).cuda(), groups * 32, HW), Mat(torch.zeros(1024).cuda(), 32, 32)
    e.unfold32(Mat(d_f.clone(), groups * HW, 32), Mat(Wm.clone(), 32, 32), Mat(inp.clone(), groups * 32, HW), HW, groups, 32, HW * 32, di_c, dw_c)
    h.unfold32(Mat(d_f.cuda(), groups * HW, 32), Mat(Wm.cuda(), 32, 32), Mat(inp.cuda(), groups * 32, HW), HW, groups, 32, HW * 32, di_g, dw_g)
    torch.cuda.synchronize()
    assert G.rel(di_g.tensor().cpu(), di_c.tensor()) < 1e-4 and G.rel(dw_g.tensor().cpu(), dw_c.tensor()) < 1e-4


@pytest.mark.parametrize('binned', ['runs', True, False])
def test_gather_backward_kernel(binned, monkeypatch):
    """sherf_gather_tokens_bwd_binned -- 'runs': the step's form since round 5 (samples sorted by their finest voxel cell, every level's sums in registers,
    run-length flushes); True: round 3's form (binned by the coarsest cell; SHERF_EXPERIMENT bit 8) -- and sherf_gather_tokens_bwd (direct atomics)
    against the oracle's tap stencils, in the kernel's folded channel-last layouts."""
    monkeypatch.setenv('SHERF_EXPERIMENT', '256' if binned is True else '0')
    import ctypes
    from oracle import backward_explicit as BX
    from sherf_amd import _lib
    from sherf_amd.backward_dense import HipOps, Mat
    cfg = 'tiny_nv'
    fx = G.fixture(cfg)
    loss, g = O.gradients_from_fixture(fx, G.seeded_state(), stages=True)
    r = G.oracle_render(cfg)
    h = G.hip_render(cfg)
    last, ws = h['last'], h['last']['ws']
    n = int(ws['counters'][0])
    b = last['bwd']
    P, (Hf, Wf) = fx['planes'].shape[-1], fx['obs_feat'].shape[-2:]
    ops = HipOps()
    d_tok = Mat(g['stage.tokens_in'].reshape(-1).contiguous().cuda(), n, 96)
    tiles = (n + 31) // 32
    d_tiled = torch.zeros(tiles * 3072, device='cuda')
    ops.tile_tokens(d_tok, n, d_tiled)
    L, taps = last['plan']['L'], last['plan']['taps']
    d_planes_f, d_feat_f, d_bias = Mat.zeros(3 * P * P, 32, 'cuda'), Mat.zeros(Hf * Wf, 64, 'cuda'), Mat.zeros(1, 96, 'cuda')
    d_rows = [Mat.zeros(L[t[0]]['cap'], 96, 'cuda') for t in taps]
    f32 = lambda t: t.detach().float().contiguous()
    bounds, vox_min = f32(b['bounds']).view(6), f32(b['vox_min']).view(3)
    args = (_lib.ptr(ws['counters']), _lib.ptr(ws['geom']), _lib.ptr(d_tiled), P, Hf, Wf, b['H'], b['W'],
            last['levels_struct'], _lib.ptr(bounds), _lib.ptr(vox_min), (ctypes.c_int32 * 3)(*b['vox_sh']), last['cap'], ops._p(d_planes_f),
            ops._p(d_feat_f), ops._p(d_rows[0]), ops._p(d_rows[1]), ops._p(d_rows[2]), ops._p(d_bias))
    if binned:
        words = ctypes.c_int64(0)
        _lib.call('sherf_gather_bwd_scratch_words', last['levels_struct'], last['cap'], ctypes.byref(words))
        scratch = torch.full((words.value,), -7, dtype=torch.int32, device='cuda')
        _lib.call('sherf_gather_tokens_bwd_binned', *args, _lib.ptr(scratch), words.value, _lib.stream())
        torch.cuda.synchronize()
        lv = last['levels_struct'][0 if binned == 'runs' else 2]
        assert int(scratch[4:4 + (lv.D + 4) * (lv.H + 4) * (lv.W + 4)].sum()) == n
    else:
        _lib.call('sherf_gather_tokens_bwd', *args, _lib.stream())
    torch.cuda.synchronize()
    dt = g['stage.tokens_in']
    bnd = torch.from_numpy(fx['input_data']['t_world_bounds']).view(2, 3)
    ref_pf = BX.triplane_bwd((3, 32, P, P), r['x_c'], bnd, dt.permute(1, 0, 2)).permute(0, 2, 3, 1).reshape(3 * P * P, 32)
    assert G.rel(d_planes_f.tensor().cpu(), ref_pf) < 1e-4
    H, W = fx['input_data']['obs_img_all'].shape[-2:]
    gg = 2.0 * r['uv'] / torch.tensor([W, H], dtype=torch.float32) - 1.0
    ref_ff = BX._grid_sample_2d_bwd((64, Hf, Wf), gg[:, 0], gg[:, 1], True, dt[:, :2].reshape(n, 64)).permute(1, 2, 0).reshape(Hf * Wf, 64)
    assert G.rel(d_feat_f.tensor().cpu(), ref_ff) < 1e-4
    for (keys, feats, shape), dr in zip(r['taps'], d_rows):
        ref = BX.trilinear_sparse_bwd(keys, feats.shape[0], shape, r['grid'], dt.reshape(n, 96))
        assert G.rel(dr.tensor().cpu()[:feats.shape[0]], ref) < 1e-4
    assert G.rel(d_bias.tensor().cpu().view(3, 32), dt.sum(0)) < 1e-4


@pytest.mark.parametrize('prec', [0, 1, 2])
def test_device_weight_pack_is_the_host_packer_bit_for_bit(prec):
    """sherf_mlp_pack_stream on the MI355X (the MLP's weight stream re-packed by a kernel after every optimiser update) == mlp_pack.pack on
    the host, including the fp16 `lo` fragments that are denormal in fp16 (2^-11 of a small weight); its flag word reports out-of-range
    and non-finite weights."""
    import ctypes
    import numpy as np
    from sherf_amd import _lib, mlp_pack
    state = G.seeded_state()
    sd = {k: v.numpy() for k, v in state.items() if not k.startswith('renderer.encoder_3d.')}
    stream, wbias, _ = mlp_pack.pack(sd, prec=prec)
    names = mlp_pack.packed_names()
    src, bsrc, n_flat = mlp_pack.stream_index({n: sd[n].shape for n in names}, prec=prec)
    flat = torch.cat([state[n].float().reshape(-1) for n in names]).cuda()
    src_t, bsrc_t = torch.from_numpy(src).cuda(), torch.from_numpy(bsrc).cuda()

    def run(flat):
        out = torch.zeros(2 * src.size, dtype=torch.uint8, device='cuda')
        wb, flag = torch.zeros(bsrc.size, device='cuda'), torch.full((1,), 7, dtype=torch.int32, device='cuda')
        _lib.call('sherf_mlp_pack_stream', _lib.ptr(flat), _lib.ptr(src_t), src.size, prec, _lib.ptr(out), _lib.ptr(bsrc_t), bsrc.size, _lib.ptr(wb),
                  _lib.ptr(flag), _lib.stream())
        torch.cuda.synchronize()
        return out.cpu().numpy(), wb.cpu().numpy(), int(flag)
    out, wb, flag = run(flat)
    assert flag == 0 and np.array_equal(out, stream) and np.array_equal(wb, wbias)
    big = flat.clone(); big[int(src[src >= 0][0]) >> 1] = 1e5
    assert run(big)[2] == (0 if prec == 0 else 2)
    nan = flat.clone(); nan[int(src[src >= 0][5]) >> 1] = float('nan')
    assert run(nan)[2] & 1
