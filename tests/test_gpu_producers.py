"""`-m gpu`: SURVEY 8(f) rows 2-4 on the MI355X -- the tri-plane / feature-map producers (StyleGAN2 backbone with the HIP bias_act /
upfirdn2d operators, ResNet-18 encoders), the whole TriPlaneGenerator.forward with its own producers, and the reconstruction loss +
weight update -- against the same references the CPU suite uses: the UNMODIFIED reference generator's outputs
(tests/golden/backbone_small.npz) and the oracle renderer."""
import numpy as np
import pytest
import torch

from tests import gpu_common as G

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]


def test_stylegan2_backbone_matches_reference_generator_on_device():
    """The small seeded generator through the HIP operators on the GPU reproduces the reference's ws / planes (training: un-fused
    modulation; inference: fused grouped convolution; truncation; const / no noise) to the CPU suite's tolerance (2e-4 rel)."""
    from sherf_amd import stylegan2 as S
    from tests import test_backbone as TB
    assert S.OPS_IMPL != 'ref'                                   # the product default: the HIP kernels
    g = TB._seed(S.Generator(**TB.SMALL)).cuda()
    z = torch.from_numpy(np.random.RandomState(3).standard_normal((2, TB.SMALL['z_dim'])).astype(np.float32))
    TB._check(g, z, dev=lambda t: t.cuda())


def test_full_size_stylegan2_backbone_matches_reference_generator_on_device():
    """Round 5: the FULL-size generator of triplane.py:58 (channel_base 32768, channel_max 512, planes [1, 96, 256, 256])
    on the MI355X through the HIP operators against the unmodified reference's run of it (tests/golden/backbone_full.npz)."""
    from sherf_amd import stylegan2 as S
    from tests import test_backbone as TB
    assert S.OPS_IMPL != 'ref'
    err = TB.check_full_size_generator(dev=lambda t: t.cuda())
    print(f'full-size backbone on the device vs the reference generator: {err:.2e} of the planes\' range')


def test_full_size_stylegan2_backbone_fp16_path_on_device():
    """Round 6: the reference's own fp16 path of the tri-plane generator (num_fp16_res = 4, conv_clamp = 256) at full size on the MI355X
    against the unmodified reference's source run in the same dtypes (tests/golden/backbone_full_fp16.npz)."""
    from sherf_amd import stylegan2 as S
    from tests import test_backbone as TB
    assert S.OPS_IMPL != 'ref'
    err = TB.check_full_size_generator_fp16(dev=lambda t: t.cuda())
    print(f'full-size backbone, fp16 path, on the device vs the reference generator: {err:.2e} of the planes\' range')


def test_stylegan2_gradients_through_the_hip_operators_on_device():
    """training-mode backward through the bias_act / upfirdn2d autograd nodes (HIP kernels, both derivative orders are exercised by
    tests/test_gpu_ops.py) equals the backward through the stock-PyTorch `ref` path."""
    from sherf_amd import stylegan2 as S
    from tests import test_backbone as TB
    z = torch.from_numpy(np.random.RandomState(4).standard_normal((1, TB.SMALL['z_dim'])).astype(np.float32)).cuda()
    grads = {}
    for impl in ('cuda', 'ref'):
        old, S.OPS_IMPL = S.OPS_IMPL, impl
        try:
            g = TB._seed(S.Generator(**TB.SMALL)).cuda().train()
            g(z, None, noise_mode='const').square().mean().backward()
            grads[impl] = {n: p.grad.detach().float().cpu() for n, p in g.named_parameters() if p.grad is not None}
        finally:
            S.OPS_IMPL = old
    assert set(grads['cuda']) == set(grads['ref']) and len(grads['ref']) > 20
    for n in grads['ref']:
        assert G.rel(grads['cuda'][n], grads['ref'][n]) < 2e-3, n


def test_resnet18_encoder_on_device_against_an_independent_formulation():
    """The ResNet-18 encoders on the MI355X (MIOpen convolutions through PyTorch-ROCm) against the architecture written out as functional calls in
    float64 on the CPU (tests/test_backbone.py: resnet18_functional; torchvision is absent offline) -- both read-outs of triplane.py:320-343, at the
    512 x 512 observation size, non-trivial BatchNorm state.  (Rounds 2-4 compared the module with itself.)"""
    from tests import test_backbone as TB
    enc = TB.seeded_resnet()
    sd = {k: v.double() for k, v in enc.state_dict().items()}
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((1, 3, 512, 512)).astype(np.float32))
    enc = enc.cuda()
    with torch.no_grad():
        code, feat = enc(x.cuda()).cpu(), enc(x.cuda(), extract_feature=True).cpu()
        assert code.shape == (1, 512) and feat.shape == (1, 64, 256, 256)
        assert G.rel(feat, TB.resnet18_functional(sd, x.double(), True)) < 1e-4
        assert G.rel(code, TB.resnet18_functional(sd, x.double(), False)) < 1e-3


def test_density_noise_statistics_on_device():
    """renderer.py:435-436 (training only): sigma += randn * density_noise between the network and the compositing.  On the device: the per-sample
    sigma of a noisy frame minus the clean frame's is N(0, noise^2) (mean / standard deviation over ~600 valid samples), rgb is untouched, the
    image changes, and noise 0 reproduces the clean frame bit for bit."""
    clean = G.hip_render('tiny')
    nv = int(clean['last']['ws']['counters'][0])
    s0 = clean['last']['ws']['sample_out'][:nv].clone()
    noisy = G.hip_render('tiny', options=dict(density_noise=0.5))
    s1 = noisy['last']['ws']['sample_out'][:nv].clone()
    d = (s1[:, 3] - s0[:, 3]).double().cpu()
    assert nv > 300 and torch.equal(s1[:, :3], s0[:, :3])
    assert abs(float(d.mean())) < 4 * 0.5 / nv ** 0.5 and abs(float(d.std()) - 0.5) < 0.06, (float(d.mean()), float(d.std()))
    assert not torch.equal(noisy['rgb'], clean['rgb'])
    again = G.hip_render('tiny', options=dict(density_noise=0))
    assert torch.equal(again['rgb'], clean['rgb']) and torch.equal(again['acc'], clean['acc'])


def test_whole_generator_with_its_own_producers_on_device():
    """TriPlaneGenerator.forward(input_data, z, c) exactly as test_loop.py:189 calls it -- encoders, mapping, tri-plane synthesis, glue,
    renderer -- on the GPU (the same check the CPU suite runs on the host build)."""
    from tests.test_hipcpu_frame import check_whole_generator
    check_whole_generator()


def test_snapshot_and_training_step_against_reference_golden_on_device(monkeypatch):
    """SURVEY 8(f) ranks 3 + 4 on the MI355X against the UNMODIFIED reference's own snapshot / `accumulate_gradients` outputs
    (tests/golden/trainstep_tiny_nv.npz; oracle/make_golden_trainstep.py): names / shapes / values of the snapshot contract, the frame the
    reference renders from its snapshot, our pickled snapshot resuming to the same bits, the six loss terms and all 240 gradient
    fingerprints of one generator step."""
    from tests.test_hipcpu_frame import check_snapshot_and_training_step_against_reference_golden
    check_snapshot_and_training_step_against_reference_golden(monkeypatch)


def test_reconstruction_loss_and_weight_update_on_device():
    """loss.py:103-176 + training_loop.py:365-383 on the GPU: the same terms and the same SGD step as the CPU run of the identical
    stub generator (tests/test_loss.py)."""
    from sherf_amd import loss as L
    from tests import test_loss as TL
    res = {}
    for dev in ('cpu', 'cuda'):
        torch.manual_seed(0)
        d, _ = TL._batch(24, 28)
        d = {k: v.to(dev) for k, v in d.items()}
        Gs = TL._StubG(24, 28).to(dev)
        loss = L.ReconstructionLoss(torch.device(dev), Gs, lpips_fn=lambda a, b: ((a - b) ** 2).mean().reshape(1) * 3.0,
                                    neural_rendering_resolution_initial=48)
        opt = torch.optim.SGD(Gs.parameters(), lr=0.1)
        g = torch.Generator().manual_seed(5)
        z, c = torch.randn(1, 8, generator=g).to(dev), torch.ones(1, 25, device=dev)
        out = L.training_step(Gs, opt, loss, d, z, c, gain=2, num_gpus=1)
        res[dev] = [float(t) for t in out] + [Gs.acc.detach().float().cpu()]
    for a, b in zip(res['cpu'][:-1], res['cuda'][:-1]):
        assert abs(a - b) < 1e-5 * max(1.0, abs(a))
    assert torch.allclose(res['cpu'][-1], res['cuda'][-1], atol=1e-6)


def test_graphed_producer_is_checked_and_falls_back_on_device():
    """Round 6: a producer's hipGraph is trusted only after its first replay reproduced an eager call; a module with a submodule
    in training mode is never graphed (BatchNorm statistics advance once per call, as in eager mode); a capture that cannot reproduce the eager
    call (here: the forward does its work on a stream the capture does not see) turns itself off with a warning and keeps returning eager results."""
    import warnings
    from sherf_amd.triplane import _GraphedProducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU()).cuda().eval()
    gp = _GraphedProducer(net, net.__call__)
    with torch.no_grad():
        for k in range(3):
            x = torch.randn(1, 3, 16, 16, device='cuda')
            y = gp(x)
            assert gp.graph is not None and not gp.off
            assert torch.allclose(y, net(x), rtol=1e-4, atol=1e-6), k           # fresh inputs reach the replay
        # training mode: eager, statistics advance exactly once per call
        net.train()
        n0 = int(net[1].num_batches_tracked)
        gp(x)
        assert int(net[1].num_batches_tracked) == n0 + 1
        net.eval()
        # a capture that cannot reproduce the eager call (the forward bakes a host-side counter into its launch: every replay would return the
        # capture's value) and a capture that fails (the forward refuses to run under capture): off + eager results + one warning each
        class Counting(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.ones(1))
                self.k = 0

            def forward(self, x):
                self.k += 1
                return x * self.w * float(self.k)

        class Refusing(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.ones(1))

            def forward(self, x):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('not under capture')
                return x * 2 * self.w

        for mod, want in ((Refusing(), lambda k, i: 2.0 * k), (Counting(), None)):
            m = mod.cuda().eval()
            gq = _GraphedProducer(m, m.__call__)
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter('always')
                outs = [gq(torch.full((4,), float(k), device='cuda')) for k in range(1, 4)]
            torch.cuda.synchronize()
            assert gq.off and gq.error and sum('eager calls from now on' in str(r.message) for r in rec) == 1, (type(mod).__name__, gq.error)
            if want is not None:
                for k, o in enumerate(outs, 1):
                    assert torch.equal(o.cpu(), torch.full((4,), want(k, 0))), (k, o)
            else:                                       # every call after the failed check is an eager call: the counter keeps advancing
                r = [float(o[0]) / k for k, o in enumerate(outs, 1)]
                assert r[1] == r[0] + 1 and r[2] == r[1] + 1, r
