"""sherf_amd.stylegan2 (the tri-plane producer) against the UNMODIFIED reference's StyleGAN2 generator:
checkpoint contract (every parameter / buffer name and shape of the generator SHERF instantiates) and numerics on a small seeded
generator (tests/golden/backbone_small.npz, oracle/make_golden_backbone.py) -- through the operators' explicit `impl='ref'` path and
through the HIP kernels' source executed on the CPU (tests/hipcpu)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle.make_golden_backbone import FULL, SMALL
from synthdata import fixtures
from sherf_amd import _lib, stylegan2 as S
from tests import gpu_common as G
from tests.hipcpu import build_cpu

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
SHAPES = json.load(open(os.path.join(GOLDEN, 'backbone_shapes.json')))
GOLD = np.load(os.path.join(GOLDEN, 'backbone_small.npz'))


def _table(m):
    return {n: list(t.shape) for n, t in list(m.named_parameters()) + list(m.named_buffers())}


def _seed(m, prefix='backbone.'):
    with torch.no_grad():
        for name, t in list(m.named_parameters()) + list(m.named_buffers()):
            if not name.endswith('resample_filter'):
                t.copy_(torch.from_numpy(np.asarray(fixtures.seeded_param(prefix + name, t.shape), np.float32).reshape(tuple(t.shape))))
    return m


def test_checkpoint_contract_of_the_generator_sherf_instantiates():
    """name -> shape of the 256 x 256 x 96 generator (cbase 32768, cmax 512, map_depth 2) equals the reference's, as
    copy_params_and_buffers(require_all=True) demands (training_loop.py:207-208)."""
    g = S.Generator(**FULL)
    assert _table(g) == SHAPES['full'] and g.num_ws == SHAPES['num_ws_full']
    assert _table(S.Generator(**SMALL)) == SHAPES['small']


def _check(g, z, dev=lambda t: t, tol=2e-4):
    def close(a, key):
        a = G.plain(a).numpy()
        ref = GOLD[key + '.sub']
        assert np.abs(a[:, ::7, ::3, ::3] - ref).max() <= tol * np.abs(ref).max(), (key, np.abs(a[:, ::7, ::3, ::3] - ref).max())
        m = GOLD[key + '.moments']
        got = np.array([a.sum(dtype=np.float64), np.abs(a).sum(dtype=np.float64), np.square(a, dtype=np.float64).sum()])
        assert np.all(np.abs(got[1:] - m[1:]) <= 1e-3 * np.abs(m[1:])), (key, got, m)
    with torch.no_grad():
        g.train()
        ws = g.mapping(dev(z), None)
        assert np.abs(G.plain(ws).numpy() - GOLD['ws']).max() < 1e-5
        assert np.abs(G.plain(g.mapping(dev(z), None, truncation_psi=0.7, truncation_cutoff=5)).numpy() - GOLD['ws_trunc']).max() < 1e-5
        close(g.synthesis(ws, noise_mode='const'), 'img_train_const')            # training: un-fused modulation
        close(g.synthesis(ws, noise_mode='none'), 'img_train_none')
        g.eval()
        close(g.synthesis(ws, noise_mode='const'), 'img_eval_const')             # inference: fused (grouped convolution)
        close(g(dev(z), None, noise_mode='const'), 'img_eval_fwd')


def test_small_generator_matches_reference_ref_ops(monkeypatch):
    monkeypatch.setattr(S, 'OPS_IMPL', 'ref')                                    # explicit choice of the operators' stock-PyTorch path
    g = _seed(S.Generator(**SMALL))
    z = torch.from_numpy(np.random.RandomState(3).standard_normal((2, SMALL['z_dim'])).astype(np.float32))
    _check(g, z)


def check_full_size_generator_fp16(dev, tol=5e-3):
    """The full-size generator with the reference's own fp16 path (num_fp16_res = 4, conv_clamp = 256: train.py:427-428, networks_stylegan2.py:423-431) on the
    device against the unmodified reference's source run in the same dtypes (tests/golden/backbone_full_fp16.npz; oracle/make_golden_backbone.py fp16: fp16
    convolutions on the host behind a device-type stand-in -- the reference forces fp32 off-GPU).  Eight fp16 layers round differently on the two sides (the
    library's fp16 convolutions accumulate in fp32 and round once per layer, as the host's do, but in another order): bound 5e-3 of the planes' range (measured on the MI355X: 6.9e-4)."""
    gold = np.load(os.path.join(GOLDEN, 'backbone_full_fp16.npz'))
    g = dev(_seed(S.Generator(**dict(FULL, num_fp16_res=4, conv_clamp=256)))).eval()
    z = torch.from_numpy(np.random.RandomState(5).standard_normal((1, FULL['z_dim'])).astype(np.float32))
    with torch.no_grad():
        ws = g.mapping(dev(z), None)
        img = g.synthesis(ws, noise_mode='const')
    assert img.dtype == torch.float32 and any(b.use_fp16 for b in g.synthesis.children() if hasattr(b, 'use_fp16'))
    img = G.plain(img).numpy()
    ref = gold['img_eval_const.sub']
    err = np.abs(img[:, ::5, ::9, ::9] - ref).max() / np.abs(ref).max()
    m = gold['img_eval_const.moments']
    got = np.array([img.sum(dtype=np.float64), np.abs(img).sum(dtype=np.float64), np.square(img, dtype=np.float64).sum()])
    assert np.isfinite(img).all() and err <= tol and np.all(np.abs(got[1:] - m[1:]) <= 2e-2 * np.abs(m[1:])), (err, got, m)
    return err


def check_full_size_generator(dev=lambda t: t, tol=2e-4):
    """The generator SHERF instantiates (256 x 256 x 96 planes, channel_base 32768, channel_max 512, 28.7 M parameters) RUN at full size, eval mode,
    against the unmodified reference's run of the same seeded generator (tests/golden/backbone_full.npz, oracle/make_golden_backbone.py)."""
    gold = np.load(os.path.join(GOLDEN, 'backbone_full.npz'))
    g = dev(_seed(S.Generator(**FULL))).eval()
    z = torch.from_numpy(np.random.RandomState(5).standard_normal((1, FULL['z_dim'])).astype(np.float32))
    with torch.no_grad():
        ws = g.mapping(dev(z), None)
        img = G.plain(g.synthesis(ws, noise_mode='const')).numpy()
    assert np.abs(G.plain(ws).numpy() - gold['ws']).max() < 1e-5
    assert img.shape == (1, 96, 256, 256)
    ref = gold['img_eval_const.sub']
    err = np.abs(img[:, ::5, ::9, ::9] - ref).max() / np.abs(ref).max()
    m = gold['img_eval_const.moments']
    got = np.array([img.sum(dtype=np.float64), np.abs(img).sum(dtype=np.float64), np.square(img, dtype=np.float64).sum()])
    assert err <= tol and np.all(np.abs(got[1:] - m[1:]) <= 1e-3 * np.abs(m[1:])), (err, got, m)
    return err


def test_full_size_generator_matches_reference_ref_ops(monkeypatch):
    monkeypatch.setattr(S, 'OPS_IMPL', 'ref')
    check_full_size_generator()


def test_small_generator_matches_reference_through_the_hip_kernels_on_cpu(tmp_path_factory, monkeypatch):
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build')
    path = build_cpu.build('sherf_hipcpu_ops', ['ops_lib.hip', 'ops_bias_act.hip', 'ops_upfirdn2d.hip'], str(tmp_path_factory.mktemp('hipcpu_ops_bb')),
                           compiler=build_cpu.CLANG)
    monkeypatch.setattr(_lib, 'LIB_OPS_PATH', path); monkeypatch.setattr(_lib, '_lib_ops', None)
    monkeypatch.setattr(_lib, 'ptr', lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr()))
    monkeypatch.setattr(_lib, 'stream', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))       # every host tensor is a "device" tensor of the shim
    g = _seed(S.Generator(**SMALL))
    z = torch.from_numpy(np.random.RandomState(3).standard_normal((2, SMALL['z_dim'])).astype(np.float32))
    _check(g, z)


def test_gradients_flow_through_the_operators(monkeypatch):
    """training-mode backward through bias_act / upfirdn2d autograd nodes: finite gradients for every parameter that is used."""
    monkeypatch.setattr(S, 'OPS_IMPL', 'ref')
    g = _seed(S.Generator(**SMALL)).train()
    z = torch.from_numpy(np.random.RandomState(4).standard_normal((1, SMALL['z_dim'])).astype(np.float32))
    g(z, None, noise_mode='const').square().mean().backward()
    missing = [n for n, p in g.named_parameters() if p.grad is None]
    assert missing == [], missing
    assert all(torch.isfinite(p.grad).all() for p in g.parameters())


def _torchvision_resnet18_table():
    """name -> shape of torchvision.models.resnet18().state_dict() (the layout a SHERF checkpoint stores under
    `encoder_2d.backbone.` / `encoder_2d_feature.backbone.`), written out from the architecture's definition."""
    t = {'conv1.weight': [64, 3, 7, 7]}

    def bn(prefix, c):
        t.update({f'{prefix}.weight': [c], f'{prefix}.bias': [c], f'{prefix}.running_mean': [c], f'{prefix}.running_var': [c],
                  f'{prefix}.num_batches_tracked': []})
    bn('bn1', 64)
    inp = 64
    for li, planes in enumerate((64, 128, 256, 512), 1):
        for bi in range(2):
            p = f'layer{li}.{bi}'
            t[f'{p}.conv1.weight'] = [planes, inp if bi == 0 else planes, 3, 3]
            bn(f'{p}.bn1', planes)
            t[f'{p}.conv2.weight'] = [planes, planes, 3, 3]
            bn(f'{p}.bn2', planes)
            if bi == 0 and li > 1:
                t[f'{p}.downsample.0.weight'] = [planes, inp, 1, 1]
                bn(f'{p}.downsample.1', planes)
        inp = planes
    t.update({'fc.weight': [1000, 512], 'fc.bias': [1000]})
    return t


def test_resnet18_encoders_contract_and_paths():
    from sherf_amd.resnet import ResNet18Classifier
    enc = ResNet18Classifier().eval()
    got = {k: list(v.shape) for k, v in enc.state_dict().items()}
    want = {'backbone.' + k: v for k, v in _torchvision_resnet18_table().items()}
    assert got == want and len(got) == 122
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((2, 3, 64, 48)).astype(np.float32))
    with torch.no_grad():
        code, feat = enc(x), enc(x, extract_feature=True)
        b = enc.backbone
        manual = b.layer1(torch.relu(b.bn1(b.conv1(x))))                          # triplane.py:327-334: max-pool skipped
    assert code.shape == (2, 512) and feat.shape == (2, 64, 32, 24)
    assert torch.equal(feat, manual)
    assert sum(p.numel() for p in enc.parameters()) == 11689512                    # ResNet-18


def resnet18_functional(sd, x, extract_feature=False, prefix='backbone.'):
    """ResNet-18 (He et al. 2016, table 1: 7x7/2 stem, four stages of two basic blocks at 64 / 128 / 256 / 512 channels, stride-2 1x1 projection
    shortcuts) written as plain torch.nn.functional calls over a torchvision-layout state dict -- an INDEPENDENT formulation of what
    sherf_amd.resnet.ResNet18Classifier computes (torchvision itself is not installed offline), with SHERF's two read-outs (triplane.py:320-343):
    the pooled 512-d code, and layer1's map with the max-pool skipped."""
    F = torch.nn.functional
    g = lambda n: sd[prefix + n]

    def bn(x, n):                                                  # inference-mode batch norm: (x - mean) / sqrt(var + eps) * gamma + beta
        sh = (1, -1, 1, 1)
        return (x - g(n + '.running_mean').view(sh)) / torch.sqrt(g(n + '.running_var').view(sh) + 1e-5) * g(n + '.weight').view(sh) + g(n + '.bias').view(sh)

    def block(x, n, stride):
        y = F.relu(bn(F.conv2d(x, g(n + '.conv1.weight'), None, stride, 1), n + '.bn1'))
        y = bn(F.conv2d(y, g(n + '.conv2.weight'), None, 1, 1), n + '.bn2')
        if (prefix + n + '.downsample.0.weight') in sd:
            x = bn(F.conv2d(x, g(n + '.downsample.0.weight'), None, stride, 0), n + '.downsample.1')
        return F.relu(y + x)
    x = F.relu(bn(F.conv2d(x, g('conv1.weight'), None, 2, 3), 'bn1'))
    if not extract_feature:
        x = F.max_pool2d(x, 3, 2, 1)
    x = block(block(x, 'layer1.0', 1), 'layer1.1', 1)
    if extract_feature:
        return x
    for li in (2, 3, 4):
        x = block(block(x, f'layer{li}.0', 2), f'layer{li}.1', 1)
    return x.mean((2, 3))


def seeded_resnet(seed=0):
    """A ResNet18Classifier with non-trivial running statistics and affine parameters (a fresh module's are 0 / 1: they would not test the norm)."""
    from sherf_amd.resnet import ResNet18Classifier
    torch.manual_seed(seed)
    enc = ResNet18Classifier().eval()
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, t in enc.state_dict().items():
            if n.endswith('running_mean'):
                t.copy_(0.1 * torch.randn(t.shape, generator=gen))
            elif n.endswith('running_var'):
                t.copy_(0.5 + torch.rand(t.shape, generator=gen))
            elif n.endswith('bn1.weight') or n.endswith('bn2.weight') or n.endswith('downsample.1.weight'):
                t.copy_(0.5 + torch.rand(t.shape, generator=gen))
            elif n.endswith('.bias') and 'fc' not in n:
                t.copy_(0.1 * torch.randn(t.shape, generator=gen))
    return enc


def test_resnet18_encoder_against_an_independent_formulation():
    """the encoder was only ever compared with itself.  Here: against resnet18_functional above (the architecture's definition as
    functional calls), both read-outs, non-trivial BatchNorm state."""
    enc = seeded_resnet()
    sd = {k: v.double() for k, v in enc.state_dict().items()}
    x = torch.from_numpy(np.random.RandomState(0).standard_normal((2, 3, 96, 64)).astype(np.float32))
    with torch.no_grad():
        for ef in (False, True):
            got, want = enc(x, extract_feature=ef), resnet18_functional(sd, x.double(), ef)
            assert got.shape == want.shape and G.rel(got, want) < 1e-4, (ef, G.rel(got, want))


def test_generator_builds_its_producers_by_default():
    """TriPlaneGenerator() without injected modules owns the reference's sub-module names (checkpoint contract of
    triplane.py:53-65, minus the super-resolution module no SHERF script uses)."""
    from sherf_amd.triplane import TriPlaneGenerator
    g = TriPlaneGenerator(512, 0, 512, True, True, True, True, True, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=2),
                          rendering_kwargs={}, smpl=G.smpl(), channel_base=1024, channel_max=32, num_fp16_res=0, conv_clamp=None,
                          fused_modconv_default='inference_only')
    kids = dict(g.named_children())
    assert {'renderer', 'ray_sampler', 'encoder_2d', 'encoder_2d_feature', 'conv1d_projection', 'backbone', 'decoder'} <= set(kids)
    assert type(kids['backbone']).__name__ == 'Generator' and kids['backbone'].img_resolution == 256 and kids['backbone'].img_channels == 96
    names = {n for n, _ in g.named_parameters()}
    assert 'backbone.synthesis.b256.torgb.weight' in names and 'encoder_2d_feature.backbone.layer1.1.conv2.weight' in names
    assert 'backbone.mapping.fc1.weight' in names and 'decoder.pts_linears.7.weight' in names
