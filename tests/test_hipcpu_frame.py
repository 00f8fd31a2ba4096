"""The WHOLE product path on the CPU: sherf_amd's Python host code + the native frame driver (csrc/frame.hip) + every kernel
source of libsherf_hip.so / libsherf_hip_bwd.so, compiled unchanged for the host against the HIP-on-CPU shim (tests/hipcpu) and
driven through the same ctypes binding as on the MI355X.  ImportanceRenderer.forward renders the `tiny` fixtures on CPU
tensors and is compared with the oracle and with the unmodified reference's golden outputs; render_backward is compared with
the reference's gradient fingerprints.  These are the `-m gpu` parity tests' counterparts for the sessions without a GPU: they
check the kernels' arithmetic and indexing, not their timing or anything gfx950-specific (LDS-DMA, wave scheduling)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import sherf_oracle as O
from synthdata import fixtures
from sherf_amd import _lib
from tests import gpu_common as G
from tests.hipcpu import build_cpu

FWD_SOURCES = ['smpl.hip', 'sample.hip', 'gather.hip', 'mlp.hip', 'composite.hip', 'svox.hip', 'rays.hip', 'fold.hip', 'frame.hip']


class _FakeCuda(torch.Tensor):
    is_cuda = True


@pytest.fixture(scope='module')
def cpu_product(tmp_path_factory):
    """sherf_amd._lib pointed at host builds of both libraries; pointer helpers accept CPU tensors; streams are dummies."""
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the bf16 kernels')
    fwd = build_cpu.build('sherf_hipcpu_full', FWD_SOURCES, str(tmp_path_factory.mktemp('hipcpu_full')), compiler=build_cpu.CLANG)
    bwd = build_cpu.build('sherf_hipcpu_bwd', ['bwd_dense.hip', 'bwd_encoder.hip'], str(tmp_path_factory.mktemp('hipcpu_bwd')))
    from sherf_amd import backward_dense
    mp = pytest.MonkeyPatch()
    mp.setattr(_lib, 'LIB_PATH', fwd); mp.setattr(_lib, '_lib', None)
    mp.setattr(_lib, 'LIB_BWD_PATH', bwd); mp.setattr(_lib, '_lib_bwd', None)
    mp.setattr(_lib, 'ptr', lambda t, dtype=None: None if t is None else ctypes.c_void_p(t.data_ptr()))
    mp.setattr(_lib, 'addr', lambda t, dtype=None: None if t is None else t.data_ptr())
    mp.setattr(_lib, 'stream', lambda: ctypes.c_void_p(0))
    mp.setattr(torch.cuda, 'current_stream', lambda dev=None: type('S', (), {'cuda_stream': 0})())
    mp.setattr(torch.cuda, 'synchronize', lambda dev=None: None)
    mp.setattr(backward_dense.HipOps, '_p', staticmethod(lambda m: ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off)))
    yield
    _CACHE.clear()
    mp.undo()


def _modules(precision='bf16x3', training=True):
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=G.smpl(), mlp_precision=precision)
    dec = NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.'); fixtures.load_seeded_state(dec, 'decoder.')
    rend.train(training); dec.train(training)
    rend._side = lambda dev, idx=0: type('X', (), {'cuda_stream': 8 + 8 * idx})()
    return rend, dec


_CACHE = {}


def cpu_render(cfg, precision='bf16x3', training=True, options=None, fx=None):
    """tests.gpu_common.hip_render with CPU tensors (the libraries behind sherf_amd._lib are the host builds)."""
    key = (cfg, precision, training, tuple(sorted((options or {}).items())))
    if fx is None and key in _CACHE:
        return _CACHE[key]
    r = _cpu_render(cfg, precision, training, options, fx)
    if fx is None:
        _CACHE[key] = r
    return r


def _cpu_render(cfg, precision, training, options, fx):
    from sherf_amd.voxel import SparseConvTensor
    fx = fx or G.fixture(cfg)
    rend, dec = _modules(precision, training)
    d = fixtures.to_torch(fx['input_data'])
    spi = G.oracle_render(cfg)['sp_input'] if cfg else O.render_from_fixture(fx, G.seeded_state(), keep=False)['sp_input']
    sp = SparseConvTensor(torch.from_numpy(fx['vertex_feat']), spi['coord'], spi['out_sh'], 1)
    spd = dict(coord=spi['coord'], out_sh=spi['out_sh'], batch_size=1, bounds=spi['bounds'][None])
    opts = dict(fx['options']); opts['mlp_precision'] = precision
    opts.update(options or {})
    with torch.no_grad():
        rgb, depth, acc = rend(torch.from_numpy(fx['planes']), d['obs_img_all'][:, 0], torch.from_numpy(fx['obs_feat']), sp, None, spd, dec,
                               d['ray_o_all'][:, 0].as_subclass(_FakeCuda), d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, opts)
    return dict(rgb=rgb[0], depth=depth[0, :, 0], acc=acc[0, :, 0], last=rend.last, rend=rend, dec=dec)


@pytest.mark.parametrize('cfg', ['tiny', 'tiny_nv'])
def test_frame_matches_oracle_and_reference_golden(cpu_product, cfg):
    o = G.oracle_render(cfg)
    h = cpu_render(cfg)
    ws = h['last']['ws']
    nv = o['valid'].numel()
    assert int(ws['counters'][0]) == nv                                            # shell mask: same sample set
    out = ws['sample_out'][:nv]
    sig_ref = torch.relu(o['sample_sigma'])
    assert float((torch.relu(out[:, 3]) - sig_ref).abs().max() / sig_ref.max()) < 1e-3
    assert float((out[:, :3] - o['sample_rgb']).abs().max()) < 1e-3
    assert G.rel(h['rgb'], o['rgb']) < 1e-3 and G.rel(h['acc'], o['acc']) < 1e-3
    assert torch.allclose(h['depth'], o['depth'], rtol=1e-3, atol=1e-4)
    g = np.load(os.path.join(G.GOLDEN, f'renderer_{cfg}.npz'))                     # outputs of the UNMODIFIED reference
    ref_rgb = torch.from_numpy(g['rgb'])
    assert G.rel(h['rgb'], ref_rgb) < 1e-3
    assert G.rel(h['acc'], torch.from_numpy(g['acc'][:, 0])) < 1e-3
    assert O.psnr(h['rgb'], ref_rgb) > 60.0


@pytest.mark.parametrize('shape', ['4x2', '8x1split', '8x1split2', '8x1persist'])
def test_mlp_shapes_agree_inside_the_frame(cpu_product, shape):
    a = cpu_render('tiny')
    b = cpu_render('tiny', options=dict(mlp_shape=shape))
    assert G.rel(b['rgb'], a['rgb']) < 1e-4 and G.rel(b['acc'], a['acc']) < 1e-4


def test_eval_mode_and_bf16_precision(cpu_product):
    o = G.oracle_render('tiny', training=False)
    h = cpu_render('tiny', training=False)
    assert G.rel(h['rgb'], o['rgb']) < 1e-3 and G.rel(h['acc'], o['acc']) < 1e-3
    o = G.oracle_render('tiny')
    h = cpu_render('tiny', precision='bf16')
    assert G.rel(h['rgb'], o['rgb']) < 1e-1                  # sanity only: plain bf16 is outside the parity bar by design


def test_training_step_through_autograd_matches_reference_gradients(cpu_product, monkeypatch):
    """BASELINE config 5 on the CPU: forward recorded as ONE autograd node (renderer.enable_autograd), stub loss,
    loss.backward() through the native backward pipeline; gradients against the fingerprints of the UNMODIFIED reference's
    gradients (tests/golden/grad_tiny_nv.npz) and against the explicit backward evaluated at our forward point."""
    from sherf_amd import backward as B
    from sherf_amd.voxel import SparseConvTensor
    from tests.bwd_emulator import EmuOps
    cfg = 'tiny_nv'
    fx = G.fixture(cfg)
    ref = np.load(os.path.join(G.GOLDEN, f'grad_{cfg}.npz'))
    rend, dec = _modules()
    rend.enable_autograd = True
    d = fixtures.to_torch(fx['input_data'])
    spi = G.oracle_render(cfg)['sp_input']
    planes = torch.from_numpy(fx['planes']).requires_grad_(True)
    obs_feat = torch.from_numpy(fx['obs_feat']).requires_grad_(True)
    vfeat = torch.from_numpy(fx['vertex_feat']).requires_grad_(True)
    sp = SparseConvTensor(vfeat, spi['coord'], spi['out_sh'], 1)
    spd = dict(coord=spi['coord'], out_sh=spi['out_sh'], batch_size=1, bounds=spi['bounds'][None])
    mean0 = rend.encoder_3d.conv0[1].running_mean.clone()
    rgb, depth, acc = rend(planes, d['obs_img_all'][:, 0], obs_feat, sp, None, spd, dec, d['ray_o_all'][:, 0].as_subclass(_FakeCuda),
                           d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, dict(fx['options']))
    assert rgb.requires_grad and acc.requires_grad and not depth.requires_grad
    assert not torch.equal(rend.encoder_3d.conv0[1].running_mean, mean0)           # a training forward updates the running statistics
    seen = {}
    orig = B.encoder_backward
    monkeypatch.setattr(B, 'encoder_backward', lambda ops, state, ctx, d_levels: seen.update(state=state, ctx=ctx, d_levels=d_levels) or orig(ops, state, ctx, d_levels))
    O.stub_loss(rgb[0], acc[0, :, 0]).backward()
    grads = {'input.planes': planes.grad, 'input.obs_feat': obs_feat.grad, 'input.vertex_feat': vfeat.grad}
    for mod, pre in ((rend, 'renderer.'), (dec, 'decoder.')):
        for n, p in mod.named_parameters():
            if p.grad is not None:
                assert p.grad.shape == p.shape, pre + n
                grads[pre + n] = p.grad
    names = [k for k in ref.files if k not in ('loss', 'ref_cpu_seconds')]
    assert set(names) == set(grads), set(names) ^ set(grads)
    # The encoder's gradients are ill-conditioned on this fixture: one level-3 activation sits within 3e-5 of the ReLU kink and
    # carries a large gradient, so a 3e-5 relative perturbation of the forward (ours differs from the fp32 reference by about
    # that: fixed-point BatchNorm statistics, bf16x3 MLP) moves the reference's OWN encoder gradients by 5-6 % (measured by
    # perturbing the oracle's input).  They are therefore held to a loose bound against the reference and to a tight one
    # against the explicit backward (tests/bwd_emulator.py, itself verified against autograd and the reference in
    # tests/test_backward_math.py / test_backward_dense.py) evaluated at OUR forward point.
    enc = lambda k: 'encoder_3d' in k or k == 'input.vertex_feat'
    for k in names:
        ours, r = O.grad_fingerprint(grads[k].float()), ref[k]
        tn, tv = (0.15, 0.15) if enc(k) else (1e-2, 5e-2)
        assert abs(ours[2] - r[2]) < tn * r[2] + 1e-30, (k, ours[2], r[2])
        assert np.linalg.norm(ours[3:] - r[3:]) < tv * np.linalg.norm(r[3:]) + 1e-30, k
    d_feat_e, g_e = orig(EmuOps(), seen['state'], seen['ctx'], seen['d_levels'])
    assert G.rel(vfeat.grad, d_feat_e.tensor()) < 1e-4
    for k, v in g_e.items():
        assert G.rel(grads[k].reshape(-1), v.reshape(-1)) < 1e-4, k
