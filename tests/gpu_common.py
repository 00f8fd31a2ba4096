"""Shared harness for the `-m gpu` parity tests: same seeded inputs and weights for the oracle (CPU restatement
pinned to the reference) and for the HIP path behind the drop-in classes. Nothing here reads /root/reference."""
import functools
import json
import os

import numpy as np
import torch

from oracle import fixtures, sherf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@functools.lru_cache(None)
def state(variant='seeded'):
    """name -> tensor for every renderer / decoder parameter: 'seeded' = the adversarial weights, 'ri' = the reference constructors'
    distribution (synthdata/fixtures.py: refinit_param)."""
    shapes = json.load(open(os.path.join(GOLDEN, 'param_shapes.json')))
    vals = {n: fixtures.param_value(variant, n, s, shapes) for n, s in shapes.items()}
    return {n: torch.from_numpy(v) for n, v in vals.items() if v is not None}


def seeded_state():
    return state('seeded')


def state_for(cfg):
    return state(fixtures.variant_of(cfg))


@functools.lru_cache(None)
def smpl():
    from oracle import synth
    return synth.make_synth_smpl(0)


@functools.lru_cache(4)
def fixture(cfg):
    return fixtures.renderer_inputs(cfg, smpl())


@functools.lru_cache(4)
def oracle_render(cfg, training=True):
    return O.render_from_fixture(fixture(cfg), state_for(cfg), training=training, keep=True)


# tests/test_hipcpu_frame.py sets this: the same parity tests then run with CPU tensors against the libraries built for the
# host from the unchanged kernel sources (tests/hipcpu).  Tensors are tagged so the product's "must be on a GPU" checks pass.
CPU_SHIM = False


class _HostTensor(torch.Tensor):
    is_cuda = True


def dev_tensor(t):
    """`.cuda()` of the GPU tests; in CPU-shim mode the tensor stays on the host (floating point ones tagged as device tensors)."""
    if not CPU_SHIM:
        return t.cuda()
    return t.as_subclass(_HostTensor) if t.is_floating_point() else t


def dev_module(m):
    return m if CPU_SHIM else m.cuda()


def plain(t):
    """`.cpu()` of the GPU tests (drops the shim's tag)."""
    return t.detach().as_subclass(torch.Tensor).cpu() if isinstance(t, torch.Tensor) else t


def to_cuda(x):
    if isinstance(x, dict):
        return {k: to_cuda(v) for k, v in x.items()}
    if isinstance(x, np.ndarray):
        return dev_tensor(torch.from_numpy(np.ascontiguousarray(x)))
    if isinstance(x, torch.Tensor):
        return dev_tensor(x)
    return x


def hip_modules(precision='f16x3', variant='seeded', use_trans=True, branches=(True, True, True)):
    """(renderer, decoder) with the variant's weights, one pair per (precision, variant, use_trans, feature branches) for the whole session."""
    return _hip_modules(precision, variant, use_trans, tuple(bool(b) for b in branches))


@functools.lru_cache(None)
def _hip_modules(precision, variant, use_trans=True, branches=(True, True, True)):
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder
    rend = ImportanceRenderer(*branches, use_trans=use_trans, use_NeRF_decoder=True, smpl=smpl(), mlp_precision=precision)
    dec = NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.', variant)
    fixtures.load_seeded_state(dec, 'decoder.', variant)
    if CPU_SHIM:
        rend._side = lambda dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})()
    return dev_module(rend).train(), dev_module(dec).train()


hip_modules.cache_clear = _hip_modules.cache_clear
hip_modules.__wrapped__ = lambda precision='f16x3', variant='seeded', use_trans=True: _hip_modules.__wrapped__(precision, variant, use_trans)      # a fresh, uncached pair


def hip_render(cfg, precision='f16x3', training=True, fx=None, sp_input=None, options=None, use_trans=True, branches=(True, True, True)):
    """Runs sherf_amd.ImportanceRenderer.forward on the fixture; the voxel coordinates come from the oracle's
    prepare_sp_input so this isolates the renderer (the TriPlaneGenerator glue has its own test)."""
    from sherf_amd.voxel import SparseConvTensor
    fx = fx or fixture(cfg)
    rend, dec = hip_modules(precision, fixtures.variant_of(cfg) if cfg in fixtures.CONFIGS else 'seeded', use_trans, branches)
    rend.train(training)
    if sp_input is None:
        sp_input = oracle_render(cfg)['sp_input']
    d = to_cuda(fx['input_data'])
    sp = SparseConvTensor(to_cuda(fx['vertex_feat']), dev_tensor(sp_input['coord']), sp_input['out_sh'], 1)
    spi = dict(coord=dev_tensor(sp_input['coord']), out_sh=sp_input['out_sh'], batch_size=1, bounds=dev_tensor(sp_input['bounds'])[None])
    opts = dict(fx['options'])
    opts['mlp_precision'] = precision
    if options:
        opts.update(options)
    with torch.no_grad():
        rgb, depth, acc = rend(to_cuda(fx['planes']), d['obs_img_all'][:, 0], to_cuda(fx['obs_feat']), sp, None, spi, dec,
                               d['ray_o_all'][:, 0], d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, opts)
    if not CPU_SHIM:
        torch.cuda.synchronize()
    return dict(rgb=plain(rgb[0]), depth=plain(depth[0, :, 0]), acc=plain(acc[0, :, 0]), last=rend.last, rend=rend, dec=dec)


def untile_tokens(tokens, n):
    """tokens[tile][3][8][32][4] -> [n,3,32]."""
    tiles = (n + 31) // 32
    t = tokens[:tiles * 3072].view(tiles, 3, 8, 32, 4).permute(0, 3, 1, 2, 4).reshape(tiles * 32, 3, 32)
    return t[:n]


def untile_extras(extras, n):
    tiles = (n + 31) // 32
    return extras[:tiles * 384].view(tiles, 12, 32).permute(0, 2, 1).reshape(tiles * 32, 12)[:n]


def rel(a, b):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
