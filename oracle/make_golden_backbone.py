"""TEST INFRASTRUCTURE. Golden outputs of the UNMODIFIED reference's StyleGAN2 generator (training/networks_stylegan2.py:538,
`StyleGAN2Backbone` of triplane.py:58) for tests/test_backbone.py, on CPU in the build container (the reference's custom ops fall
back to their `_ref` implementations there):

  * a SMALL generator (64 x 64 x 96 channels, channel_base 1024, channel_max 48, 2 mapping layers) with every parameter / buffer filled
    by synthdata.fixtures.seeded_param under its own name -> ws and the synthesised planes in training mode (un-fused modulation) and
    eval mode (fused), noise_mode const / none, truncation;
  * the name -> shape table of the FULL generator SHERF instantiates (256 x 256 x 96, cbase 32768, cmax 512, map_depth 2): the
    checkpoint contract, checked without running it.

    python -m oracle.make_golden_backbone"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/sherf'
SMALL = dict(z_dim=64, c_dim=0, w_dim=48, img_resolution=64, img_channels=96, mapping_kwargs=dict(num_layers=2), channel_base=1024, channel_max=48,
             num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')
FULL = dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=256, img_channels=96, mapping_kwargs=dict(num_layers=2), channel_base=32768, channel_max=512,
            num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')


def seed_module(m, prefix):
    from synthdata import fixtures
    with torch.no_grad():
        for name, t in list(m.named_parameters()) + list(m.named_buffers()):
            if name.endswith('resample_filter'):
                continue
            t.copy_(torch.from_numpy(np.asarray(fixtures.seeded_param(prefix + name, t.shape), np.float32).reshape(tuple(t.shape))).to(t.dtype))


FULL_FP16 = dict(FULL, num_fp16_res=4, conv_clamp=256)          # train.py:427-428 with --g_num_fp16_res 4: the last four resolutions (32 .. 256) in fp16


class _SaysCuda(torch.Tensor):
    """A tensor that answers 'cuda' where networks_stylegan2.py:423 asks `ws.device.type` -- the reference forces fp32 off-GPU, and this container has no GPU.
    A stand-in like oracle/ref_shims: the reference's SOURCE runs unmodified, in the dtype it would run in on a GPU (fp16 convolutions on the host)."""
    @property
    def device(self):
        import types
        return types.SimpleNamespace(type='cuda')


def main_fp16():
    """Round 6: the full-size generator with the reference's own fp16 path (num_fp16_res = 4, conv_clamp = 256; eval mode, noise const) ->
    tests/golden/backbone_full_fp16.npz (a strided subset + whole-tensor moments of the planes)."""
    sys.path.insert(0, REF)
    from training import networks_stylegan2 as N
    from torch_utils.ops import bias_act as BA, upfirdn2d as UP
    BA._init = lambda: False; UP._init = lambda: False             # (no plugin build here: the `_ref` implementations, as for the fp32 goldens)
    full = N.Generator(**FULL_FP16)
    seed_module(full, 'backbone.')
    zf = torch.from_numpy(np.random.RandomState(5).standard_normal((1, FULL['z_dim'])).astype(np.float32))
    with torch.no_grad():
        full.eval()
        wsf = full.mapping(zf, None)
        img = full.synthesis(wsf.as_subclass(_SaysCuda), noise_mode='const')
    v = torch.Tensor(img.float()).numpy() if isinstance(img, torch.Tensor) else img
    v = np.asarray(v, np.float32)
    fo = {'ws': wsf.numpy(), 'img_eval_const.sub': v[:, ::5, ::9, ::9].copy(),
          'img_eval_const.moments': np.array([v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64), np.square(v, dtype=np.float64).sum()])}
    np.savez_compressed(os.path.join(HERE, '..', 'tests', 'golden', 'backbone_full_fp16.npz'), **fo)
    print('wrote backbone_full_fp16.npz', {k: x.shape for k, x in fo.items()}, 'planes', v.shape, 'dtype', img.dtype, 'abs max', float(np.abs(v).max()))


def main():
    sys.path.insert(0, REF)
    from training import networks_stylegan2 as N                     # the reference's file, untouched
    out = {}
    g = N.Generator(**SMALL)
    seed_module(g, 'backbone.')
    z = torch.from_numpy(np.random.RandomState(3).standard_normal((2, SMALL['z_dim'])).astype(np.float32))
    with torch.no_grad():
        g.train()
        ws = g.mapping(z, None)
        out['ws'] = ws.numpy()
        out['ws_trunc'] = g.mapping(z, None, truncation_psi=0.7, truncation_cutoff=5).numpy()
        out['img_train_const'] = g.synthesis(ws, noise_mode='const').numpy()
        out['img_train_none'] = g.synthesis(ws, noise_mode='none').numpy()
        g.eval()
        out['img_eval_const'] = g.synthesis(ws, noise_mode='const').numpy()
        out['img_eval_fwd'] = g(z, None, noise_mode='const').numpy()
    for k in [k for k in out if k.startswith('img_')]:                 # keep the fixture small: a strided subset + whole-tensor moments
        v = out.pop(k)
        out[k + '.sub'] = v[:, ::7, ::3, ::3].copy()
        out[k + '.moments'] = np.array([v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64), np.square(v, dtype=np.float64).sum()])
    np.savez_compressed(os.path.join(HERE, '..', 'tests', 'golden', 'backbone_small.npz'), **out)
    full = N.Generator(**FULL)
    table = {n: list(t.shape) for n, t in list(full.named_parameters()) + list(full.named_buffers())}
    # round 5: the FULL-size generator SHERF instantiates (28.7 M parameters, planes [1, 96, 256, 256]) RUN once, eval mode (fused modulation),
    # noise_mode const, with the same seeded parameters -> a strided subset + whole-tensor moments of ws and the planes (backbone_full.npz):
    # what tests/test_gpu_producers.py holds the device run of sherf_amd.stylegan2 at full size against
    seed_module(full, 'backbone.')
    zf = torch.from_numpy(np.random.RandomState(5).standard_normal((1, FULL['z_dim'])).astype(np.float32))
    fo = {}
    with torch.no_grad():
        full.eval()
        wsf = full.mapping(zf, None)
        fo['ws'] = wsf.numpy()
        v = full.synthesis(wsf, noise_mode='const').numpy()
    fo['img_eval_const.sub'] = v[:, ::5, ::9, ::9].copy()
    fo['img_eval_const.moments'] = np.array([v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64), np.square(v, dtype=np.float64).sum()])
    np.savez_compressed(os.path.join(HERE, '..', 'tests', 'golden', 'backbone_full.npz'), **fo)
    print('wrote backbone_full.npz', {k: v.shape for k, v in fo.items()}, 'planes', v.shape, 'abs max', float(np.abs(v).max()))
    small = {n: list(t.shape) for n, t in list(g.named_parameters()) + list(g.named_buffers())}
    json.dump(dict(full=table, small=small, num_ws_full=full.num_ws, num_ws_small=g.num_ws), open(os.path.join(HERE, '..', 'tests', 'golden', 'backbone_shapes.json'), 'w'), indent=0)
    print('wrote backbone_small.npz', {k: v.shape for k, v in out.items()}, len(table), 'full entries')


if __name__ == '__main__':
    main_fp16() if sys.argv[1:] == ['fp16'] else main()
