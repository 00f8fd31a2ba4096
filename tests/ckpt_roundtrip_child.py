"""Child processes of tests/test_checkpoint_roundtrip.py (the reference's own persistence around our path).

  write <dir>   the UNMODIFIED reference (its own renderer / decoder / sparse convolutions, through the stand-ins of oracle/ref_shims):
                builds `training.triplane.TriPlaneGenerator`, renders one frame, and writes the network snapshot EXACTLY as
                training_loop.py:563-579 does -- `pickle.dump(dict(G=copy.deepcopy(G).eval().requires_grad_(False).cpu(), G_ema=...,
                training_set_kwargs=...))`, the generator being a `@persistence.persistent_class` (triplane.py:29).
  read <dir>    a process WITHOUT pytorch3d / spconv (neither the packages nor the test stand-ins): `sherf_amd.install.install()`,
                then the reference's own resume path -- `legacy.load_network_pkl` (legacy.py:24-62), a fresh generator from
                `dnnlib.util.construct_class_by_name` (training_loop.py:193), `misc.copy_params_and_buffers(require_all=True)`
                (training_loop.py:207-208) -- renders the same frame through this package's kernels (host build, tests/hipcpu) and
                compares with the image the writer rendered before snapshotting; then snapshots the HOSTED generator the same way and
                loads that back too.
Prints one `CKPT_JSON {...}` line."""
import copy
import ctypes
import json
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/sherf'


def gen_kwargs(opts):
    opts = dict(opts)
    opts.update(superresolution_module='training.superresolution.SuperresolutionHybrid2X', sr_antialias=True, c_gen_conditioning_zero=True,
                c_scale=0, superresolution_noise_mode='none')
    return dict(class_name='training.triplane.TriPlaneGenerator', z_dim=512, c_dim=0, w_dim=48, use_1d_feature=True, use_2d_feature=True,
                use_3d_feature=True, use_trans=True, use_NeRF_decoder=True, img_resolution=128, img_channels=3, mapping_kwargs=dict(num_layers=2),
                rendering_kwargs=opts, channel_base=512, channel_max=16, num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')


def snapshot(G, path):
    """training_loop.py:563-579, single process (num_gpus == 1: no check_ddp_consistency)."""
    snapshot_data = dict(training_set_kwargs=dict(class_name='training.RenderPeople_dataset.RenderPeopleDatasetBatch'))
    for name, module in [('G', G), ('G_ema', G)]:
        module = copy.deepcopy(module).eval().requires_grad_(False).cpu()
        snapshot_data[name] = module
        del module
    with open(path, 'wb') as f:
        pickle.dump(snapshot_data, f)


def render(G, d):
    z, c = torch.zeros(1, 512), torch.zeros(1, 0)
    G.eval(); G.renderer.train(); G.decoder.train()      # as tests/ref_dropin_child.py: the reference renders in train mode (training_loop.py:193,321)
    with torch.no_grad():
        out = G(d, z, c, use_sr_module=False, noise_mode='const')
    return {k: v.detach().as_subclass(torch.Tensor).clone() for k, v in out.items() if torch.is_tensor(v)}


def write(out_dir):
    from oracle import make_golden, fixtures
    R, T = make_golden.import_reference()
    import dnnlib
    fx = fixtures.renderer_inputs('tiny')
    d = fixtures.to_torch(fx['input_data'])
    torch.manual_seed(0)
    kw = gen_kwargs(fx['options'])
    G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(False)          # training_loop.py:193
    fixtures.load_seeded_state(G.renderer, 'renderer.'); fixtures.load_seeded_state(G.decoder, 'decoder.')
    # the back-face test of the per-vertex features uses the reference's ill-defined vertex normals (renderer.py:50-63: index assignment
    # with duplicate indices); computed ONCE here and handed to the reader, so both sides cull the same vertices
    normals = R.compute_normal(d['obs_vertices'].reshape(1, -1, 3), G.renderer.SMPL_NEUTRAL['f'])
    R.compute_normal = lambda vertices, faces: normals       # (two evaluations of the ill-defined function need not agree: this render uses the SAME one)
    a = render(G, d)
    snapshot(G, os.path.join(out_dir, 'network-snapshot-000000.pkl'))
    torch.save(dict(image=a['image'], image_raw=a['image_raw'], weights=a['weights_image'], normals=normals,
                    names=[n for n, _ in list(G.named_parameters()) + list(G.named_buffers())]), os.path.join(out_dir, 'writer.pt'))
    print('CKPT_JSON ' + json.dumps(dict(wrote=os.path.getsize(os.path.join(out_dir, 'network-snapshot-000000.pkl')),
                                         n_tensors=len(list(G.named_parameters()) + list(G.named_buffers())))), flush=True)


def read(out_dir):
    # the reference + ONLY the stand-ins a ROCm box would also need for reasons unrelated to this path (torchvision, imageio); no
    # pytorch3d, no spconv: sherf_amd.install provides import stubs and the unpickle-only parameter containers
    shims = tempfile.mkdtemp(prefix='sherf_shims_')
    for m in ('torchvision', 'imageio'):
        os.symlink(os.path.join(ROOT, 'oracle', 'ref_shims', m), os.path.join(shims, m))
    sys.path.insert(0, shims); sys.path.insert(0, REF)
    torch.cuda.current_device = lambda: 0
    torch.Tensor.cuda = lambda self, *a, **k: self
    from synthdata import fixtures, synth
    from tests.hipcpu import build_cpu
    from sherf_amd import _lib
    from sherf_amd.build import SOURCES
    import sherf_amd.renderer as AR
    fwd = build_cpu.build('sherf_hipcpu_full', SOURCES, tempfile.mkdtemp(prefix='sherf_ckpt_'), compiler=build_cpu.CLANG)
    _lib.LIB_PATH, _lib._lib = fwd, None
    _lib.ptr = lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
    _lib.stream = lambda: ctypes.c_void_p(0)
    torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
    torch.cuda.synchronize = lambda dev=None: None
    AR.ImportanceRenderer._side = lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})()
    AR.ImportanceRenderer.SMPL_NEUTRAL = property(lambda self: self._smpl(torch.device('cpu')))
    AR.read_pickle = lambda path: synth.make_synth_smpl(0)

    import sherf_amd.install
    done = sherf_amd.install.install()
    assert sorted(done['stubs']) == ['pytorch3d', 'spconv'], done
    import dnnlib
    import legacy
    from torch_utils import misc
    w = torch.load(os.path.join(out_dir, 'writer.pt'))
    AR.compute_normal = lambda vertices, faces: w['normals']
    fx = fixtures.renderer_inputs('tiny')
    d = fixtures.to_torch(fx['input_data'])
    kw = gen_kwargs(fx['options'])
    res = {}

    def resume(pkl):
        with open(pkl, 'rb') as f:
            data = legacy.load_network_pkl(f)                                            # legacy.py:24-62
        torch.manual_seed(1)                                                             # (a different init: every value must come from the file)
        G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(False)      # training_loop.py:193 -> the hosted generator
        misc.copy_params_and_buffers(data['G'], G, require_all=True)                     # training_loop.py:207-208
        return G, data
    torch.Tensor.is_cuda = property(lambda self: True)                                   # host tensors stand in for device tensors
    G, data = resume(os.path.join(out_dir, 'network-snapshot-000000.pkl'))
    res['hosted_renderer'] = type(G.renderer).__module__ + '.' + type(G.renderer).__name__
    res['generator_class'] = type(G).__module__ + '.' + type(G).__name__
    res['pickled_generator_class'] = type(data['G']).__name__
    layer0 = data['G'].renderer.encoder_3d.conv0._modules['0']
    res['pickled_sparse_layer'] = type(layer0).__module__ + '.' + type(layer0).__name__
    names = [n for n, _ in list(G.named_parameters()) + list(G.named_buffers())]
    res['names_equal'] = names == w['names']
    res['n_tensors'] = len(names)
    # KRSC: the sparse-convolution weights arrive in spconv's [out, kz, ky, kx, in] layout and are consumed as such
    res['sparse_weight_shape'] = list(G.renderer.encoder_3d.conv0[0].weight.shape)
    b = render(G, d)
    rel = lambda x, y: float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-12))
    res.update(image_rel=rel(b['image'], w['image']), weights_rel=rel(b['weights_image'], w['weights']),
               image_range=[float(w['image'].min()), float(w['image'].max())], valid_samples=int(G.renderer.last['ws']['counters'][0]))
    # the hosted generator snapshots and resumes the same way (a run that trains with this package and is resumed later)
    snapshot(G, os.path.join(out_dir, 'network-snapshot-000001.pkl'))
    G2, _ = resume(os.path.join(out_dir, 'network-snapshot-000001.pkl'))
    c = render(G2, d)
    res['second_generation_bit_equal'] = bool(torch.equal(c['image'], b['image']) and torch.equal(c['weights_image'], b['weights_image']))
    print('CKPT_JSON ' + json.dumps(res), flush=True)


if __name__ == '__main__':
    {'write': write, 'read': read}[sys.argv[1]](sys.argv[2])
