"""Golden vectors of SURVEY section 8(f) ranks 3 and 4 from the UNMODIFIED reference, for the `-m gpu` tests (tests/test_gpu_producers.py):
TEST INFRASTRUCTURE, run once in the build container (needs /root/reference):

    python -m oracle.make_golden_trainstep            ->  tests/golden/trainstep_tiny_nv.npz

What is recorded (everything from the reference's own code; cv2 / pytorch_msssim / lpips / pytorch3d / spconv are the documented stand-ins
of oracle/ref_shims):

  * the WHOLE generator (`training.triplane.TriPlaneGenerator`, triplane.py:29) with every parameter / buffer set by NAME from
    synthdata.fixtures.seeded_param -- so the GPU box rebuilds the very same weights from the names alone, no weight file travels;
    `names` / `shapes` = the snapshot contract (`misc.copy_params_and_buffers(require_all=True)`, training_loop.py:207-208);
  * f4: the frame the reference renders with those weights, and the frame it renders after its snapshot has gone through its own
    persistence (pickle as training_loop.py:563-579 -> `legacy.load_network_pkl` -> a freshly constructed generator ->
    `copy_params_and_buffers`), which must be the same bits;
  * f3: one generator step `StyleGAN2Loss.accumulate_gradients(phase='Gmain', ...)` (loss.py:103-176) + the flat-gradient sanitising of
    training_loop.py:365-383: the six returned loss terms and a fingerprint of EVERY parameter gradient (oracle.make_golden.grad_fingerprint);
  * the reference's ill-defined vertex normals (renderer.py:50-63: index assignment with duplicate indices) computed once, so that the
    back-face culling of the per-vertex features is the same set of vertices on both sides.
"""
import copy
import io
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = 'tiny_nv'


def seed_name(n):
    """Name under which synthdata.fixtures seeds generator entry `n` (the renderer / decoder keep the names the renderer fixtures use)."""
    return 'generator.' + n if n.startswith('conv1d_projection.') else n


def load_whole_generator_state(gen, variant='seeded'):
    """Every parameter / buffer of a TriPlaneGenerator (the reference's or sherf_amd's: same names) from its name."""
    from synthdata import fixtures
    fixtures.load_seeded_state(gen.renderer, 'renderer.', variant)
    fixtures.load_seeded_state(gen.decoder, 'decoder.', variant)
    with torch.no_grad():
        for n, t in list(gen.named_parameters()) + list(gen.named_buffers()):
            if n.startswith(('renderer.', 'decoder.')) or n.endswith('resample_filter'):
                continue
            v = fixtures.seeded_param(seed_name(n), t.shape)
            if v is not None:
                t.copy_(torch.from_numpy(np.asarray(v, np.float32).reshape(tuple(t.shape))).to(t.dtype))


def batch(fx):
    """The synthetic training batch: the renderer fixture + a seeded target image / background mask (tests/ref_trainstep_child.py)."""
    from synthdata import fixtures
    d = fixtures.to_torch(fx['input_data'])
    H, W = d['obs_img_all'].shape[-2:]
    g = torch.Generator().manual_seed(4)
    d['img_all'] = torch.rand(1, 1, 3, H, W, generator=g)
    d['bkgd_msk_all'] = (torch.rand(1, 1, H * W, generator=g) > 0.5).to(torch.uint8)
    d['mask_at_box_all'] = d['mask_at_box_all'].bool()
    return d


def gen_kwargs(opts):
    opts = dict(opts)
    opts.update(superresolution_module='training.superresolution.SuperresolutionHybrid2X', sr_antialias=True, c_gen_conditioning_zero=True,
                c_scale=0, superresolution_noise_mode='none', density_noise=0, density_reg=0)
    return dict(z_dim=512, c_dim=0, w_dim=48, use_1d_feature=True, use_2d_feature=True, use_3d_feature=True, use_trans=True, use_NeRF_decoder=True,
                img_resolution=128, img_channels=3, mapping_kwargs=dict(num_layers=2), rendering_kwargs=opts, channel_base=512, channel_max=16,
                num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')


def main():
    sys.path.insert(0, ROOT)
    from oracle import make_golden, fixtures
    from tests.ref_trainstep_child import step
    R, T = make_golden.import_reference()
    import dnnlib
    import legacy
    from torch_utils import misc
    from training import loss as RL
    fx = fixtures.renderer_inputs(CFG)
    d = batch(fx)
    H, W = d['obs_img_all'].shape[-2:]
    kw = dict(class_name='training.triplane.TriPlaneGenerator', **gen_kwargs(fx['options']))

    def build(seed):
        torch.manual_seed(seed)
        G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(False)          # training_loop.py:193
        return G

    G = build(0)
    load_whole_generator_state(G)
    n0 = R.compute_normal(d['obs_vertices'].reshape(1, -1, 3), G.renderer.SMPL_NEUTRAL['f'])
    R.compute_normal = lambda vertices, faces: n0

    def render(G):
        G.eval(); G.renderer.train(); G.decoder.train()                                     # (as the reference renders: training_loop.py:193,321)
        with torch.no_grad():
            out = G(d, torch.zeros(1, 512), torch.zeros(1, 0), use_sr_module=False, noise_mode='const')
        return {k: v.detach().as_subclass(torch.Tensor).clone() for k, v in out.items() if torch.is_tensor(v)}

    out = dict(normals=n0.numpy().astype(np.float32))
    named = list(G.named_parameters()) + list(G.named_buffers())
    out['names'] = np.array([n for n, _ in named])
    out['shapes'] = np.array(['x'.join(str(int(s)) for s in t.shape) for _, t in named])
    out['is_param'] = np.array([i < len(list(G.named_parameters())) for i in range(len(named))])
    out['state_l2'] = np.array([float(t.detach().double().norm()) for _, t in named])           # pins the name -> value rule itself

    # ---- f4: the reference's persistence ----
    snap0 = copy.deepcopy(G)                                                                  # (render updates the BatchNorm running statistics)
    a = render(G)
    buf = io.BytesIO()
    snapshot_data = dict(training_set_kwargs=dict(class_name='training.RenderPeople_dataset.RenderPeopleDatasetBatch'))
    for name, module in [('G', snap0), ('G_ema', snap0)]:                                     # training_loop.py:563-579
        snapshot_data[name] = copy.deepcopy(module).eval().requires_grad_(False).cpu()
    pickle.dump(snapshot_data, buf)
    buf.seek(0)
    data = legacy.load_network_pkl(buf)                                                       # legacy.py:24-62
    G2 = build(1)                                                                             # a different init: every value must come from the file
    misc.copy_params_and_buffers(data['G'], G2, require_all=True)                             # training_loop.py:207-208
    b = render(G2)
    assert torch.equal(a['image_raw'], b['image_raw']) and torch.equal(a['weights_image'], b['weights_image'])
    out['snapshot_bytes'] = np.int64(buf.getbuffer().nbytes)
    out['image_raw'] = a['image_raw'].numpy()
    out['weights_image'] = a['weights_image'].numpy()
    out['image_depth'] = a['image_depth'].numpy()

    # ---- f3: one generator step of the reference's own loss ----
    G3 = build(2)
    misc.copy_params_and_buffers(data['G'], G3, require_all=True)
    G3.eval(); G3.renderer.train(); G3.decoder.train()      # the image encoders in eval mode (a 24 x 40 image shrinks to 1 x 1 maps: no batch statistics)
    loss = RL.StyleGAN2Loss(device=torch.device('cpu'), G=G3, D=None, r1_gamma=0, neural_rendering_resolution_initial=max(H, W))
    opt = torch.optim.Adam([p for p in G3.parameters()], lr=2e-4, betas=(0.0, 0.99), eps=1e-8)        # train.py:261 (G_opt_kwargs)
    terms, grads = step(G3, loss, opt, d, torch.device('cpu'))
    out['loss_terms'] = np.array(terms, np.float64)
    out['grad_names'] = np.array(sorted(grads))
    for n, g in grads.items():
        out['grad.' + n] = make_golden.grad_fingerprint(g.as_subclass(torch.Tensor))
    path = os.path.join(ROOT, 'tests', 'golden', f'trainstep_{CFG}.npz')
    np.savez_compressed(path, **out)
    print(f'{path}: {len(named)} state entries, snapshot {int(out["snapshot_bytes"])} B, loss terms {terms}, {len(grads)} gradient fingerprints, '
          f'{os.path.getsize(path) / 1e3:.0f} KB')


if __name__ == '__main__':
    main()
