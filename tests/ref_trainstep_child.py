"""Child process of tests/test_reference_trainstep.py: the reference's OWN training step around our path.

The unmodified `training.loss.StyleGAN2Loss.accumulate_gradients` (loss.py:103-176; cv2 / pytorch_msssim / lpips replaced by the
documented stand-ins of oracle/ref_shims) and the weight update of training_loop.py:354-386, restated line for line below (`step`),
are driven for two iterations on one synthetic batch, twice:

  1. on the unmodified reference generator (its own renderer / decoder / sparse convolutions through the stand-ins),
  2. after `sherf_amd.install.install()`: the same reference generator class and the same reference loss object code, now hosting this
     package's renderer -- forward through the HIP kernels' source, backward through the HIP backward kernels' source (host builds,
     tests/hipcpu) --

from the same initial weights.  Compared: the loss terms of both iterations, every parameter's gradient of both iterations, and the
weights after the two Adam steps.  Prints one `TRAINSTEP_JSON {...}` line."""
import ctypes
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def step(G, loss, opt, input_data, device, num_gpus=1):
    """training_loop.py:354-386 for the single 'Gmain' phase SHERF trains (interval 1, batch_gpu 1)."""
    from torch_utils import misc
    phase_real_img = input_data['img_all'][:, 0].to(device).to(torch.float32).split(1)
    phase_real_c = torch.zeros(1, 25).to(device).split(1)
    gen_z = torch.zeros(1, G.z_dim, device=device)                                   # (randn in the reference: z is ignored numerically, triplane.py:75)
    gen_c = torch.zeros(1, 25, device=device)
    opt.zero_grad(set_to_none=True)
    G.requires_grad_(True)
    for real_img, real_c in zip(phase_real_img, phase_real_c):
        out = loss.accumulate_gradients(phase='Gmain', input_data=input_data, real_img=real_img, real_c=real_c, gen_z=gen_z, gen_c=gen_c, gain=1,
                                        cur_nimg=0, use_sr_module=False, recons_loss=True, rank=0)
    G.requires_grad_(False)
    params = [param for param in G.parameters() if param.numel() > 0 and param.grad is not None]
    grads_before = {id(p): p.grad.detach().clone() for p in params}
    flat = torch.cat([param.grad.flatten() for param in params])
    if num_gpus > 1:
        torch.distributed.all_reduce(flat)
        flat /= num_gpus
    misc.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
    grads = flat.split([param.numel() for param in params])
    for param, grad in zip(params, grads):
        param.grad = grad.reshape(param.shape)
    opt.step()
    named = {n: grads_before[id(p)] for n, p in G.named_parameters() if id(p) in grads_before}
    return [float(torch.as_tensor(t).detach().as_subclass(torch.Tensor).reshape(-1)[0]) for t in out], named


def main():
    from oracle import make_golden, fixtures, synth
    from tests.hipcpu import build_cpu
    from sherf_amd import _lib, backward_dense
    import sherf_amd.renderer as AR
    from sherf_amd.build import SOURCES
    tmp = tempfile.mkdtemp(prefix='sherf_trainstep_')
    fwd = build_cpu.build('sherf_hipcpu_full', SOURCES, tmp, compiler=build_cpu.CLANG)
    bwd = build_cpu.build('sherf_hipcpu_bwd', ['bwd_dense.hip', 'bwd_gemm.hip', 'bwd_encoder.hip'], tmp, compiler=build_cpu.CLANG)
    _lib.LIB_PATH, _lib._lib = fwd, None
    _lib.LIB_BWD_PATH, _lib._lib_bwd = bwd, None
    _lib.ptr = lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
    _lib.stream = lambda: ctypes.c_void_p(0)
    torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
    torch.cuda.synchronize = lambda dev=None: None
    backward_dense.HipOps._p = staticmethod(lambda m: ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off))
    AR.ImportanceRenderer._side = lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})()
    AR.ImportanceRenderer.SMPL_NEUTRAL = property(lambda self: self._smpl(torch.device('cpu')))
    AR.read_pickle = lambda path: synth.make_synth_smpl(0)

    R, T = make_golden.import_reference()
    import dnnlib
    from training import loss as RL                                                  # the unmodified reference loss module
    fx = fixtures.renderer_inputs('tiny_nv')
    d = fixtures.to_torch(fx['input_data'])
    H, W = d['obs_img_all'].shape[-2:]
    g = torch.Generator().manual_seed(4)
    d['img_all'] = torch.rand(1, 1, 3, H, W, generator=g)
    d['bkgd_msk_all'] = (torch.rand(1, 1, H * W, generator=g) > 0.5).to(torch.uint8)
    d['mask_at_box_all'] = d['mask_at_box_all'].bool()
    opts = dict(fx['options'])
    opts.update(superresolution_module='training.superresolution.SuperresolutionHybrid2X', sr_antialias=True, c_gen_conditioning_zero=True,
                c_scale=0, superresolution_noise_mode='none', density_noise=0, density_reg=0)
    kw = dict(class_name='training.triplane.TriPlaneGenerator', z_dim=512, c_dim=0, w_dim=48, use_1d_feature=True, use_2d_feature=True,
              use_3d_feature=True, use_trans=True, use_NeRF_decoder=True, img_resolution=128, img_channels=3, mapping_kwargs=dict(num_layers=2),
              rendering_kwargs=opts, channel_base=512, channel_max=16, num_fp16_res=0, conv_clamp=None, fused_modconv_default='inference_only')
    normals = {}

    def build():
        torch.manual_seed(0)
        G = dnnlib.util.construct_class_by_name(**kw).train().requires_grad_(False)
        fixtures.load_seeded_state(G.renderer, 'renderer.'); fixtures.load_seeded_state(G.decoder, 'decoder.')
        # the image encoders in eval mode: their BatchNorm cannot take batch statistics of the 1x1 maps a 24x40 test image shrinks to
        G.eval(); G.renderer.train(); G.decoder.train()
        return G

    def run(G):
        loss = RL.StyleGAN2Loss(device=torch.device('cpu'), G=G, D=None, r1_gamma=0, neural_rendering_resolution_initial=max(H, W))
        opt = torch.optim.Adam([p for p in G.parameters()], lr=2e-4, betas=(0.0, 0.99), eps=1e-8)     # train.py:261 (G_opt_kwargs)
        rec = []
        w0 = {n: p.detach().as_subclass(torch.Tensor).clone() for n, p in G.named_parameters()}
        for it in range(2):
            rec.append(step(G, loss, opt, d, torch.device('cpu')))
        return rec, {n: p.detach().as_subclass(torch.Tensor).clone() - w0[n] for n, p in G.named_parameters()}

    G_ref = build()
    n0 = R.compute_normal(d['obs_vertices'].reshape(1, -1, 3), G_ref.renderer.SMPL_NEUTRAL['f'])   # ill-defined in the reference: computed once
    R.compute_normal = lambda vertices, faces: n0
    rec_ref, w_ref = run(G_ref)

    import sherf_amd.install
    done = sherf_amd.install.install()
    AR.compute_normal = R.compute_normal
    torch.Tensor.is_cuda = property(lambda self: True)
    G_new = build()
    assert type(G_new.renderer).__module__ == 'sherf_amd.renderer' and G_new.renderer.enable_autograd
    rec_new, w_new = run(G_new)

    plain = lambda t: t.detach().as_subclass(torch.Tensor).double()
    res = dict(hosted_renderer=type(G_new.renderer).__module__, loss_module=RL.__file__, iterations=[])
    for it in range(2):
        (l_ref, g_ref), (l_new, g_new) = rec_ref[it], rec_new[it]
        assert set(g_ref) == set(g_new), sorted(set(g_ref) ^ set(g_new))[:5]
        worst, worst_name = 0.0, None
        cos_min = 1.0
        for n in g_ref:
            a, b = plain(g_new[n]), plain(g_ref[n])
            e = float((a - b).abs().max() / (b.abs().max() + 1e-12))
            c = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
            if b.abs().max() > 1e-9:
                cos_min = min(cos_min, c)
            if e > worst and b.abs().max() > 1e-9:
                worst, worst_name = e, n
        res['iterations'].append(dict(loss_ref=l_ref, loss_new=l_new, n_grads=len(g_ref), grad_rel_err_max=worst, grad_rel_err_argmax=worst_name,
                                      grad_cosine_min=cos_min,
                                      renderer_grad_rel={n: float((plain(g_new[n]) - plain(g_ref[n])).abs().max() / (plain(g_ref[n]).abs().max() + 1e-12))
                                                         for n in ('decoder.pts_linears.0.weight', 'renderer.conv1d_reprojection.weight',
                                                                   'renderer.encoder_3d.conv0.0.weight', 'conv1d_projection.weight')}))
    # the two Adam(beta1 = 0) steps move every weight by ~lr * sign(gradient): the updates are compared in the L1 sense over all parameters
    # (entries whose gradient is at rounding level legitimately take opposite signs)
    num = sum(float((w_new[n].double() - w_ref[n].double()).abs().sum()) for n in w_ref)
    den = sum(float(w_ref[n].double().abs().sum()) for n in w_ref)
    res['update_mismatch_l1'] = num / den
    print('TRAINSTEP_JSON ' + json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
