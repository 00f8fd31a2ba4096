"""Drop-in counterparts of /root/reference/sherf/training/triplane.py for the hot path:

  * `NeRFDecoder`        (triplane.py:267-316)  -- parameter container; its math runs inside sherf_nerf_mlp.
  * `TriPlaneGenerator`  (triplane.py:29-236)   -- same constructor / mapping / synthesis / forward signatures and
    output dict; the per-frame glue of `synthesis` (triplane.py:105-137, 150-172, 174-217: per-vertex features,
    voxelisation, image reshapes) feeds the MI355X `ImportanceRenderer`.

The feature PRODUCERS (StyleGAN2 backbone, ResNet18 encoders, super-resolution) are outside the hot path
(SURVEY.md section 8f rank 2): they are taken from the reference package when it is importable, or injected.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .ray_sampler import RaySampler
from .renderer import ImportanceRenderer, V, compute_normal  # noqa: F401  (compute_normal: re-exported for the tests / oracle pins)
from .voxel import SparseConvTensor


class NeRFDecoder(nn.Module):
    """triplane.py:267-283 (W=128, skip at layer 4, 39-d position encoding + 32-d feature)."""

    def __init__(self, n_features=32):
        super().__init__()
        W = 128
        self.with_viewdirs = True
        self.skips = [4]
        cin = n_features + 39
        self.pts_linears = nn.ModuleList([nn.Linear(cin, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + cin, W)
                                                               for i in range(7)])
        self.views_linear = nn.Linear(n_features + W + 27, W // 2)
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)

    def forward(self, *a, **k):
        raise RuntimeError('NeRFDecoder is evaluated inside the fused HIP kernel (sherf_nerf_mlp); '
                           'pass it to ImportanceRenderer.forward as `decoder`')


class _GraphedProducer:
    """One per-frame PRODUCER call (tri-plane synthesis, an image encoder, the mapping network: ~150 small library launches each) captured once
    as a hipGraph and replayed: the launches of a frame's producers then cost the host one call and run without inter-launch gaps.  Inference
    only (no autograd, no submodule in training mode -- a capture would advance BatchNorm's running statistics three times and replays never),
    static shapes: the capture is keyed on the arguments' shapes / dtypes / device, the keyword arguments and the module's parameter storage
    (in-place weight updates keep the graph valid: it reads the live parameters; a re-allocated parameter re-captures).

    A capture is only trusted after it has been CHECKED (round 6): the first replay's output must reproduce an eager call on the same
    arguments (to 1e-3 of the output's range) -- an empty or partial capture (work that ran on another stream or device, a forward that does not read its input)
    replays as a no-op and would return the first frame's output for ever.  Any failure (capture error, mismatch) falls back to the eager call
    for good and is reported once through `warnings`."""

    def __init__(self, module, fn):
        self.module, self.fn, self.key, self.graph, self.static_in, self.static_out, self.off = module, fn, None, None, None, None, False
        self.error = None

    def _key(self, args, kw):
        ps = list(self.module.parameters())
        ptrs = (len(ps), ps[0].data_ptr(), ps[-1].data_ptr()) if ps else None
        return (tuple((tuple(a.shape), a.dtype, a.device) if torch.is_tensor(a) else a for a in args), tuple(sorted(kw.items())), ptrs)

    @staticmethod
    def _same(a, b):
        if torch.is_tensor(a):
            if not (torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype):
                return False
            if not a.is_floating_point():
                return bool(torch.equal(a, b))
            # (library convolutions may sum in a different order from call to call: a tolerance, not bits; a poisoned / stale buffer is far outside it)
            scale = float(b.abs().max()) if b.numel() else 0.0
            return bool(torch.isfinite(a).all()) and float((a - b).abs().max() if a.numel() else 0.0) <= 1e-3 * scale + 1e-6
        if isinstance(a, (tuple, list)):
            return isinstance(b, (tuple, list)) and len(a) == len(b) and all(_GraphedProducer._same(x, y) for x, y in zip(a, b))
        return a == b

    @staticmethod
    def _clone(o):
        if torch.is_tensor(o):
            return o.clone()
        if isinstance(o, (tuple, list)):
            return type(o)(_GraphedProducer._clone(x) for x in o)
        return o

    def _capture(self, args, kw):
        dev = next(a.device for a in args if torch.is_tensor(a))
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            self.static_in = [a.clone() if torch.is_tensor(a) else a for a in args]
            with torch.cuda.stream(side):                        # warm-up outside the capture: library workspaces, algorithm searches
                for _ in range(2):
                    eager = self.fn(*self.static_in, **kw)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            import warnings
            with warnings.catch_warnings():
                warnings.filterwarnings('error', message='.*[Gg]raph is empty.*')     # an empty capture is a failed capture
                with torch.cuda.graph(g, capture_error_mode='thread_local'):          # (a DataLoader's pin-memory thread may touch the device meanwhile)
                    self.static_out = self.fn(*self.static_in, **kw)
            # the check: poison the output buffer, replay, compare with the eager result of the same arguments
            eager = self._clone(eager)
            for o in (self.static_out if isinstance(self.static_out, (tuple, list)) else [self.static_out]):
                if torch.is_tensor(o) and o.is_floating_point():
                    o.fill_(float('nan'))
            g.replay()
            if not self._same(self.static_out, eager):
                raise RuntimeError('the captured graph does not reproduce the eager call (empty / partial capture?)')
            self.graph = g

    def __call__(self, *args, **kw):
        if (self.off or torch.is_grad_enabled() or not any(torch.is_tensor(a) for a in args)
                or not all((not torch.is_tensor(a)) or a.device.type == 'cuda' for a in args)
                or any(m.training for m in self.module.modules())):
            return self.fn(*args, **kw)
        try:
            key = self._key(args, kw)
            if key != self.key:
                self.graph = None
                self._capture(args, kw)
                self.key = key
            for dst, src in zip(self.static_in, args):
                if torch.is_tensor(dst):
                    dst.copy_(src)
            self.graph.replay()
            return self._clone(self.static_out)                      # (the graph's output buffer is rewritten by the next replay)
        except Exception as ex:                                      # capture not possible here (an op that synchronises, an old runtime): eager from now on
            self.off, self.graph, self.key = True, None, None
            self.error = f'{type(ex).__name__}: {str(ex)[:200]}'
            try:
                torch.cuda.synchronize()                             # (a failed capture leaves nothing in flight; surface anything it did leave here)
            except Exception:
                pass
            import warnings
            warnings.warn(f'sherf_amd: hipGraph replay of {type(self.module).__name__} disabled, eager calls from now on ({self.error})')
            return self.fn(*args, **kw)


class TriPlaneGenerator(nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, use_1d_feature, use_2d_feature, use_3d_feature, use_trans, use_NeRF_decoder,
                 img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={}, rendering_kwargs={}, sr_kwargs={},
                 backbone=None, encoder_2d=None, encoder_2d_feature=None, superresolution=None, smpl=None, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.renderer = ImportanceRenderer(use_1d_feature, use_2d_feature, use_3d_feature, use_trans, use_NeRF_decoder, smpl=smpl)
        self.ray_sampler = RaySampler()
        self.encoder_2d = encoder_2d
        self.encoder_2d_feature = encoder_2d_feature
        self.conv1d_projection = nn.Conv1d(96, 32, 1)
        self.backbone = backbone
        self.superresolution = superresolution
        # the per-frame producers (experimental): ResNet-18 encoders and the StyleGAN2 tri-plane generator, built
        # as in triplane.py:55-58 unless the caller injects its own modules
        if self.encoder_2d is None:
            from .resnet import ResNet18Classifier
            self.encoder_2d = ResNet18Classifier()
        if self.encoder_2d_feature is None:
            from .resnet import ResNet18Classifier
            self.encoder_2d_feature = ResNet18Classifier()
        if self.backbone is None:
            from .stylegan2 import Generator as StyleGAN2Backbone
            self.backbone = StyleGAN2Backbone(z_dim, c_dim, w_dim, img_resolution=256, img_channels=32 * 3, mapping_kwargs=mapping_kwargs,
                                              **synthesis_kwargs)
        if not use_NeRF_decoder:
            raise NotImplementedError('OSGDecoder path is unused by SHERF (--use_nerf_decoder True in every script)')
        self.decoder = NeRFDecoder(32)
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self.use_1d_feature, self.use_2d_feature, self.use_3d_feature = use_1d_feature, use_2d_feature, use_3d_feature
        self._last_planes = None
        # vertex features + voxelisation as two HIP launches (csrc/glue.hip) instead of ~40 tensor ops, when no
        # gradient is being recorded (training keeps the tensor-op glue so that autograd reaches the image encoders)
        # the per-frame glue (triplane.py:105-137, 174-217) as two HIP launches whenever no gradient is recorded (csrc/glue.hip; checked on the
        # MI355X against the unmodified reference's glue outputs, tests/test_gpu_glue.py): the default since round 4
        self.fused_glue = os.environ.get('SHERF_FUSED_GLUE', '1') == '1'
        # round 5: the producers of an inference frame replayed as hipGraphs (_GraphedProducer).  MI355X, full-size generator
        # (profiles/r05_call_i_*): forward() 8.0 -> 5.9 ms -- tri-plane synthesis 4.8 -> 2.65 ms, ResNet-18 code 1.18 -> 0.70 ms -- same image.
        # On by default (SHERF_GRAPH_PRODUCERS=0 or `graph_producers = False` turn it off; a capture that fails falls back to eager calls)
        self.graph_producers = os.environ.get('SHERF_GRAPH_PRODUCERS', '1') == '1'
        self.__dict__['_graphed'] = {}

    def __getstate__(self):
        # snapshots (training_loop.py:563-579 pickles the generator) and deep copies (G_ema) carry no captured graphs and no device memo
        state = dict(self.__dict__)
        state['_graphed'] = {}
        state.pop('_out_sh_memo', None)
        return state

    def _producer(self, name, module, fn):
        """`fn` (a bound call of `module`), through its hipGraph when graph_producers is on and no gradient is recorded."""
        if not getattr(self, 'graph_producers', False) or torch.is_grad_enabled():
            return fn
        gp = self.__dict__.setdefault('_graphed', {})
        if name not in gp or gp[name].module is not module:
            gp[name] = _GraphedProducer(module, fn)
        return gp[name]

    def mapping(self, z, c, input_img=None, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        z = self._producer('encoder_2d', self.encoder_2d, self.encoder_2d)(input_img)
        if self.rendering_kwargs.get('c_gen_conditioning_zero', True):
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0), truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    # ---- triplane.py:111-137 ------------------------------------------------------------------
    def vertex_features(self, input_data, obs_input_img, obs_input_feature):
        """Per-vertex 32-d features (zero for back-facing vertices) + the back-face mask."""
        smpl = self.renderer.SMPL_NEUTRAL
        verts = input_data['obs_vertices'].float()                                            # [1,V,3]
        Rc, Tc, K = input_data['obs_R_all'][:, 0].float(), input_data['obs_T_all'][:, 0].float(), input_data['obs_K_all'][:, 0].float()
        cam = torch.matmul(verts, Rc.transpose(1, 2)) + Tc.transpose(1, 2)                    # [1,V,3]
        normal = compute_normal(verts, smpl['f'])
        ncam = torch.matmul(normal, Rc.transpose(1, 2))
        mask = (ncam * cam).sum(-1) < 0                                                       # renderer.py:693-695
        h = torch.matmul(cam, K.transpose(1, 2))
        uv = h[..., :2] / (h[..., 2:] + 1e-5)
        Himg, Wimg = obs_input_img.shape[-2:]
        g = 2.0 * uv.unsqueeze(2) / torch.tensor([Wimg, Himg], device=uv.device, dtype=torch.float32) - 1.0
        vf = F.grid_sample(obs_input_feature, g, align_corners=True)[..., 0].permute(0, 2, 1)
        vrgb = F.grid_sample(obs_input_img, g, align_corners=True)[..., 0].permute(0, 2, 1)
        sh = vrgb.shape
        vrgb = self.renderer.rgb_enc(vrgb.reshape(-1, 3)).reshape(*sh[:2], 33)[..., :32]
        f3d = self.conv1d_projection(torch.cat((vf, vrgb), -1).permute(0, 2, 1)).permute(0, 2, 1)
        f3d = f3d * mask.unsqueeze(-1).to(f3d.dtype)                                          # triplane.py:126
        return f3d, mask

    def fused_vertex_features(self, input_data, obs_input_img, obs_input_feature):
        """`vertex_features` as ONE launch (sherf_vertex_features): -> (f3d [1,V,32], front mask [1,V] bool)."""
        dev = obs_input_img.device
        smpl = self.renderer._smpl(dev)
        f32 = lambda t: t.detach().float().contiguous()
        verts = f32(input_data['obs_vertices']).view(V, 3)
        Rc, Tc, K = f32(input_data['obs_R_all']).view(9), f32(input_data['obs_T_all']).view(3), f32(input_data['obs_K_all']).view(9)
        feat, img = f32(obs_input_feature), f32(obs_input_img)
        Wp, bp = f32(self.conv1d_projection.weight).view(32, 96), f32(self.conv1d_projection.bias)
        f3d = torch.empty(V, 32, device=dev, dtype=torch.float32)
        front = torch.empty(V, device=dev, dtype=torch.uint8)
        P = _lib.ptr
        _lib.call('sherf_vertex_features', P(verts), P(smpl['f_i32']), P(smpl['last_face_i32']), V, P(Rc), P(Tc), P(K), P(feat), feat.shape[-2],
                  feat.shape[-1], P(img), img.shape[-2], img.shape[-1], P(Wp), P(bp), P(f3d), P(front), _lib.stream())
        return f3d.unsqueeze(0), front.bool().unsqueeze(0)

    def fused_prepare_sp_input(self, vertex, xyz):
        """`prepare_sp_input` as ONE launch (sherf_voxelize) + the read-back of the three shape integers."""
        dev = xyz.device
        f32 = lambda t: t.detach().float().contiguous()
        tv, can = f32(vertex).view(V, 3), f32(xyz).view(V, 3)
        bounds = torch.empty(2, 3, device=dev, dtype=torch.float32)
        coord = torch.empty(V, 4, device=dev, dtype=torch.int32)
        out_sh = torch.empty(3, device=dev, dtype=torch.int32)
        P = _lib.ptr
        _lib.call('sherf_voxelize', P(tv), P(can), V, P(bounds), P(coord), P(out_sh), _lib.stream())
        # out_sh (three integers the encoder's plan needs on the HOST) depends on the T-pose vertices' bounding box only: read back once per
        # T-pose tensor instead of every frame -- the read-back is a host wait behind everything queued so far (the producers' graphs included)
        key = (vertex.data_ptr(), vertex._version, tuple(vertex.shape))
        memo = self.__dict__.get('_out_sh_memo')
        if memo is None or memo[0] != key:
            memo = self.__dict__['_out_sh_memo'] = (key, out_sh.tolist(), vertex)          # (the tensor is kept: its address cannot be recycled)
        return {'coord': coord, 'out_sh': list(memo[1]), 'batch_size': 1, 'bounds': bounds.unsqueeze(0)}, None

    def canonical_obs_vertices(self, input_data):
        """coarse_deform_target2c(obs_params, obs_vertices, t_params, smpl_obs_pts) (triplane.py:129-132) through the
        per-vertex affine table built by the HIP SMPL kernels (each vertex is its own nearest vertex)."""
        op = input_data['obs_params']
        verts = input_data['obs_vertices'].float()
        # (the same expression the renderer uses for the posed vertices: bit-equal, so every vertex is recognised as its own nearest one)
        smpl_obs_pts = torch.matmul(verts.detach().contiguous().view(V, 3) - op['Th'].detach().float().contiguous().view(1, 3),
                                    op['R'].detach().float().contiguous().view(3, 3)).view(1, V, 3)
        return self.renderer.coarse_deform_target2c(op, verts, input_data['t_params'], smpl_obs_pts)

    def prepare_sp_input(self, vertex, xyz):
        """triplane.py:174-217 (big_box=True): 5 mm voxel coords of `xyz` inside the +-5 cm box of `vertex`."""
        min_xyz = torch.min(vertex, dim=1)[0] - 0.05
        max_xyz = torch.max(vertex, dim=1)[0] + 0.05
        bounds = torch.cat([min_xyz.unsqueeze(1), max_xyz.unsqueeze(1)], 1)
        dhw = xyz[:, :, [2, 1, 0]]
        min_dhw, max_dhw = min_xyz[:, [2, 1, 0]], max_xyz[:, [2, 1, 0]]
        vs = torch.tensor([0.005, 0.005, 0.005], device=dhw.device)
        coord = torch.round((dhw - min_dhw.unsqueeze(1)) / vs).to(torch.int32)
        out_sh = torch.ceil((max_dhw - min_dhw) / vs).to(torch.int32)
        out_sh = (out_sh | 31) + 1
        sh = dhw.shape
        idx = torch.cat([torch.full([sh[1]], i) for i in range(sh[0])]).to(coord)
        coord = torch.cat([idx[:, None], coord.view(-1, 3)], 1)
        out_sh, _ = torch.max(out_sh, dim=0)
        return {'coord': coord, 'out_sh': out_sh.tolist(), 'batch_size': sh[0], 'bounds': bounds}, None

    def synthesis(self, ws, input_data, c, neural_rendering_resolution=None, use_sr_module=True, update_emas=False,
                  cache_backbone=False, use_cached_backbone=False, test_flag=False, planes=None, **synthesis_kwargs):
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        ray_origins, ray_directions = input_data['ray_o_all'][:, 0], input_data['ray_d_all'][:, 0]
        near, far = input_data['near_all'][:, 0], input_data['far_all'][:, 0]
        N, M, _ = ray_origins.shape
        if planes is None:
            if use_cached_backbone and self._last_planes is not None:
                planes = self._last_planes
            else:
                planes = self._producer('backbone.synthesis', self.backbone.synthesis, self.backbone.synthesis)(ws, update_emas=update_emas, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        obs_input_img = input_data['obs_img_all'][:, 0]
        obs_input_feature = self._producer('encoder_2d_feature', self.encoder_2d_feature, self.encoder_2d_feature)(obs_input_img, extract_feature=True)
        fused = self.fused_glue and not torch.is_grad_enabled() and obs_input_img.shape[0] == 1
        f3d, mask = (self.fused_vertex_features if fused else self.vertex_features)(input_data, obs_input_img, obs_input_feature)
        can = self.canonical_obs_vertices(input_data)
        sp_input, _ = (self.fused_prepare_sp_input if fused else self.prepare_sp_input)(input_data['t_vertices'].float(), can)
        sp = SparseConvTensor(f3d.reshape(-1, f3d.shape[-1]), sp_input['coord'], sp_input['out_sh'], sp_input['batch_size'])
        planes = planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])
        if test_flag:
            self.rendering_kwargs.update({'density_noise': 0})
        rgb, depth, acc = self.renderer(planes, obs_input_img, obs_input_feature, sp, mask, sp_input, self.decoder, ray_origins,
                                        ray_directions, near, far, input_data, self.rendering_kwargs)
        H, W = input_data['obs_img_all'].shape[-2:]
        feature_image = rgb.permute(0, 2, 1).reshape(N, rgb.shape[-1], H, W).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(N, 1, H, W)
        weights_image = acc.permute(0, 2, 1).reshape(N, 1, H, W)
        rgb_image = feature_image[:, :3]
        if use_sr_module:
            if self.superresolution is None:
                raise RuntimeError('no superresolution module attached (every SHERF script passes --use_sr_module False)')
            sr_image = self.superresolution(rgb_image, feature_image, ws, noise_mode=self.rendering_kwargs['superresolution_noise_mode'],
                                            **{k: synthesis_kwargs[k] for k in synthesis_kwargs if k != 'noise_mode'})
        else:
            sr_image = rgb_image
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image, 'weights_image': weights_image}

    def forward(self, input_data, z, c, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, use_sr_module=True,
                update_emas=False, cache_backbone=False, use_cached_backbone=False, test_flag=False, **synthesis_kwargs):
        input_img = input_data['obs_img_all'][:, 0]
        ws = self.mapping(z, c, input_img=input_img, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff,
                          update_emas=update_emas)
        return self.synthesis(ws, input_data, c, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              use_sr_module=use_sr_module, cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone,
                              test_flag=test_flag, **synthesis_kwargs)
