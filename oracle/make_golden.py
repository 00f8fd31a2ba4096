"""TEST INFRASTRUCTURE. Generates tests/golden/*.npz by running the UNMODIFIED reference files
(/root/reference/sherf/training/volumetric_rendering/renderer.py, ray_marcher.py, ray_sampler.py,
triplane.py::NeRFDecoder / prepare_sp_input) on CPU in the build container, through the stand-in
packages in oracle/ref_shims (pytorch3d K-NN, spconv, torchvision, imageio) and three monkeypatches
(no GPU here; SMPL pickle replaced by the synthetic SMPL of oracle/synth.py).

    python -m oracle.make_golden [tiny tiny_nv cfg1]
    python -m oracle.make_golden tiny_ri cfg1_ri        # the reference-init variants only (+ the constructor check)

Inputs and weights are regenerated from seeds by oracle/fixtures.py, so the files hold OUTPUTS only.
/root/reference does not exist on the GPU box: nothing at test time imports this module.
"""
import os
import sys
import time
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/sherf'


def import_reference():
    sys.path.insert(0, os.path.join(HERE, 'ref_shims'))
    sys.path.insert(0, REF)
    torch.cuda.current_device = lambda: 0
    torch.Tensor.cuda = lambda self, *a, **k: self
    from training.volumetric_rendering import renderer as R
    from oracle import synth
    orig = R.SMPL_to_tensor
    R.SMPL_to_tensor = lambda params, device: orig(params, torch.device('cpu'))
    R.read_pickle = lambda path: synth.make_synth_smpl(0)
    import training.triplane as T
    T.SMPL_to_tensor = R.SMPL_to_tensor
    T.read_pickle = R.read_pickle
    return R, T


def run(cfg_name, R, T, out_dir, use_trans=True, branches=(True, True, True)):
    """use_trans=False (round 5): the same frame through the unmodified reference built WITHOUT its transformer (renderer.py:261, 427) ->
    renderer_<cfg>_notrans.npz (final image + per-sample rgb / sigma + the discrete selections; the reference-init variant only).
    branches (round 6): the constructor's use_1d / use_2d / use_3d_feature switches (renderer.py:261-269, 405-425) -> renderer_<cfg>_f<abc>.npz."""
    from oracle import fixtures
    fx = fixtures.renderer_inputs(cfg_name)
    c = fx['cfg']
    torch.manual_seed(0)
    rend = R.ImportanceRenderer(*[bool(b) for b in branches], use_trans=use_trans, use_NeRF_decoder=True)
    dec = T.NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.', fixtures.variant_of(cfg_name))
    fixtures.load_seeded_state(dec, 'decoder.', fixtures.variant_of(cfg_name))
    rend.train(); dec.train()        # the reference never calls .eval() before test (training_loop.py:193,321)

    d = fixtures.to_torch(fx['input_data'])
    planes = torch.from_numpy(fx['planes'])
    obs_img = d['obs_img_all'][:, 0]
    obs_feat = torch.from_numpy(fx['obs_feat'])
    vfeat = torch.from_numpy(fx['vertex_feat'])

    cap = {}
    # --- a17 glue done with the reference's own functions (triplane.py:129-137) ---
    smpl_obs_pts = torch.matmul(d['obs_vertices'] - d['obs_params']['Th'], d['obs_params']['R'])
    obs_can = rend.coarse_deform_target2c(d['obs_params'], d['obs_vertices'], d['t_params'], smpl_obs_pts)
    sp_input, _ = T.TriPlaneGenerator.prepare_sp_input(types.SimpleNamespace(), d['t_vertices'].clone(), obs_can)
    import spconv.pytorch as spconv
    sp = spconv.core.SparseConvTensor(vfeat, sp_input['coord'], sp_input['out_sh'], sp_input['batch_size'])
    cap['obs_vertex_canonical'] = obs_can[0].numpy()
    cap['sp_coord'] = sp_input['coord'].numpy()
    cap['sp_out_sh'] = np.asarray(sp_input['out_sh'])
    cap['sp_bounds'] = sp_input['bounds'].numpy()

    # --- capture intermediates of the unmodified forward by wrapping bound methods ---
    knn_calls = []
    orig_knn = R.knn_points

    def knn_rec(a, b, K=1):
        out = orig_knn(a, b, K=K)
        knn_calls.append((out[0].clone(), out[1].clone()))
        return out
    R.knn_points = knn_rec

    def wrap(obj, name, key, pick):
        f = getattr(obj, name)

        def g(*a, **k):
            out = f(*a, **k)
            cap.setdefault(key, []).append(pick(out, a, k))
            return out
        setattr(obj, name, g)

    wrap(rend, 'coarse_deform_target2c', 't2c', lambda o, a, k: [x.clone() for x in (o if isinstance(o, tuple) else (o,))])
    wrap(rend, 'coarse_deform_c2source', 'c2s', lambda o, a, k: o[1].clone())
    wrap(rend, 'projection', 'uv', lambda o, a, k: o.clone())
    wrap(rend, 'run_model', 'run', lambda o, a, k: dict(rgb=o['rgb'].clone(), sigma=o['sigma'].clone(), f2d=a[1].clone(), f3d=a[2].clone()))
    wrap(rend, 'get_grid_coords', 'grid', lambda o, a, k: o.clone())
    enc_fwd = rend.encoder_3d.forward

    def enc_rec(x, g):
        out = enc_fwd(x, g)
        cap['f3d_raw'] = out.clone()
        return out
    rend.encoder_3d.forward = enc_rec
    if use_trans:
        tr_fwd = rend.transformer.forward

        def tr_rec(x):
            out = tr_fwd(x)
            cap['tokens_in'] = x.clone(); cap['tokens_out'] = out.clone()
            return out
        rend.transformer.forward = tr_rec
    rm = rend.ray_marcher.run_forward

    def rm_rec(*a):
        out = rm(*a)
        cap['weights'] = out[2].clone()
        return out
    rend.ray_marcher.run_forward = rm_rec

    t0 = time.time()
    with torch.no_grad():
        rgb, depth, acc = rend(planes, obs_img, obs_feat, sp, None, sp_input, dec,
                               d['ray_o_all'][:, 0], d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0],
                               d, fx['options'])
    dt = time.time() - t0
    R.knn_points = orig_knn

    dist1, idx1 = knn_calls[0]
    mask = (dist1.view(-1) < 0.05 ** 2)
    valid = torch.nonzero(mask)[:, 0]
    out = dict(
        rgb=rgb[0].numpy(), depth=depth[0].numpy(), acc=acc[0].numpy(),
        mask_bits=np.packbits(mask.numpy().astype(np.uint8)),
        n_samples=np.int64(mask.numel()), n_valid=np.int64(valid.numel()),
        sample_rgb=cap['run'][0]['rgb'][0].numpy(), sample_sigma=cap['run'][0]['sigma'][0, :, 0].numpy(),
        vert_id=idx1.view(-1)[valid].numpy().astype(np.int32),
        vert_d2=dist1.view(-1)[valid].numpy(),
        ref_cpu_seconds=np.float64(dt),
        sp_coord=cap['sp_coord'], sp_out_sh=cap['sp_out_sh'], sp_bounds=cap['sp_bounds'],
    )
    all_on = all(branches)
    if use_trans and all_on and c['H'] * c['W'] * c['S'] <= 64 * 64 * 32:   # full per-stage intermediates only for the tiny configs
        out.update(
            x_c=cap['t2c'][0][0][0].numpy(), v_c=cap['t2c'][0][1][0].numpy(),
            t_vert_id=knn_calls[2][1].view(-1).numpy().astype(np.int32),
            x_w=cap['c2s'][0][0].numpy(), uv=cap['uv'][0].reshape(-1, 2).numpy(),
            f2d=cap['run'][0]['f2d'][0].numpy(), f3d=cap['run'][0]['f3d'][0].numpy(),
            f3d_raw=cap['f3d_raw'][0].numpy(), grid=cap['grid'][0].reshape(-1, 3).numpy(),
            tokens_in=cap['tokens_in'].numpy(), tokens_out=cap['tokens_out'].numpy(),
            weights=cap['weights'][0, :, :, 0].numpy(),
            obs_vertex_canonical=cap['obs_vertex_canonical'],
        )
    tag = ('' if use_trans else '_notrans') + ('' if all_on else '_f' + ''.join('1' if b else '0' for b in branches))
    path = os.path.join(out_dir, f'renderer_{cfg_name}{tag}.npz')
    np.savez_compressed(path, **out)
    print(f'{cfg_name}: R={rgb.shape[1]} Nv={valid.numel()}/{mask.numel()} ({valid.numel()/mask.numel():.3%}) '
          f'ref forward {dt:.2f}s  rgb range [{rgb.min():.3f},{rgb.max():.3f}] acc max {acc.max():.3f} -> {path} '
          f'({os.path.getsize(path)/1e6:.2f} MB)')


def run_refinit_check(R, T, out_dir):
    """What `fixtures.refinit_param` claims about the reference's constructors, recorded from modules the UNMODIFIED reference builds
    under torch.manual_seed(0): per tensor its kind (ones / zeros / uniform) and, for the uniform ones, max |value| * sqrt(fan_in)
    (must be <= 1: the kaiming_uniform(a=sqrt(5)) bound) and std * sqrt(3 fan_in) (-> 1).  tests/test_oracle_golden.py compares."""
    import json
    torch.manual_seed(0)
    rend = R.ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True)
    dec = T.NeRFDecoder(32)
    rec = {}
    for prefix, mod in (('renderer.', rend), ('decoder.', dec)):
        named = list(mod.named_parameters()) + list(mod.named_buffers())
        shapes = {prefix + n: tuple(t.shape) for n, t in named}
        for n, t in named:
            name = prefix + n
            if name.split('.')[-1] in ('_freqs', '_phases', 'num_batches_tracked'):
                continue
            v = t.detach().double()
            if bool((v == 1).all()):
                rec[name] = dict(kind='ones')
            elif bool((v == 0).all()):
                rec[name] = dict(kind='zeros')
            else:
                w = shapes.get(name[:-len('bias')] + 'weight') if name.endswith('.bias') else tuple(t.shape)
                fan_in = int(np.prod(w[1:]))
                rec[name] = dict(kind='uniform', fan_in=fan_in, max_scaled=float(v.abs().max() * np.sqrt(fan_in)),
                                 std_scaled=float(v.std() * np.sqrt(3 * fan_in)) if v.numel() > 1 else None, numel=int(v.numel()))
    path = os.path.join(out_dir, 'refinit_reference_constructors.json')
    json.dump(rec, open(path, 'w'), indent=0, sort_keys=True)
    print('refinit check ->', path, {k: sum(1 for r in rec.values() if r['kind'] == k) for k in ('ones', 'zeros', 'uniform')})


GRAD_SAMPLE = 64      # entries kept per gradient tensor (strided), next to its sum / abs-sum / L2 norm


def grad_fingerprint(g):
    """Compact golden for one gradient tensor: [sum, sum|.|, L2] in float64 + GRAD_SAMPLE strided entries."""
    g = g.detach().reshape(-1).double()
    n = g.numel()
    idx = torch.linspace(0, n - 1, min(GRAD_SAMPLE, n)).round().long()
    return np.concatenate([np.array([float(g.sum()), float(g.abs().sum()), float(g.norm())]), g[idx].numpy()])


def loss_targets(shape_rgb, shape_acc):
    """Seeded targets of the stub training loss of BASELINE config 5: L = MSE(rgb, T_rgb) + MSE(acc, T_acc)."""
    rs = np.random.RandomState(11)
    return (torch.from_numpy(rs.uniform(-1, 1, tuple(shape_rgb)).astype(np.float32)),
            torch.from_numpy(rs.uniform(0, 1, tuple(shape_acc)).astype(np.float32)))


def run_grad(cfg_name, R, T, out_dir, use_trans=True):
    """Golden GRADIENTS: forward + backward of the unmodified reference (training mode, density_noise 0) under the stub
    loss, w.r.t. every renderer / decoder parameter and the three feature inputs (tri-planes, 2-D feature map, per-vertex
    voxel features).  Stored as fingerprints (grad_fingerprint) -- the full set would be 10 MB."""
    from oracle import fixtures
    fx = fixtures.renderer_inputs(cfg_name)
    torch.manual_seed(0)
    rend = R.ImportanceRenderer(True, True, True, use_trans=use_trans, use_NeRF_decoder=True)      # (use_trans=False, round 6: grad_<cfg>_notrans.npz)
    dec = T.NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.')
    fixtures.load_seeded_state(dec, 'decoder.')
    rend.train(); dec.train()
    d = fixtures.to_torch(fx['input_data'])
    planes = torch.from_numpy(fx['planes']).requires_grad_(True)
    obs_img = d['obs_img_all'][:, 0]
    obs_feat = torch.from_numpy(fx['obs_feat']).requires_grad_(True)
    vfeat = torch.from_numpy(fx['vertex_feat']).requires_grad_(True)
    with torch.no_grad():
        smpl_obs_pts = torch.matmul(d['obs_vertices'] - d['obs_params']['Th'], d['obs_params']['R'])
        obs_can = rend.coarse_deform_target2c(d['obs_params'], d['obs_vertices'], d['t_params'], smpl_obs_pts)
    sp_input, _ = T.TriPlaneGenerator.prepare_sp_input(types.SimpleNamespace(), d['t_vertices'].clone(), obs_can)
    import spconv.pytorch as spconv
    sp = spconv.core.SparseConvTensor(vfeat, sp_input['coord'], sp_input['out_sh'], sp_input['batch_size'])
    t0 = time.time()
    rgb, depth, acc = rend(planes, obs_img, obs_feat, sp, None, sp_input, dec, d['ray_o_all'][:, 0], d['ray_d_all'][:, 0],
                           d['near_all'][:, 0], d['far_all'][:, 0], d, fx['options'])
    t_rgb, t_acc = loss_targets(rgb.shape, acc.shape)
    loss = ((rgb - t_rgb) ** 2).mean() + ((acc - t_acc) ** 2).mean()
    loss.backward()
    dt = time.time() - t0
    out = {'loss': np.float64(loss.item()), 'ref_cpu_seconds': np.float64(dt)}
    for name, t in (('input.planes', planes), ('input.obs_feat', obs_feat), ('input.vertex_feat', vfeat)):
        out[name] = grad_fingerprint(t.grad)
    n_none = 0
    for prefix, mod in (('renderer.', rend), ('decoder.', dec)):
        for name, p_ in mod.named_parameters():
            if p_.grad is None:
                n_none += 1
                continue
            out[prefix + name] = grad_fingerprint(p_.grad)
    path = os.path.join(out_dir, f'grad_{cfg_name}.npz' if use_trans else f'grad_{cfg_name}_notrans.npz')
    np.savez_compressed(path, **out)
    print(f'grad {cfg_name}: loss {loss.item():.6f}, {len(out) - 2} gradient fingerprints ({n_none} parameters without grad), '
          f'ref fwd+bwd {dt:.2f}s -> {path} ({os.path.getsize(path)/1e3:.0f} KB)')


def run_glue(R, T, out_dir, cfg_name='tiny'):
    """Golden for the TriPlaneGenerator.synthesis glue (triplane.py:105-126): per-vertex features + back-face mask,
    produced with the reference's own projection / compute_normal / rgb_enc and a seeded conv1d_projection."""
    import torch.nn.functional as F
    from oracle import fixtures
    fx = fixtures.renderer_inputs(cfg_name)
    d = fixtures.to_torch(fx['input_data'])
    rend = R.ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True)
    conv = torch.nn.Conv1d(96, 32, 1)
    fixtures.load_seeded_state(conv, 'generator.conv1d_projection.')
    obs_img = d['obs_img_all'][:, 0]
    obs_feat = torch.from_numpy(fx['obs_feat'])
    with torch.no_grad():
        uv, mask = rend.projection(d['obs_vertices'].reshape(1, -1, 3), d['obs_R_all'], d['obs_T_all'], d['obs_K_all'], rend.SMPL_NEUTRAL['f'])
        uv = uv.view(-1, *uv.shape[2:])
        uv_ = 2.0 * uv.unsqueeze(2).type(torch.float32) / torch.Tensor([obs_img.shape[-1], obs_img.shape[-2]]) - 1.0
        vf = F.grid_sample(obs_feat, uv_, align_corners=True)[..., 0].permute(0, 2, 1)
        vrgb = F.grid_sample(obs_img, uv_, align_corners=True)[..., 0].permute(0, 2, 1)
        sh = vrgb.shape
        vrgb = rend.rgb_enc(vrgb.reshape(-1, 3)).reshape(*sh[:2], 33)[..., :32]
        f3d = conv(torch.cat((vf, vrgb), -1).permute(0, 2, 1)).permute(0, 2, 1)
        f3d[mask == 0] = 0
    path = os.path.join(out_dir, f'glue_{cfg_name}.npz')
    np.savez_compressed(path, vertex_feat=f3d[0].numpy(), front_mask=mask[0].numpy())
    print('glue ->', path, 'front-facing', float(mask.float().mean()))


def run_small_units(R, T, out_dir):
    """Golden vectors for the standalone API pieces: RaySampler, MipRayMarcher2, PositionalEncoding, linspace."""
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.volumetric_rendering.ray_marcher import MipRayMarcher2
    from training.volumetric_rendering import math_utils
    rs = np.random.RandomState(7)
    out = {}
    # RaySampler (ray_sampler.py:24-61)
    c2w = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    from scipy.spatial.transform import Rotation
    c2w[:, :3, :3] = Rotation.from_rotvec(rs.normal(0, 0.5, (2, 3))).as_matrix()
    c2w[:, :3, 3] = rs.normal(0, 1, (2, 3))
    intr = np.tile(np.array([[1.2, 0.03, 0.5], [0, 1.1, 0.52], [0, 0, 1]], np.float32), (2, 1, 1))
    o, dd = RaySampler()(torch.from_numpy(c2w), torch.from_numpy(intr), 8)
    out.update(rs_c2w=c2w, rs_intr=intr, rs_origins=o.numpy(), rs_dirs=dd.numpy())
    # MipRayMarcher2 (ray_marcher.py:25-64), incl. sigma=-80 fill, white_back, all-empty ray (NaN depth path)
    Rr, S = 37, 13
    colors = rs.uniform(0, 1, (1, Rr, S, 3)).astype(np.float32)
    dens = (rs.normal(0, 20, (1, Rr, S, 1))).astype(np.float32)
    dens[0, :5] = -80.0
    near = rs.uniform(1, 2, (1, Rr, 1, 1)).astype(np.float32); far = near + rs.uniform(0.5, 2, (1, Rr, 1, 1)).astype(np.float32)
    depths = near + np.linspace(0, 1, S, dtype=np.float32).reshape(1, 1, S, 1) * (far - near)
    rd = rs.normal(0, 1, (1, Rr, 3)).astype(np.float32)
    for wb in (False, True):
        rgb, dep, w = MipRayMarcher2()(torch.from_numpy(colors), torch.from_numpy(dens), torch.from_numpy(depths),
                                       torch.from_numpy(rd), dict(clamp_mode='relu', white_back=wb))
        out.update({f'mrm_rgb_{int(wb)}': rgb.numpy(), f'mrm_depth_{int(wb)}': dep.numpy(), f'mrm_w_{int(wb)}': w.numpy()})
    out.update(mrm_colors=colors, mrm_dens=dens, mrm_depths=depths, mrm_rd=rd)
    # PositionalEncoding (renderer.py:875-916) and linspace (math_utils.py:101-118)
    x = rs.normal(0, 1, (11, 3)).astype(np.float32)
    for F_ in (4, 5, 6):
        out[f'pe_{F_}'] = R.PositionalEncoding(num_freqs=F_)(torch.from_numpy(x)).numpy()
    out['pe_x'] = x
    a = rs.uniform(0, 2, (1, 9, 1)).astype(np.float32); b = a + rs.uniform(0, 3, (1, 9, 1)).astype(np.float32)
    out['ls_a'] = a; out['ls_b'] = b
    out['ls_out'] = math_utils.linspace(torch.from_numpy(a), torch.from_numpy(b), 17).numpy()
    path = os.path.join(out_dir, 'units.npz')
    np.savez_compressed(path, **out)
    print('units ->', path)


def reference_ray_functions():
    """`get_rays` and `get_near_far` EXACTLY as the reference defines them (training/RenderPeople_dataset.py:14-27, 68-101):
    the module itself cannot be imported here (cv2 / imageio / the SMPL assets are absent), but the two functions need numpy
    only, so their source text is cut out of the unmodified file with `ast` and executed."""
    import ast
    path = os.path.join(REF, 'training', 'RenderPeople_dataset.py')
    src = open(path).read()
    ns = {'np': np}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ('get_rays', 'get_near_far'):
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), ns)
    return ns['get_rays'], ns['get_near_far']


def run_rays(out_dir):
    """Golden vectors for SURVEY rows a1 / a2: the reference's get_rays + get_near_far + the casts and (0, 1) packing of
    sample_ray_RenderPeople_batch (RenderPeople_dataset.py:121-134, restated here line for line because that function also
    needs cv2) on five seeded cameras: the `tiny` fixture's own camera, a wide one whose frame overshoots the box (misses),
    a non-square frame, an axis-aligned camera (exact zeros in ray_d -> the reference's in-place 1e-8 patch) and cfg1's."""
    from oracle import synth, fixtures
    get_rays, get_near_far = reference_ray_functions()
    smpl = synth.make_synth_smpl(0)
    out = {}
    cases = []
    for cfg in ('tiny', 'cfg1'):
        d = fixtures.renderer_inputs(cfg, smpl)['input_data']
        verts = d['vertices'][0]
        H = int(round(np.sqrt(d['ray_o_all'].shape[2])))
        wb = np.stack([verts.min(0) - 0.05, verts.max(0) + 0.05]).astype(np.float32)
        K, R, T = synth.orbit_camera(0.4, verts.mean(0).astype(np.float64), 3.0, H, H)
        cases.append((cfg, H, H, K, R, T, wb))
    d = fixtures.renderer_inputs('tiny', smpl)['input_data']
    verts = d['vertices'][0]
    wb = np.stack([verts.min(0) - 0.05, verts.max(0) + 0.05]).astype(np.float32)
    c = verts.mean(0).astype(np.float64)
    K, R, T = synth.orbit_camera(1.1, c, 3.0, 48, 48, fill=4.5)
    cases.append(('wide', 48, 48, K, R, T, wb))
    K, R, T = synth.orbit_camera(-2.0, c, 2.5, 40, 56, fill=2.0)
    cases.append(('rect', 40, 56, K, R, T, wb))
    K = np.array([[64.0, 0, 16.0], [0, 64.0, 16.0], [0, 0, 1.0]]); R = np.eye(3); T = np.array([[0.0], [0.0], [3.0]]) - c.reshape(3, 1) * np.array([[1.0], [1.0], [1.0]])
    cases.append(('axis', 32, 32, K, R, T, wb))
    for name, H, W, K, R, T, wb in cases:
        ro, rd = get_rays(H, W, K, R, T)
        ray_o = ro.reshape(-1, 3).astype(np.float32)                     # RenderPeople_dataset.py:121-134
        ray_d = rd.reshape(-1, 3).astype(np.float32)
        near, far, mask_at_box = get_near_far(wb, ray_o, ray_d)
        near = near.astype(np.float32); far = far.astype(np.float32)
        near_all = np.zeros_like(ray_o[:, 0]); far_all = np.ones_like(ray_o[:, 0])
        near_all[mask_at_box] = near; far_all[mask_at_box] = far
        out.update({f'{name}_H': H, f'{name}_W': W, f'{name}_K': K, f'{name}_R': R, f'{name}_T': T, f'{name}_bounds': wb,
                    f'{name}_ray_o': ray_o, f'{name}_ray_d': ray_d, f'{name}_near': near_all, f'{name}_far': far_all,
                    f'{name}_mask_at_box': mask_at_box})
        print(f'rays {name}: {H}x{W}, at box {int(mask_at_box.sum())}/{H * W}, zero-patched {int((ray_d == np.float32(1e-8)).sum())}')
    out['cases'] = np.array([c[0] for c in cases])
    path = os.path.join(out_dir, 'rays.npz')
    np.savez_compressed(path, **out)
    print('rays ->', path)


if __name__ == '__main__':
    names = sys.argv[1:] or ['tiny', 'tiny_nv', 'cfg1']
    out_dir = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    if names == ['rays']:
        run_rays(out_dir); sys.exit(0)
    R, T = import_reference()
    if names == ['notrans']:                            # round 5: the reference built with use_trans = False, the reference-init tiny frame
        run('tiny_ri', R, T, out_dir, use_trans=False); sys.exit(0)
    if names and names[0] == 'branches':                # round 6: the feature-branch switches, the reference-init tiny frame (e.g. `branches 110 101 011 100`)
        for b in names[1:] or ['110', '101', '011', '100']:
            run('tiny_ri', R, T, out_dir, branches=tuple(ch == '1' for ch in b))
        sys.exit(0)
    if names == ['grad_notrans']:                       # round 6: the reference's gradients WITHOUT its transformer (use_trans = False), the tiny novel-view frame
        run_grad('tiny_nv', R, T, out_dir, use_trans=False); sys.exit(0)
    if names == ['refinit']:
        run_refinit_check(R, T, out_dir); sys.exit(0)
    if all(n.endswith('_ri') for n in names):           # only the reference-init variants: leave the other files alone
        run_refinit_check(R, T, out_dir)
        for n in names:
            run(n, R, T, out_dir)
        sys.exit(0)
    run_rays(out_dir)
    run_small_units(R, T, out_dir)
    run_glue(R, T, out_dir)
    for n in names:
        run(n, R, T, out_dir)
    for n in names:
        if n in ('tiny', 'tiny_nv'):
            run_grad(n, R, T, out_dir)
