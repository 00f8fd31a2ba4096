"""TEST INFRASTRUCTURE -- CPU restatement (torch-CPU / numpy, fp32) of SHERF's volumetric-rendering
hot path. It is the parity oracle for the HIP kernels in `sherf_amd/csrc`; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it. The product package never does.

Pinned against the reference: `tests/test_oracle_golden.py` checks every stage below against
`tests/golden/*.npz`, which `oracle/make_golden.py` produced by running the UNMODIFIED reference files
on the same seeded inputs in the build container. Two third-party pieces are NOT pinned ("parity
unpinned"): pytorch3d's K-NN tie-breaking (we: exact fp32 sum of squares,
lowest index wins) and spconv's behaviour on duplicate voxel coordinates (we: published indice-pair
algorithm => features of rows sharing a voxel sum at that voxel, non-winning rows output zero).

Every function cites the reference lines it restates (paths relative to /root/reference/sherf/).
All tensors are fp32 torch CPU tensors unless noted; B == 1 throughout (the reference forces it,
renderer.py:320-321,567).
"""
import math
import numpy as np
import torch

F32 = torch.float32
THRESH2 = 0.05 ** 2          # renderer.py:318
VOXEL = 0.005                # triplane.py:194, renderer.py:550
CHUNK = 700000               # renderer.py:355


# ----------------------------------------------------------------------------------------------
# rays and depths
# ----------------------------------------------------------------------------------------------
def ray_sampler(cam2world, intrinsics, res):
    """training/volumetric_rendering/ray_sampler.py:24-61 (EG3D convention, pixel centres, normalised dirs)."""
    N = cam2world.shape[0]
    fx, fy = intrinsics[:, 0, 0:1], intrinsics[:, 1, 1:2]
    cx, cy, sk = intrinsics[:, 0, 2:3], intrinsics[:, 1, 2:3], intrinsics[:, 0, 1:2]
    a = torch.arange(res, dtype=F32) * (1.0 / res) + (0.5 / res)
    # uv = stack(meshgrid(ij)).flip(0) -> first component varies fastest along the row
    x_cam = a.repeat(res)[None].expand(N, -1)
    y_cam = a.repeat_interleave(res)[None].expand(N, -1)
    z = torch.ones_like(x_cam)
    x_lift = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx * z
    y_lift = (y_cam - cy) / fy * z
    pts = torch.stack([x_lift, y_lift, z, torch.ones_like(z)], -1)
    world = torch.bmm(cam2world, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    loc = cam2world[:, :3, 3]
    dirs = torch.nn.functional.normalize(world - loc[:, None], dim=2)
    return loc[:, None].expand(-1, res * res, -1).contiguous(), dirs


def depths(near, far, S):
    """renderer.py:472-474 + math_utils.py:101-118: t_k = near + (k/(S-1)) * (far-near), no jitter. near,far [R]."""
    steps = torch.arange(S, dtype=F32) / (S - 1)
    return near[:, None] + steps[None, :] * (far - near)[:, None]          # [R,S]


# ----------------------------------------------------------------------------------------------
# SMPL helpers
# ----------------------------------------------------------------------------------------------
def smpl_tensors(smpl):
    """renderer.py:65-74.  (Rounded to fp32 first, as the reference holds them, also in the float64 truth mode: the tables, like the
    0.005 voxel size and the encodings' phase buffer below, are fp32 INPUTS of the function.)"""
    out = {}
    out['v_template'] = torch.tensor(np.asarray(smpl['v_template'], dtype=float), dtype=torch.float32).to(F32)
    out['shapedirs'] = torch.tensor(np.asarray(smpl['shapedirs'], dtype=float), dtype=torch.float32).to(F32)
    out['posedirs'] = torch.tensor(np.asarray(smpl['posedirs'], dtype=float), dtype=torch.float32).to(F32)
    out['weights'] = torch.tensor(np.asarray(smpl['weights'], dtype=float), dtype=torch.float32).to(F32)
    out['J_regressor'] = torch.tensor(smpl['J_regressor'].toarray().astype(float), dtype=torch.float32).to(F32)
    out['parents'] = torch.tensor(np.asarray(smpl['kintree_table']).astype(float), dtype=torch.long)[0]
    out['f'] = torch.tensor(np.asarray(smpl['f']).astype(float), dtype=torch.long)
    return out


def rodrigues(theta):
    """renderer.py:76-94 / 159-190: R = I + sin(a) K + (1-cos a) K^2, a = ||theta + 1e-8||. theta [n,3]."""
    a = torch.norm(theta + 1e-8, dim=1, keepdim=True)
    u = theta / a
    z = torch.zeros_like(a)
    K = torch.cat([z, -u[:, 2:3], u[:, 1:2], u[:, 2:3], z, -u[:, 0:1], -u[:, 1:2], u[:, 0:1], z], 1).view(-1, 3, 3)
    return torch.eye(3)[None] + torch.sin(a)[:, None] * K + (1 - torch.cos(a))[:, None] * torch.matmul(K, K)


def bone_transforms(st, poses, shapes):
    """renderer.py:129-157 + 96-126. poses [72], shapes [10] -> A [24,4,4] (rest pose removed)."""
    v_shaped = st['v_template'] + torch.sum(st['shapedirs'] * shapes.view(1, 1, 10), -1)
    Rm = rodrigues(poses.view(24, 3))
    J = st['J_regressor'] @ v_shaped
    par = st['parents']
    rel = J.clone()
    rel[1:] -= J[par[1:]]
    M = torch.zeros(24, 4, 4)
    M[:, :3, :3] = Rm; M[:, :3, 3] = rel; M[:, 3, 3] = 1
    chain = [M[0]]
    for i in range(1, 24):
        chain.append(chain[int(par[i])] @ M[i])
    G = torch.stack(chain)
    Jh = torch.cat([J, torch.zeros(24, 1)], 1)
    G = G.clone()
    G[:, :, 3] = G[:, :, 3] - torch.sum(G * Jh[:, None, :], dim=2)
    return G


def pose_offsets(st, poses):
    """renderer.py:578-584: posedirs @ vec(R_1..23 - I) -> [6890,3]."""
    Rm = rodrigues(poses.view(24, 3))
    feat = (Rm[1:] - torch.eye(3)[None]).reshape(1, -1)
    return torch.matmul(feat, st['posedirs'].view(6890 * 3, -1).t()).view(6890, 3)


def shape_offsets(st, shapes):
    """renderer.py:590-591."""
    return torch.matmul(st['shapedirs'], shapes.view(10, 1)).squeeze(-1)


NN_CHUNK = 2048     # query rows per distance block (bench.py's GPU baseline raises it)


def nearest_vertex(x, verts, chunk=None):
    """Exact K=1 nearest neighbour (the role of pytorch3d.ops.knn_points at renderer.py:315,564,627).
    d^2 = ((dx*dx)+(dy*dy))+(dz*dz) in fp32, ties -> lowest index. Returns (d2 [n], idx [n] int64)."""
    n = x.shape[0]
    chunk = chunk or NN_CHUNK
    d2 = torch.empty(n, dtype=F32); idx = torch.empty(n, dtype=torch.long)
    for s in range(0, n, chunk):
        q = x[s:s + chunk]
        dx = q[:, None, 0] - verts[None, :, 0]; dy = q[:, None, 1] - verts[None, :, 1]; dz = q[:, None, 2] - verts[None, :, 2]
        dd = (dx * dx + dy * dy) + dz * dz
        m, i = torch.min(dd, 1)
        d2[s:s + chunk] = m; idx[s:s + chunk] = i
    return d2, idx


def second_nearest_gap(x, verts, idx, chunk=None):
    """d^2 of the SECOND nearest vertex minus d^2 of the nearest one (`idx`, as returned by nearest_vertex) [n]: the margin by
    which the discontinuous selector `idx` is decided -- SURVEY.md section 7 hard part 1 (parity is only defined off the ties)."""
    n = x.shape[0]
    chunk = chunk or NN_CHUNK
    gap = torch.empty(n, dtype=F32)
    for s in range(0, n, chunk):
        q = x[s:s + chunk]
        dx = q[:, None, 0] - verts[None, :, 0]; dy = q[:, None, 1] - verts[None, :, 1]; dz = q[:, None, 2] - verts[None, :, 2]
        dd = (dx * dx + dy * dy) + dz * dz
        first = dd.gather(1, idx[s:s + chunk, None])
        dd.scatter_(1, idx[s:s + chunk, None], float('inf'))
        gap[s:s + chunk] = dd.min(1)[0] - first[:, 0]
    return gap


def target_to_canonical(st, params, t_params, verts_smpl_unused, x_s, v_s, vid):
    """renderer.py:558-621, the literal per-point chain. x_s, v_s [n,3] in the SMPL frame; vid [n] nearest posed vertex."""
    A = bone_transforms(st, params['poses'].view(-1), params['shapes'].view(-1))
    bw = st['weights'][vid]                                   # [n,24]
    M = torch.matmul(bw, A.reshape(24, 16)).view(-1, 4, 4)
    p = x_s - M[:, :3, 3]
    Rinv = torch.inverse(M[:, :3, :3])
    p = torch.matmul(Rinv, p[..., None]).squeeze(-1)
    u = torch.matmul(Rinv, v_s[..., None]).squeeze(-1) if v_s is not None else None
    p = p - pose_offsets(st, params['poses'].view(-1))[vid]
    p = p - shape_offsets(st, params['shapes'].view(-1))[vid]
    p = p + pose_offsets(st, t_params['poses'].view(-1))[vid]
    Ab = bone_transforms(st, t_params['poses'].view(-1), t_params['shapes'].view(-1))
    Mb = torch.matmul(bw, Ab.reshape(24, 16)).view(-1, 4, 4)
    x_c = torch.matmul(Mb[:, :3, :3], p[..., None]).squeeze(-1) + Mb[:, :3, 3]
    v_c = torch.matmul(Mb[:, :3, :3], u[..., None]).squeeze(-1) if u is not None else None
    return x_c, v_c


def canonical_to_obs_world(st, obs_params, t_params, t_vertices, x_c, k=None):
    """renderer.py:623-684 -> (world_src_pts [n,3], nearest T-vertex id [n]).  k: the nearest T-vertex ids, when they are GIVEN
    (fp64 truth mode: the discrete decisions of the fp32 run are reused, see truth64_from_fixture)."""
    if k is None:
        _, k = nearest_vertex(x_c, t_vertices)
    bw = st['weights'][k]
    bw = bw + 0.2 * 0
    bw = bw / torch.sum(bw, -1, keepdim=True)
    Ab = bone_transforms(st, t_params['poses'].view(-1), t_params['shapes'].view(-1))
    Mb = torch.matmul(bw, Ab.reshape(24, 16)).view(-1, 4, 4)
    p = x_c - Mb[:, :3, 3]
    p = torch.matmul(torch.inverse(Mb[:, :3, :3]), p[..., None]).squeeze(-1)
    p = p - pose_offsets(st, t_params['poses'].view(-1))[k]
    p = p + shape_offsets(st, obs_params['shapes'].view(-1))[k]
    p = p + pose_offsets(st, obs_params['poses'].view(-1))[k]
    Ao = bone_transforms(st, obs_params['poses'].view(-1), obs_params['shapes'].view(-1))
    Mo = torch.matmul(bw, Ao.reshape(24, 16)).view(-1, 4, 4)
    x_o = torch.matmul(Mo[:, :3, :3], p[..., None]).squeeze(-1) + Mo[:, :3, 3]
    Rg = obs_params['R'].view(3, 3); Th = obs_params['Th'].view(1, 3)
    return torch.matmul(x_o, torch.inverse(Rg)) + Th, k


# ----------------------------------------------------------------------------------------------
# feature taps (hand-written interpolation; grid_sample semantics of ATen)
# ----------------------------------------------------------------------------------------------
def positional_encoding(x, F):
    """renderer.py:875-916: [x, sin(f0 x), sin(f0 x + pi/2), sin(f1 x), ...] each a d_in-vector."""
    freqs = 2.0 ** torch.linspace(0.0, F - 1, F)
    fr = torch.repeat_interleave(freqs, 2).view(1, -1, 1)
    ph = torch.zeros(2 * F); ph[1::2] = float(np.float32(math.pi * 0.5))      # the fp32 `_phases` buffer (renderer.py:893)
    e = torch.sin(torch.addcmul(ph.view(1, -1, 1), x.unsqueeze(1).repeat(1, 2 * F, 1), fr)).view(x.shape[0], -1)
    return torch.cat([x, e], -1)


def _bilinear(img, px, py):
    """img [C,H,W]; px,py pixel-space floats [n]; zeros padding -> [n,C]."""
    C, H, W = img.shape
    x0 = torch.floor(px); y0 = torch.floor(py)
    fx = px - x0; fy = py - y0
    out = torch.zeros(px.shape[0], C)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = (x0 + dx).long(); yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            v = img[:, yi.clamp(0, H - 1), xi.clamp(0, W - 1)].t()
            out += (wx * wy * ok.float())[:, None] * v
    return out


def grid_sample_2d(img, gx, gy, align_corners):
    """F.grid_sample(bilinear, zeros) on one image [C,H,W] at normalised coords (gx -> width)."""
    C, H, W = img.shape
    if align_corners:
        px = (gx + 1) / 2 * (W - 1); py = (gy + 1) / 2 * (H - 1)
    else:
        px = ((gx + 1) * W - 1) / 2; py = ((gy + 1) * H - 1) / 2
    return _bilinear(img, px, py)


def project_uv(x_w, Rc, Tc, K):
    """renderer.py:686-704 without faces: uv = (K (R x + T))[:2] / (z + 1e-5)."""
    c = torch.matmul(Rc.view(3, 3), x_w[..., None]) + Tc.view(3, 1)
    h = torch.matmul(K.view(3, 3), c)[..., 0]
    return h[:, :2] / (h[:, 2:] + 1e-5)


def pixel_aligned_features(uv, obs_feat, obs_img):
    """renderer.py:330-340: both taps with align_corners=True; rgb -> PE5 -> first 32 of 33."""
    H, W = obs_img.shape[-2:]
    g = 2.0 * uv / torch.tensor([W, H], dtype=F32) - 1.0
    f = grid_sample_2d(obs_feat, g[:, 0], g[:, 1], True)
    rgb = grid_sample_2d(obs_img, g[:, 0], g[:, 1], True)
    return torch.cat([f, positional_encoding(rgb, 5)[:, :32]], -1), rgb


def triplane_features(planes, x_c, bounds):
    """renderer.py:234-243,218-232,192-216: planes [3,32,P,P]; normalise by t_world_bounds; (x,y),(x,z),(z,y);
    bilinear, zeros, align_corners=False -> [3,n,32]."""
    n = 2 * (x_c - bounds[0:1]) / (bounds[1:2] - bounds[0:1]) - 1
    sel = ((0, 1), (0, 2), (2, 1))
    return torch.stack([grid_sample_2d(planes[p], n[:, a], n[:, b], False) for p, (a, b) in enumerate(sel)])


def compute_normal(vertices, faces):
    """renderer.py:50-63. NOTE the reference writes `norm[:, faces[:, c]] += n` with advanced indexing: for a vertex
    listed by several faces in column c only ONE face's normal lands (index assignment, not accumulation; on CPU the
    last face wins). That behaviour is restated here (deterministically: highest face index). [V,3], [F,3]."""
    tri = vertices[faces]
    n = torch.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], dim=-1)
    ln = torch.sqrt((n ** 2).sum(-1)).clamp_min(1e-8)
    n = n / ln[:, None]
    norm = torch.zeros_like(vertices)
    nf = faces.shape[0]
    for c in range(3):
        last = torch.full((vertices.shape[0],), -1, dtype=torch.long).scatter_reduce_(0, faces[:, c], torch.arange(nf), reduce='amax')
        has = last >= 0
        norm[has] = norm[has] + n[last[has]]
    ln = torch.sqrt((norm ** 2).sum(-1)).clamp_min(1e-8)
    return norm / ln[:, None]


def vertex_features(state, st, obs_vertices, Rc, Tc, K, obs_feat, obs_img, prefix='generator.'):
    """triplane.py:111-126: per-vertex 32-d features for the sparse voxel encoder. Projects the observation-pose
    vertices into the observation image, taps feature map + image (align_corners=True), PE5(rgb)[:32], conv1d 96->32,
    zeroes back-facing vertices (normal . view >= 0, renderer.py:691-695). -> (feat [6890,32], front mask [6890])."""
    cam = torch.matmul(obs_vertices, Rc.view(3, 3).t()) + Tc.view(1, 3)
    ncam = torch.matmul(compute_normal(obs_vertices, st['f']), Rc.view(3, 3).t())
    front = (ncam * cam).sum(-1) < 0
    h = torch.matmul(cam, K.view(3, 3).t())
    uv = h[:, :2] / (h[:, 2:] + 1e-5)
    f2d, _ = pixel_aligned_features(uv, obs_feat, obs_img)
    W = state[prefix + 'conv1d_projection.weight'][:, :, 0]
    f = f2d @ W.t() + state[prefix + 'conv1d_projection.bias']
    return f * front[:, None].float(), front


# ----------------------------------------------------------------------------------------------
# sparse voxel encoder (unique-voxel formulation, see module docstring)
# ----------------------------------------------------------------------------------------------
def prepare_sp_input(t_vertices, xyz):
    """triplane.py:174-217. t_vertices [6890,3] (canonical), xyz [6890,3] canonicalised obs verts."""
    mn = t_vertices.min(0)[0] - 0.05
    mx = t_vertices.max(0)[0] + 0.05
    vs = torch.tensor([VOXEL] * 3, dtype=torch.float32).to(F32)
    dhw = xyz[:, [2, 1, 0]]
    coord = torch.round((dhw - mn[[2, 1, 0]][None]) / vs).to(torch.int32)
    out_sh = torch.ceil((mx[[2, 1, 0]] - mn[[2, 1, 0]]) / vs).to(torch.int32)
    out_sh = (out_sh | 31) + 1
    coord = torch.cat([torch.zeros(coord.shape[0], 1, dtype=torch.int32), coord], 1)
    return dict(coord=coord, out_sh=[int(v) for v in out_sh], bounds=torch.stack([mn, mx]))


_ENC_LAYERS = (  # renderer.py:728-740, executed part for num_layers=4 (renderer.py:756-785)
    ('conv0', 'subm', 2), ('down0', 'down', 1), ('conv1', 'subm', 2), ('TAP', None, 0),
    ('down1', 'down', 1), ('conv2', 'subm', 3), ('TAP', None, 0),
    ('down2', 'down', 1), ('conv3', 'subm', 3), ('TAP', None, 0),
)


def _lin(z, y, x, sh):
    return (z * sh[1] + y) * sh[2] + x


def _bn_relu(raw, mult, n_rows, state, prefix, training):
    """nn.BatchNorm1d(eps=1e-3) + ReLU over the ROW set (renderer.py:807-808): `raw` are the winning rows,
    the other (n_rows - len(raw)) rows are zeros. Returns per-voxel summed activations."""
    g, b = state[prefix + '.weight'], state[prefix + '.bias']
    if training:
        mean = raw.sum(0) / n_rows
        var = (((raw - mean) ** 2).sum(0) + (n_rows - raw.shape[0]) * mean ** 2) / n_rows
    else:
        mean, var = state[prefix + '.running_mean'], state[prefix + '.running_var']
    inv = 1.0 / torch.sqrt(var + 1e-3)
    act = torch.relu((raw - mean) * inv * g + b)
    v0 = torch.relu((0 - mean) * inv * g + b)
    return act + (mult - 1).to(F32)[:, None] * v0[None]


def sparse_encoder(state, feat, coord, out_sh, training=True, prefix='renderer.encoder_3d.'):
    """renderer.py:744-797 on the sparse set. Returns [(keys sorted int64, feats [U,C], shape)] for the three
    tapped levels (after conv1, conv2, conv3)."""
    sh = [int(v) for v in out_sh]
    c = coord.long()
    keys = _lin(c[:, 1], c[:, 2], c[:, 3], sh)
    uk, inv = torch.unique(keys, sorted=True, return_inverse=True)
    mult = torch.bincount(inv, minlength=uk.numel())
    g = torch.zeros(uk.numel(), feat.shape[1]).index_add_(0, inv, feat)
    n_rows = feat.shape[0]
    taps = []
    for name, kind, nconv in _ENC_LAYERS:
        if name == 'TAP':
            taps.append((uk.clone(), g.clone(), list(sh)))
            continue
        z = uk // (sh[1] * sh[2]); y = (uk // sh[2]) % sh[1]; x = uk % sh[2]
        if kind == 'subm':
            for ci in range(nconv):
                W = state[f'{prefix}{name}.{3 * ci}.weight']            # [out,3,3,3,in]
                raw = torch.zeros(uk.numel(), W.shape[0])
                for kz in range(3):
                    for ky in range(3):
                        for kx in range(3):
                            qz, qy, qx = z + kz - 1, y + ky - 1, x + kx - 1
                            ok = (qz >= 0) & (qz < sh[0]) & (qy >= 0) & (qy < sh[1]) & (qx >= 0) & (qx < sh[2])
                            qk = _lin(qz, qy, qx, sh)
                            pos = torch.searchsorted(uk, qk).clamp(max=uk.numel() - 1)
                            ok &= uk[pos] == qk
                            o = torch.nonzero(ok)[:, 0]
                            if o.numel():
                                raw[o] += g[pos[o]] @ W[:, kz, ky, kx, :].t()
                g = _bn_relu(raw, mult, n_rows, state, f'{prefix}{name}.{3 * ci + 1}', training)
        else:
            W = state[f'{prefix}{name}.0.weight']
            osh = [(d - 1) // 2 + 1 for d in sh]
            ok_all, key_all, src_all, k_all = [], [], [], []
            for kz in range(3):
                for ky in range(3):
                    for kx in range(3):
                        nz, ny, nx = z + 1 - kz, y + 1 - ky, x + 1 - kx
                        ok = (nz % 2 == 0) & (ny % 2 == 0) & (nx % 2 == 0)
                        oz, oy, ox = nz // 2, ny // 2, nx // 2
                        ok &= (oz >= 0) & (oz < osh[0]) & (oy >= 0) & (oy < osh[1]) & (ox >= 0) & (ox < osh[2])
                        o = torch.nonzero(ok)[:, 0]
                        key_all.append(_lin(oz, oy, ox, osh)[o]); src_all.append(o)
                        k_all.append(torch.full_like(o, (kz * 3 + ky) * 3 + kx))
            key_all = torch.cat(key_all); src_all = torch.cat(src_all); k_all = torch.cat(k_all)
            nuk, ninv = torch.unique(key_all, sorted=True, return_inverse=True)
            raw = torch.zeros(nuk.numel(), W.shape[0])
            Wf = W.reshape(W.shape[0], 27, W.shape[4])
            for kk in range(27):
                m = k_all == kk
                if m.any():
                    raw.index_add_(0, ninv[m], g[src_all[m]] @ Wf[:, kk, :].t())
            uk, sh = nuk, osh
            mult = torch.ones(uk.numel(), dtype=torch.long)
            n_rows = uk.numel()
            g = _bn_relu(raw, mult, n_rows, state, f'{prefix}{name}.1', training)
    return taps


def voxel_grid_coords(x_c, bounds, out_sh):
    """renderer.py:544-556 -> normalised (x,y,z) grid coords in [-1,1] of the level-0 volume."""
    dhw = x_c[:, [2, 1, 0]] - bounds[0][[2, 1, 0]][None]
    dhw = dhw / torch.tensor([VOXEL] * 3, dtype=torch.float32).to(F32)
    dhw = dhw / torch.tensor(out_sh, dtype=F32) * 2 - 1
    return dhw[:, [2, 1, 0]]


def trilinear_sparse(keys, feats, shape, g):
    """F.grid_sample(net.dense(), g, zeros, align_corners=True) (renderer.py:762-764) on the sparse set. g [n,3] (x,y,z)."""
    D, H, W = shape
    px = (g[:, 0] + 1) / 2 * (W - 1); py = (g[:, 1] + 1) / 2 * (H - 1); pz = (g[:, 2] + 1) / 2 * (D - 1)
    x0, y0, z0 = torch.floor(px), torch.floor(py), torch.floor(pz)
    fx, fy, fz = px - x0, py - y0, pz - z0
    out = torch.zeros(g.shape[0], feats.shape[1])
    for dz, wz in ((0, 1 - fz), (1, fz)):
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi, yi, zi = (x0 + dx).long(), (y0 + dy).long(), (z0 + dz).long()
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H) & (zi >= 0) & (zi < D)
                k = _lin(zi, yi, xi, shape)
                pos = torch.searchsorted(keys, k).clamp(max=keys.numel() - 1)
                ok &= keys[pos] == k
                out += (wx * wy * wz * ok.float())[:, None] * feats[pos]
    return out


# ----------------------------------------------------------------------------------------------
# fusion + decoder
# ----------------------------------------------------------------------------------------------
def fuse_tokens(state, tri, f2d, f3d, prefix='renderer.'):
    """renderer.py:423-424: slot s input = [triplane_s | f2d[32s:32s+32] | f3d[32s:32s+32]] -> conv1d 96->32. -> [n,3,32]"""
    W = state[prefix + 'conv1d_reprojection.weight'][:, :, 0]; b = state[prefix + 'conv1d_reprojection.bias']
    n = f2d.shape[0]
    comb = torch.cat([tri.permute(1, 0, 2), f2d.view(n, 3, 32), f3d.view(n, 3, 32)], -1)    # [n,3,96]
    return comb @ W.t() + b


def _layer_norm(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-5) * w + b


def transformer(state, tok, prefix='renderer.transformer.layers.0.'):
    """renderer.py:949-993: pre-norm attention (3 heads x 16, scale 16^-0.5) + pre-norm FF (GELU exact). tok [n,3,32].
    A state without the transformer's parameters is a renderer built with use_trans = False: renderer.py:427 is skipped, the tokens pass through."""
    p = prefix
    if p + '0.fn.fn.to_qkv.weight' not in state:
        return tok
    h = _layer_norm(tok, state[p + '0.fn.norm.weight'], state[p + '0.fn.norm.bias'])
    qkv = h @ state[p + '0.fn.fn.to_qkv.weight'].t()                      # [n,3,144]
    n = tok.shape[0]
    q, k, v = [t.view(n, 3, 3, 16).permute(0, 2, 1, 3) for t in qkv.chunk(3, -1)]   # [n,head,tok,16]
    att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (16 ** -0.5), -1)
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(n, 3, 48)
    y = o @ state[p + '0.fn.fn.to_out.0.weight'].t() + state[p + '0.fn.fn.to_out.0.bias'] + tok
    h = _layer_norm(y, state[p + '1.fn.norm.weight'], state[p + '1.fn.norm.bias'])
    h = h @ state[p + '1.fn.fn.net.0.weight'].t() + state[p + '1.fn.fn.net.0.bias']
    h = 0.5 * h * (1 + torch.erf(h / math.sqrt(2.0)))
    return h @ state[p + '1.fn.fn.net.3.weight'].t() + state[p + '1.fn.fn.net.3.bias'] + y


def nerf_decoder(state, pe_x, z, pe_v, prefix='decoder.'):
    """triplane.py:285-316. pe_x [n,39], z [n,3,32] fused tokens, pe_v [n,27] -> rgb [n,3], sigma [n]."""
    x0 = torch.cat([pe_x, z[:, 0]], -1)
    h = x0
    for i in range(8):
        h = torch.relu(h @ state[f'{prefix}pts_linears.{i}.weight'].t() + state[f'{prefix}pts_linears.{i}.bias'])
        if i == 4:
            h = torch.cat([x0, h], -1)
    sigma = (h @ state[prefix + 'alpha_linear.weight'].t() + state[prefix + 'alpha_linear.bias'])[:, 0]
    f = h @ state[prefix + 'feature_linear.weight'].t() + state[prefix + 'feature_linear.bias']
    g = torch.relu(torch.cat([f, pe_v, z[:, 1]], -1) @ state[prefix + 'views_linear.weight'].t() + state[prefix + 'views_linear.bias'])
    rgb = torch.sigmoid(g @ state[prefix + 'rgb_linear.weight'].t() + state[prefix + 'rgb_linear.bias']) * (1 + 2 * 0.001) - 0.001
    return rgb, sigma


# ----------------------------------------------------------------------------------------------
# compositing
# ----------------------------------------------------------------------------------------------
def composite(colors, sigma, t, rays_d, white_back=False):
    """ray_marcher.py:25-64 (clamp_mode='relu'). colors [R,S,3], sigma [R,S], t [R,S], rays_d [R,3]
    -> rgb [R,3] in [-1,1], depth [R], weights [R,S]."""
    delta = torch.cat([t[:, 1:] - t[:, :-1], torch.full_like(t[:, :1], 1e10)], 1)
    delta = delta * torch.norm(rays_d, dim=-1)[:, None]
    alpha = 1 - torch.exp(-(torch.relu(sigma) * delta))
    shifted = torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], 1)
    w = alpha * torch.cumprod(shifted, 1)[:, :-1]
    rgb = torch.sum(w[..., None] * colors, 1)
    acc = w.sum(1)
    depth = torch.sum(w * t, 1) / acc
    depth = torch.nan_to_num(depth, float('inf'))
    depth = torch.clamp(depth, torch.min(t), torch.max(t))
    if white_back:
        rgb = rgb + 1 - acc[:, None]
    return rgb * 2 - 1, depth, w


# ----------------------------------------------------------------------------------------------
# the whole path
# ----------------------------------------------------------------------------------------------
def render(state, st, planes, obs_img, obs_feat, vertex_feat, sp_input, ray_o, ray_d, near, far,
           input_data, options, training=True, keep=True, decisions=None):
    """ImportanceRenderer.forward, renderer.py:286-398 (all feature branches on, NeRF decoder, transformer).
    planes [3,32,P,P]; obs_img [3,H,W]; obs_feat [64,H/2,W/2]; vertex_feat [6890,32]; ray_* [R,3]; near/far [R].
    Returns dict with rgb [R,3], depth [R], acc [R] and (keep=True) every intermediate.
    decisions: dict(valid, vert_id, t_vert_id) = the three discrete selections of ANOTHER run (truth64_from_fixture); the K-NN
    searches are then skipped and the continuous part of the path is evaluated on exactly those branches."""
    S = int(options['depth_resolution'])
    R_ = ray_o.shape[0]
    P = input_data['params']; OP = input_data['obs_params']; TP = input_data['t_params']
    Rg = P['R'].view(3, 3); Th = P['Th'].view(1, 3)
    t = depths(near, far, S)                                                   # [R,S]
    x = (ray_o[:, None, :] + t[..., None] * ray_d[:, None, :]).reshape(-1, 3)  # renderer.py:304
    v = ray_d[:, None, :].expand(-1, S, -1).reshape(-1, 3)
    x_s = torch.matmul(x - Th, Rg); v_s = torch.matmul(v, Rg)                  # :309-310
    verts_s = torch.matmul(input_data['vertices'].view(-1, 3) - Th, Rg)        # :313-314
    if decisions is None:
        d2, vid_all = nearest_vertex(x_s, verts_s)                             # :315
        mask = d2 < THRESH2                                                    # :316-319
        valid = torch.nonzero(mask)[:, 0]
    else:
        valid = decisions['valid'].long()
        mask = torch.zeros(R_ * S, dtype=torch.bool); mask[valid] = True
        vid_all = torch.zeros(R_ * S, dtype=torch.long); vid_all[valid] = decisions['vert_id'].long()
        d2 = None
        keep = False
    out = dict(t=t, mask=mask, valid=valid)
    nv = valid.numel()
    sig_full = torch.full((R_ * S,), -80.0); col_full = torch.zeros(R_ * S, 3)  # :364-368
    if nv > 0:
        xs, vs, vid = x_s[valid], v_s[valid], vid_all[valid]
        x_c, v_c = target_to_canonical(st, P, TP, verts_s, xs, vs, vid)        # :323
        x_w, tvid = canonical_to_obs_world(st, OP, TP, input_data['t_vertices'].view(-1, 3), x_c,
                                           None if decisions is None else decisions['t_vert_id'].long())   # :328
        uv = project_uv(x_w, input_data['obs_R_all'].view(3, 3), input_data['obs_T_all'].view(3, 1), input_data['obs_K_all'].view(3, 3))
        f2d, tap_rgb = pixel_aligned_features(uv, obs_feat, obs_img)           # :330-340
        taps = sparse_encoder(state, vertex_feat, sp_input['coord'], sp_input['out_sh'], training)
        g = voxel_grid_coords(x_c, sp_input['bounds'], sp_input['out_sh'])     # :346
        f3d_raw = torch.cat([trilinear_sparse(k, f, s, g) for (k, f, s) in taps], -1)     # [nv,192]
        Wp = state['renderer.conv1d_projection.weight'][:, :, 0]
        f3d = f3d_raw @ Wp.t() + state['renderer.conv1d_projection.bias']      # :350
        rgbs, sigs, toks_in, toks_out = [], [], [], []
        bounds = input_data['t_world_bounds'].view(2, 3)
        for s0 in range(0, nv, CHUNK):                                         # :355-362
            sl = slice(s0, s0 + CHUNK)
            tri = triplane_features(planes, x_c[sl], bounds)                   # :402
            tok = fuse_tokens(state, tri, f2d[sl], f3d[sl])                    # :423-424
            z = transformer(state, tok)                                        # :427
            rgb, sig = nerf_decoder(state, positional_encoding(x_c[sl], 6), z, positional_encoding(v_c[sl], 4))   # :432
            rgbs.append(rgb); sigs.append(sig); toks_in.append(tok); toks_out.append(z)
        rgb_s, sig_s = torch.cat(rgbs), torch.cat(sigs)
        col_full[valid] = rgb_s; sig_full[valid] = sig_s
        out.update(vert_id=vid, t_vert_id=tvid)               # the discrete selections: what a float64 run on THESE branches re-uses
        if decisions is not None:
            out.update(sample_rgb=rgb_s, sample_sigma=sig_s, x_c=x_c)
        elif keep or options.get('margins'):
            # decision margins of the three discontinuous selectors on the path (shell threshold, nearest posed vertex, nearest
            # T-pose vertex): what tests / bench use to tell an implementation's legitimate boundary flips from errors
            out.update(d2_all=d2, vert_gap=second_nearest_gap(xs, verts_s, vid),
                       t_vert_gap=second_nearest_gap(x_c, input_data['t_vertices'].view(-1, 3), tvid))
            if not keep:
                out.update(vert_id=vid, t_vert_id=tvid, sample_rgb=rgb_s, sample_sigma=sig_s)
        if keep:
            out.update(vert_id=vid, vert_d2=d2[valid], x_s=xs, v_s=vs, x_c=x_c, v_c=v_c, x_w=x_w, t_vert_id=tvid, uv=uv,
                       f2d=f2d, tap_rgb=tap_rgb, grid=g, f3d_raw=f3d_raw, f3d=f3d, tokens_in=torch.cat(toks_in),
                       tokens_out=torch.cat(toks_out), sample_rgb=rgb_s, sample_sigma=sig_s, taps=taps,
                       _tok_chunks=toks_in, _z_chunks=toks_out)
    rgb, depth, w = composite(col_full.view(R_, S, 3), sig_full.view(R_, S), t, ray_d, bool(options.get('white_back', False)))
    out.update(rgb=rgb, depth=depth, acc=w.sum(1), weights=w)
    return out


def render_from_fixture(fx, state, training=True, keep=True, device=None):
    """Convenience: run `render` on a dict produced by oracle.fixtures.renderer_inputs (numpy) + a state dict.
    device: None = CPU (the oracle proper).  bench.py --torch-gpu-baseline passes the GPU: the same stock ATen ops then run there
    (`state` must already live on it) -- the "reference through stock PyTorch-ROCm" denominator of SURVEY.md section 8(d)."""
    if device is not None:
        with torch.device(device):          # factory calls below (torch.zeros, torch.tensor, ...) then allocate on `device`
            return _render_from_fixture(fx, state, training, keep, torch.device(device))
    return _render_from_fixture(fx, state, training, keep, None)


class _working_dtype:
    """Everything above takes its floating-point type from the module global `F32` and torch's default dtype; inside this context
    both are float64."""
    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        global F32
        self.prev = (F32, torch.get_default_dtype())
        F32 = self.dt
        torch.set_default_dtype(self.dt)

    def __exit__(self, *a):
        global F32
        F32 = self.prev[0]
        torch.set_default_dtype(self.prev[1])


def truth64_from_fixture(fx, state, ref32, training=True, device=None):
    """The fp64 TRUTH of the continuous part of the path: the same restatement evaluated in float64 from the same fp32 inputs and
    weights (converted exactly), on the discrete branches of the fp32 run `ref32` (its valid set, nearest posed vertex and nearest
    T-vertex per sample, its voxel coordinates: `render_from_fixture(...)` output, keep=False + options['margins'] or keep=True).
    Both fp32 implementations -- this oracle run in fp32, pinned to the unmodified reference, and the HIP path -- are then measured
    against it sample by sample (oracle/parity.py: truth_protocol): the independent arbiter VERDICT round 2 asked for in place of the
    first-order conditioning probe.  -> dict(sample_rgb [nv,3], sample_sigma [nv], rgb, acc, depth, x_c), float64."""
    dec = dict(valid=ref32['valid'], vert_id=ref32['vert_id'], t_vert_id=ref32['t_vert_id'])
    spi = ref32['sp_input']
    with _working_dtype(torch.float64):
        def run(dev):
            dd = lambda a: (a.double() if a.is_floating_point() else a) if torch.is_tensor(a) else a
            st64 = {k: dd(v if dev is None else v.to(dev)) for k, v in state.items()}
            sp64 = dict(coord=spi['coord'] if dev is None else spi['coord'].to(dev), out_sh=spi['out_sh'],
                        bounds=dd(spi['bounds'] if dev is None else spi['bounds'].to(dev)))
            if dev is not None:
                dec_d = {k: v.to(dev) for k, v in dec.items()}
            else:
                dec_d = dec
            return _render_from_fixture(fx, st64, training, False, dev, decisions=dec_d, sp_input=sp64, cast=torch.float64)
        if device is not None:
            with torch.device(device):
                return run(torch.device(device))
        return run(None)


def _render_from_fixture(fx, state, training, keep, device, decisions=None, sp_input=None, cast=None):
    def to(a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        if cast is not None and torch.is_tensor(a) and a.is_floating_point():
            a = a.to(cast)
        return a.to(device) if device is not None and torch.is_tensor(a) else a
    d = {k: ({kk: to(vv) for kk, vv in v.items()} if isinstance(v, dict) else to(v)) for k, v in fx['input_data'].items()}
    st = smpl_tensors(fx['smpl'])
    OP = d['obs_params']
    obs_s = torch.matmul(d['obs_vertices'].view(-1, 3) - OP['Th'].view(1, 3), OP['R'].view(3, 3))
    if sp_input is None:
        _, ovid = nearest_vertex(obs_s, obs_s)
        obs_can, _ = target_to_canonical(st, OP, d['t_params'], obs_s, obs_s, None, ovid)       # triplane.py:129-132
        sp_input = prepare_sp_input(d['t_vertices'].view(-1, 3), obs_can)
    else:
        obs_can = None
    res = render(state, st, to(fx['planes'])[0], d['obs_img_all'][0, 0], to(fx['obs_feat'])[0], to(fx['vertex_feat']), sp_input,
                 d['ray_o_all'][0, 0], d['ray_d_all'][0, 0], d['near_all'][0, 0, :, 0], d['far_all'][0, 0, :, 0], d, fx['options'],
                 training=training, keep=keep, decisions=decisions)
    res['sp_input'] = sp_input
    res['obs_vertex_canonical'] = obs_can
    return res


GRAD_SAMPLE = 64


def grad_fingerprint(g):
    """[sum, sum|.|, L2] (float64) + GRAD_SAMPLE strided entries of a gradient tensor -- the compact form the golden
    gradients are stored in (oracle/make_golden.py: grad_fingerprint)."""
    g = g.detach().reshape(-1).double()
    n = g.numel()
    idx = torch.linspace(0, n - 1, min(GRAD_SAMPLE, n)).round().long()
    return np.concatenate([np.array([float(g.sum()), float(g.abs().sum()), float(g.norm())]), g[idx].numpy()])


def stub_loss(rgb, acc):
    """BASELINE config 5's stub training loss: MSE(rgb, T_rgb) + MSE(acc, T_acc) with seeded targets
    (RandomState(11): rgb target first, then acc, as in make_golden.loss_targets).  rgb [R,3], acc [R]."""
    rs = np.random.RandomState(11)
    t_rgb = torch.from_numpy(rs.uniform(-1, 1, (1,) + tuple(rgb.shape)).astype(np.float32))[0].to(rgb.device)
    t_acc = torch.from_numpy(rs.uniform(0, 1, (1,) + tuple(acc.shape) + (1,)).astype(np.float32))[0, :, 0].to(acc.device)
    return ((rgb - t_rgb) ** 2).mean() + ((acc - t_acc) ** 2).mean()


STAGE_KEYS = ('sample_rgb', 'sample_sigma', 'tokens_out', 'tokens_in', 'f2d', 'f3d', 'f3d_raw')


def gradients_from_fixture(fx, state, stages=False, device=None, info=None):
    """info: an optional dict that receives 'sp_input' (the voxelisation the graph used).  device: None = CPU (the oracle proper); a GPU device runs the SAME autograd graph through stock PyTorch-ROCm ops there (the
    full-size gradient check of tests/test_gpu_backward.py: 512 x 512 x 64 takes seconds instead of the CPU's hours)."""
    if device is not None:
        with torch.device(device):
            return _gradients_from_fixture(fx, {k: v.to(device) for k, v in state.items()}, stages, torch.device(device), info)
    return _gradients_from_fixture(fx, state, stages, None, info)


def _gradients_from_fixture(fx, state, stages, device, info=None):
    """Backward of the path by autograd through this restatement (training-mode BatchNorm): returns (loss, {name: grad})
    for every renderer / decoder parameter that receives one and for the three feature inputs
    ('input.planes', 'input.obs_feat', 'input.vertex_feat').  The oracle for the HIP backward kernels (BASELINE config 5).

    stages=True additionally returns the gradient AT EVERY STAGE BOUNDARY of the forward ('stage.<key>' for STAGE_KEYS:
    per-sample rgb / sigma = what the compositing backward emits, transformer output / input tokens = what the MLP backward
    emits, f2d / f3d / f3d_raw = what the gather backward scatters; 'stage.level<i>' = gradient of the i-th tapped voxel
    level's activations), so each backward kernel can be checked on its own."""
    st = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in state.items()}
    fx = dict(fx)
    leaves = {}
    mv = lambda t: t if device is None else t.to(device)
    for key, name in (('planes', 'input.planes'), ('obs_feat', 'input.obs_feat'), ('vertex_feat', 'input.vertex_feat')):
        leaves[name] = mv(torch.from_numpy(np.ascontiguousarray(fx[key]))).requires_grad_(True)
    to = lambda a: mv(torch.from_numpy(np.ascontiguousarray(a))) if isinstance(a, np.ndarray) else (mv(a) if torch.is_tensor(a) else a)
    d = {k: ({kk: to(vv) for kk, vv in v.items()} if isinstance(v, dict) else to(v)) for k, v in fx['input_data'].items()}
    smpl = smpl_tensors(fx['smpl'])
    OP = d['obs_params']
    with torch.no_grad():
        obs_s = torch.matmul(d['obs_vertices'].view(-1, 3) - OP['Th'].view(1, 3), OP['R'].view(3, 3))
        _, ovid = nearest_vertex(obs_s, obs_s)
        obs_can, _ = target_to_canonical(smpl, OP, d['t_params'], obs_s, obs_s, None, ovid)
    sp_input = prepare_sp_input(d['t_vertices'].view(-1, 3), obs_can)
    if info is not None:
        info['sp_input'] = sp_input
    res = render(st, smpl, leaves['input.planes'][0], d['obs_img_all'][0, 0], leaves['input.obs_feat'][0], leaves['input.vertex_feat'],
                 sp_input, d['ray_o_all'][0, 0], d['ray_d_all'][0, 0], d['near_all'][0, 0, :, 0], d['far_all'][0, 0, :, 0], d,
                 fx['options'], training=True, keep=stages)
    if info is not None and 'vert_id' in res:
        info['decisions'] = dict(valid=res['valid'].detach(), vert_id=res['vert_id'].detach(), t_vert_id=res['t_vert_id'].detach())
    watched = {}
    if stages:
        for k in STAGE_KEYS:
            if k in res and res[k].requires_grad:
                res[k].retain_grad(); watched['stage.' + k] = res[k]
        for i, (_, feats, _) in enumerate(res.get('taps', [])):
            if feats.requires_grad:
                feats.retain_grad(); watched[f'stage.level{i}'] = feats
        chunked = {'stage.tokens_in': res.get('_tok_chunks', []), 'stage.tokens_out': res.get('_z_chunks', [])}
        for ts in chunked.values():                      # the graph runs through the per-chunk tensors, not their concatenation
            for t in ts:
                t.retain_grad()
    loss = stub_loss(res['rgb'], res['acc'])
    loss.backward()
    grads = {n: t.grad for n, t in leaves.items()}
    grads.update({n: t.grad for n, t in st.items() if t.is_floating_point() and t.grad is not None})
    grads.update({n: t.grad for n, t in watched.items() if t.grad is not None})
    if stages:
        grads.update({n: torch.cat([t.grad for t in ts]) for n, ts in chunked.items() if ts})
    return float(loss.detach()), grads


def gradients_truth64_from_fixture(fx, state, info, device=None):
    """The float64 TRUTH of the gradients (round 5; the backward's counterpart of truth64_from_fixture): autograd through the same restatement
    evaluated in float64 from the same fp32 inputs and weights, on the discrete branches (valid set, nearest posed vertex, nearest T-vertex,
    voxel coordinates) of the fp32 run whose `info` dict (gradients_from_fixture(..., info=info)) is passed in.  Both fp32 gradient sets -- the
    oracle's own fp32 autograd, pinned to the unmodified reference's backward, and the HIP backward -- are then measured against it per
    parameter: where the fp32 REFERENCE itself sits far from the truth (gradients that are sums of cancelling terms behind a BatchNorm, ReLU
    kinks) a looser agreement is conditioning, not a kernel error.  -> (loss, {name: float64 grad})."""
    dec, spi = info['decisions'], info['sp_input']
    with _working_dtype(torch.float64):
        def run(dev):
            mv = (lambda t: t) if dev is None else (lambda t: t.to(dev))
            dd = lambda a: (a.double() if a.is_floating_point() else a) if torch.is_tensor(a) else a
            st = {k: (dd(mv(v)).clone().requires_grad_(True) if v.is_floating_point() else mv(v)) for k, v in state.items()}
            leaves = {name: dd(mv(torch.from_numpy(np.ascontiguousarray(fx[key])))).requires_grad_(True)
                      for key, name in (('planes', 'input.planes'), ('obs_feat', 'input.obs_feat'), ('vertex_feat', 'input.vertex_feat'))}
            to = lambda a: dd(mv(torch.from_numpy(np.ascontiguousarray(a)))) if isinstance(a, np.ndarray) else (dd(mv(a)) if torch.is_tensor(a) else a)
            d = {k: ({kk: to(vv) for kk, vv in v.items()} if isinstance(v, dict) else to(v)) for k, v in fx['input_data'].items()}
            smpl = smpl_tensors(fx['smpl'])
            sp64 = dict(coord=mv(spi['coord']), out_sh=spi['out_sh'], bounds=dd(mv(spi['bounds'])))
            res = render(st, smpl, leaves['input.planes'][0], d['obs_img_all'][0, 0], leaves['input.obs_feat'][0], leaves['input.vertex_feat'], sp64,
                         d['ray_o_all'][0, 0], d['ray_d_all'][0, 0], d['near_all'][0, 0, :, 0], d['far_all'][0, 0, :, 0], d, fx['options'],
                         training=True, keep=False, decisions={k: mv(v) for k, v in dec.items()})
            loss = stub_loss(res['rgb'], res['acc'])
            loss.backward()
            grads = {n: t.grad for n, t in leaves.items()}
            grads.update({n: t.grad for n, t in st.items() if t.is_floating_point() and t.grad is not None})
            return float(loss.detach()), grads
        if device is not None:
            with torch.device(device):
                return run(torch.device(device))
        return run(None)


def psnr(a, b):
    """test_loop.py:36-37 on images mapped to [0,1]."""
    mse = torch.mean(((a / 2 + 0.5) - (b / 2 + 0.5)) ** 2)
    return float(-10.0 * torch.log(mse) / math.log(10.0))
