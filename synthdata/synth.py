"""TEST INFRASTRUCTURE (oracle/) -- never imported by the product package `sherf_amd`.

Synthetic, seeded stand-ins for the assets the reference needs but which are absent offline
(licence-gated SMPL_NEUTRAL.pkl, datasets, cameras):

  * `make_synth_smpl`     -- a 6890-vertex articulated tube body with the pickle keys the reference
                             reads (sherf/training/volumetric_rendering/renderer.py:65-74).
  * `smpl_forward`        -- numpy SMPL forward, restating sherf/smpl/smpl_numpy.py:46-98 (scipy
                             Rotation instead of cv2.Rodrigues).
  * `get_rays`/`get_near_far`/`pack_near_far`
                          -- restating sherf/training/RenderPeople_dataset.py:14-27, 68-101, 121-134.
  * `big_pose_params`     -- RenderPeople_dataset.py:222-235.
  * `make_input_data`     -- the `input_data` dict after default collate (RenderPeople_dataset.py:362-391).

Everything is generated from explicit seeds so golden fixtures need to store outputs only.
"""
import numpy as np
import scipy.sparse
from scipy.spatial.transform import Rotation

V = 6890
NJ = 24

# rest-pose joint positions (metres; y up, x to the subject's left), roughly SMPL-neutral
_JOINTS = np.array([
    [0.00, -0.24, 0.03], [0.06, -0.33, 0.02], [-0.06, -0.33, 0.02], [0.00, -0.13, 0.00],
    [0.10, -0.71, 0.02], [-0.10, -0.71, 0.02], [0.00, 0.01, 0.02], [0.09, -1.11, -0.02],
    [-0.09, -1.11, -0.02], [0.00, 0.06, 0.03], [0.12, -1.17, 0.10], [-0.12, -1.17, 0.10],
    [0.00, 0.27, -0.01], [0.08, 0.18, 0.00], [-0.08, 0.18, 0.00], [0.00, 0.34, 0.03],
    [0.17, 0.22, -0.02], [-0.17, 0.22, -0.02], [0.43, 0.21, -0.04], [-0.43, 0.21, -0.04],
    [0.68, 0.22, -0.04], [-0.68, 0.22, -0.04], [0.77, 0.21, -0.05], [-0.77, 0.21, -0.05],
], dtype=np.float64)
_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21])

# body parts: (start joint, end joint or explicit end offset, radius at start, radius at end)
_PARTS = [
    (0, 1, 0.11, 0.09), (0, 2, 0.11, 0.09), (0, 3, 0.13, 0.13), (1, 4, 0.085, 0.06), (2, 5, 0.085, 0.06),
    (3, 6, 0.13, 0.135), (4, 7, 0.055, 0.04), (5, 8, 0.055, 0.04), (6, 9, 0.135, 0.14), (7, 10, 0.04, 0.035),
    (8, 11, 0.04, 0.035), (9, 12, 0.14, 0.06), (9, 13, 0.07, 0.06), (9, 14, 0.07, 0.06), (12, 15, 0.055, 0.06),
    (13, 16, 0.06, 0.055), (14, 17, 0.06, 0.055), (16, 18, 0.05, 0.04), (17, 19, 0.05, 0.04),
    (18, 20, 0.04, 0.03), (19, 21, 0.04, 0.03), (20, 22, 0.03, 0.03), (21, 23, 0.03, 0.03),
    # leaf extensions (so every joint starts at least one part -> J_regressor rings exist)
    (15, np.array([0.0, 0.17, 0.01]), 0.095, 0.085), (22, np.array([0.09, 0.0, 0.0]), 0.03, 0.02),
    (23, np.array([-0.09, 0.0, 0.0]), 0.03, 0.02), (10, np.array([0.0, -0.02, 0.10]), 0.035, 0.03),
    (11, np.array([0.0, -0.02, 0.10]), 0.035, 0.03),
]


def _frame(axis):
    axis = axis / np.linalg.norm(axis)
    ref = np.array([0.0, 0.0, 1.0]) if abs(axis[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
    u = np.cross(axis, ref); u /= np.linalg.norm(u)
    w = np.cross(axis, u)
    return axis, u, w


def make_synth_smpl(seed=0):
    """Returns a dict with the keys/shapes of SMPL_NEUTRAL.pkl that renderer.py:65-74 converts:
    v_template[6890,3], shapedirs[6890,3,10], J_regressor (scipy sparse [24,6890]), kintree_table[2,24],
    f[F,3], weights[6890,24], posedirs[6890,3,207]."""
    rng = np.random.RandomState(seed)
    areas = []
    ends = []
    for (a, b, r0, r1) in _PARTS:
        pa = _JOINTS[a]
        pb = _JOINTS[b] if np.isscalar(b) or isinstance(b, (int, np.integer)) else pa + b
        ends.append((pa, pb))
        areas.append(np.linalg.norm(pb - pa) * (r0 + r1) * np.pi + 1e-3)
    areas = np.array(areas)
    nseg = 16
    rings = np.maximum(3, np.round(areas / areas.sum() * V / nseg).astype(int))
    # fix the total to exactly V by adjusting the torso parts' ring counts, remainder on a last ragged ring
    while rings.sum() * nseg > V:
        rings[np.argmax(rings)] -= 1
    deficit = V - rings.sum() * nseg
    verts, faces, weights = [], [], []
    jreg_rows = {}
    base = 0
    for pi, ((a, b, r0, r1), (pa, pb), nr) in enumerate(zip(_PARTS, ends, rings)):
        ns = nseg
        extra = 0
        if pi == 2 and deficit > 0:           # absorb the remainder as one ragged extra ring on the torso
            extra = deficit
        axis, u, w = _frame(pb - pa)
        L = np.linalg.norm(pb - pa)
        ts = np.linspace(0.0, 1.0, nr)
        ring_sizes = [ns] * nr + ([extra] if extra else [])
        ring_ts = list(ts) + ([0.5] if extra else [])
        ring_start = []
        for ri, (t, n_in_ring) in enumerate(zip(ring_ts, ring_sizes)):
            rad = (r0 * (1 - t) + r1 * t) * np.sqrt(max(1e-3, 1 - 0.75 * (2 * t - 1) ** 6))
            ang = 2 * np.pi * (np.arange(n_in_ring) + 0.5 * (ri % 2)) / n_in_ring
            ring = pa[None] + axis[None] * (t * L) + rad * (np.cos(ang)[:, None] * u[None] + np.sin(ang)[:, None] * w[None])
            ring_start.append(base + len(verts))
            verts.extend(list(ring))
            # skinning: driven by joint a; blend to child b near the far end and to parent(a) near the start
            for _ in range(n_in_ring):
                wrow = np.zeros(NJ)
                s_end = 0.5 * np.clip((t - 0.65) / 0.35, 0, 1) ** 2 if isinstance(b, (int, np.integer)) else 0.0
                par = _PARENTS[a]
                s_beg = 0.5 * np.clip((0.35 - t) / 0.35, 0, 1) ** 2 if par >= 0 else 0.0
                wrow[a] += 1.0 - s_end - s_beg
                if isinstance(b, (int, np.integer)):
                    wrow[b] += s_end
                if par >= 0:
                    wrow[par] += s_beg
                # a little mass on two more joints so every vertex has a generic 4-bone blend
                others = rng.choice(NJ, 2, replace=False)
                wrow[others] += rng.uniform(0.0, 0.03, 2)
                weights.append(wrow / wrow.sum())
        # quads between consecutive full rings (CCW seen from outside)
        for ri in range(nr - 1):
            s0, s1 = ring_start[ri], ring_start[ri + 1]
            for k in range(ns):
                k1 = (k + 1) % ns
                faces.append([s0 + k, s0 + k1, s1 + k])
                faces.append([s0 + k1, s1 + k1, s1 + k])
        if extra:  # fan the ragged ring to the middle ring so its vertices belong to some face
            s0, sm = ring_start[-1], ring_start[nr // 2]
            for k in range(extra - 1):
                faces.append([s0 + k, s0 + k + 1, sm + (k % ns)])
        if a not in jreg_rows:                # first ring of the first part starting at joint a -> centre == joint
            jreg_rows[a] = np.arange(ring_start[0], ring_start[0] + ns)
    verts = np.asarray(verts, dtype=np.float64)
    assert verts.shape[0] == V, verts.shape
    faces = np.asarray(faces, dtype=np.int64)
    # orient faces outward w.r.t. their part axis (checked per face against the centroid direction)
    weights = np.asarray(weights, dtype=np.float64)
    rows, cols, vals = [], [], []
    for j in range(NJ):
        idx = jreg_rows[j]
        rows += [j] * len(idx); cols += list(idx); vals += [1.0 / len(idx)] * len(idx)
    J_regressor = scipy.sparse.csc_matrix((vals, (rows, cols)), shape=(NJ, V))
    # smooth-ish shape blend shapes, small random pose blend shapes
    p = verts
    shapedirs = np.zeros((V, 3, 10))
    for b in range(10):
        A = rng.normal(0, 0.02, (3, 3))
        ph = rng.uniform(0, 2 * np.pi, 3)
        shapedirs[:, :, b] = p @ A.T + 0.006 * np.sin(4.0 * p[:, [1, 2, 0]] + ph[None])
    posedirs = rng.normal(0, 0.0015, (V, 3, 207))
    kintree = np.stack([_PARENTS.copy(), np.arange(NJ)]).astype(np.int64)
    kintree[0, 0] = 4294967295  # as in the real pickle (uint32 -1); only kintree[0,1:] is ever used
    # make the triangles wind outward: flip those whose normal points to the part axis
    tri = verts[faces]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    cen = tri.mean(1)
    # nearest bone axis point for the centroid
    best = np.full(len(cen), np.inf); outward = np.zeros_like(cen)
    for (pa, pb) in ends:
        ab = pb - pa
        t = np.clip(((cen - pa) @ ab) / (ab @ ab), 0, 1)
        q = pa[None] + t[:, None] * ab[None]
        d = np.linalg.norm(cen - q, axis=1)
        m = d < best
        best[m] = d[m]; outward[m] = (cen - q)[m]
    flip = (n * outward).sum(1) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    return {
        'v_template': verts, 'shapedirs': shapedirs, 'J_regressor': J_regressor, 'kintree_table': kintree,
        'f': faces.astype(np.uint32), 'weights': weights, 'posedirs': posedirs,
    }


def smpl_forward(smpl, pose, beta):
    """numpy SMPL forward (restates sherf/smpl/smpl_numpy.py:46-98). pose (72,), beta (10,) -> verts[6890,3], joints[24,3]."""
    pose = np.asarray(pose, dtype=np.float64).reshape(-1)
    beta = np.asarray(beta, dtype=np.float64).reshape(-1)
    v_shaped = smpl['shapedirs'].reshape(-1, 10).dot(beta).reshape(V, 3) + smpl['v_template']
    Jr = np.asarray(smpl['J_regressor'].todense())
    J = Jr.dot(v_shaped)
    R = Rotation.from_rotvec(pose.reshape(NJ, 3)).as_matrix().astype(np.float32)   # smpl_numpy.py:61-65 (float32 cast)
    lrot = (R[1:] - np.eye(3, dtype=np.float32)[None]).reshape(-1, 1)
    v_posed = v_shaped + smpl['posedirs'].reshape(-1, 207).dot(lrot).reshape(V, 3)
    parent = _PARENTS
    Jrel = J.copy(); Jrel[1:] = J[1:] - J[parent[1:]]
    G_ = np.zeros((NJ, 4, 4)); G_[:, :3, :3] = R; G_[:, :3, 3] = Jrel; G_[:, 3, 3] = 1
    G = [G_[0].copy()]
    for i in range(1, NJ):
        G.append(G[parent[i]].dot(G_[i]))
    G = np.stack(G)
    joints = G[:, :3, 3].copy()
    rest = np.concatenate([J, np.zeros((NJ, 1))], -1)[:, :, None]
    G = G - np.matmul(G, np.concatenate([np.zeros((NJ, 4, 3)), rest], -1))
    T = smpl['weights'].dot(G.reshape(NJ, -1)).reshape(V, 4, 4)
    vh = np.concatenate([v_posed, np.ones((V, 1))], -1)
    v = np.matmul(T, vh[:, :, None])[:, :3, 0]
    return v, joints


def big_pose_params():
    """RenderPeople_dataset.py:222-235 (note R is np.ones((3,3)) there; it is never used numerically)."""
    p = {'R': np.ones((3, 3), np.float32), 'Th': np.zeros((1, 3), np.float32),
         'shapes': np.zeros((1, 10), np.float32), 'poses': np.zeros((1, 72), np.float32)}
    p['poses'][0, 5] = 45 / 180 * np.pi
    p['poses'][0, 8] = -45 / 180 * np.pi
    p['poses'][0, 23] = -30 / 180 * np.pi
    p['poses'][0, 26] = 30 / 180 * np.pi
    return p


def get_rays(H, W, K, R, T):
    """RenderPeople_dataset.py:14-27. Un-normalised directions, pixel corners (no +0.5)."""
    o = -(R.T @ T).ravel()
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    pix = np.stack([i, j, np.ones_like(i)], 2) @ np.linalg.inv(K).T
    d = (pix - T.ravel()) @ R - o[None, None]
    return np.broadcast_to(o, d.shape), d


def get_near_far(bounds, ray_o, ray_d):
    """RenderPeople_dataset.py:68-101 (slab test; rays with exactly two face hits are 'at box').  Like the reference, patches
    zero components of the caller's `ray_d` IN PLACE (:71) -- the array the dataset then returns (:121-134)."""
    bounds = bounds + np.array([-0.01, 0.01])[:, None]
    ray_d[ray_d == 0.0] = 1e-8
    nom = bounds[None] - ray_o[:, None]
    d_int = (nom / ray_d[:, None]).reshape(-1, 6)
    p_int = d_int[..., None] * ray_d[:, None] + ray_o[:, None]
    mn, mx = bounds[0], bounds[1]
    eps = 1e-6
    inside = np.ones(p_int.shape[:2], bool)
    for c in range(3):
        inside &= (p_int[..., c] >= mn[c] - eps) & (p_int[..., c] <= mx[c] + eps)
    at_box = inside.sum(-1) == 2
    p_iv = p_int[at_box][inside[at_box]].reshape(-1, 2, 3)
    ro, rd = ray_o[at_box], ray_d[at_box]
    nr = np.linalg.norm(rd, axis=1)
    d0 = np.linalg.norm(p_iv[:, 0] - ro, axis=1) / nr
    d1 = np.linalg.norm(p_iv[:, 1] - ro, axis=1) / nr
    return np.minimum(d0, d1), np.maximum(d0, d1), at_box


def pack_near_far(bounds, ray_o, ray_d):
    """RenderPeople_dataset.py:121-134: rays that miss the box get (near,far)=(0,1)."""
    ray_o = ray_o.reshape(-1, 3).astype(np.float32)
    ray_d = ray_d.reshape(-1, 3).astype(np.float32)
    near, far, at_box = get_near_far(bounds, ray_o, ray_d)
    near_all = np.zeros_like(ray_o[:, 0]); far_all = np.ones_like(ray_o[:, 0])
    near_all[at_box] = near.astype(np.float32); far_all[at_box] = far.astype(np.float32)
    return ray_o, ray_d, near_all, far_all, at_box


def orbit_camera(theta, center, dist, H, W, fill=2.2):
    """OpenCV-convention extrinsics (x_cam = R x_world + T) on a horizontal orbit, y-up world."""
    C = center + dist * np.array([np.sin(theta), 0.0, np.cos(theta)])
    fwd = center - C; fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right)   # image x
    down = np.cross(fwd, right)                                 # image y
    R = np.stack([right, down, fwd]).astype(np.float64)
    T = (-R @ C).reshape(3, 1)
    f = H * dist / fill
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    return K, R, T


def make_input_data(smpl, H=32, W=32, seed=1, theta_tgt=0.4, theta_obs=-0.3, novel_pose=True, dist=3.0, fill=2.2):
    """Collated (B=1) `input_data` dict as numpy arrays (keys of RenderPeople_dataset.py:362-391)."""
    rng = np.random.RandomState(seed)
    pose_t = rng.normal(0, 0.15, (1, 72)).astype(np.float32)
    pose_o = rng.normal(0, 0.15, (1, 72)).astype(np.float32) if novel_pose else pose_t.copy()
    beta = rng.normal(0, 0.5, (1, 10)).astype(np.float32)
    Th = np.array([[0.02, 0.01, 0.03]], np.float32)
    Rg = np.eye(3, dtype=np.float32)

    def prep(pose):
        xyz, _ = smpl_forward(smpl, pose.reshape(-1), beta.reshape(-1))
        xyz = (xyz @ Rg.T + Th).astype(np.float32)
        wb = np.stack([xyz.min(0) - 0.05, xyz.max(0) + 0.05])
        return wb, xyz, {'poses': pose.copy(), 'shapes': beta.copy(), 'R': Rg.copy(), 'Th': Th.copy()}

    wb, vertices, params = prep(pose_t)
    _, obs_vertices, obs_params = prep(pose_o)
    tp = big_pose_params()
    t_vertices, _ = smpl_forward(smpl, tp['poses'].reshape(-1), tp['shapes'].reshape(-1))
    t_vertices = t_vertices.astype(np.float32)
    mn, mx = t_vertices.min(0) - 0.05, t_vertices.max(0) + 0.05
    mn[2] -= 0.1; mx[2] += 0.1
    t_world_bounds = np.stack([mn, mx]).astype(np.float32)

    center = vertices.mean(0).astype(np.float64)
    K, R, T = orbit_camera(theta_tgt, center, dist, H, W, fill=fill)
    ro, rd = get_rays(H, W, K, R, T)
    ray_o, ray_d, near, far, at_box = pack_near_far(wb, ro, rd)
    oK, oR, oT = orbit_camera(theta_obs, obs_vertices.mean(0).astype(np.float64), dist, H, W)
    obs_img = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    d = {
        't_params': {k: v[None] for k, v in tp.items()},
        't_vertices': t_vertices[None], 't_world_bounds': t_world_bounds[None],
        'params': {k: v[None] for k, v in params.items()}, 'vertices': vertices[None],
        'ray_o_all': ray_o[None, None], 'ray_d_all': ray_d[None, None],
        'near_all': near[None, None, :, None], 'far_all': far[None, None, :, None],
        'mask_at_box_all': at_box[None, None],
        'obs_params': {k: v[None] for k, v in obs_params.items()}, 'obs_vertices': obs_vertices[None],
        'obs_img_all': obs_img[None, None], 'obs_K_all': oK[None, None].astype(np.float32),
        'obs_R_all': oR[None, None].astype(np.float32), 'obs_T_all': oT[None, None].astype(np.float32),
    }
    # default collate turns (1,72)->(1,1,72), (1,10)->(1,1,10), R (3,3)->(1,3,3), Th (1,3)->(1,1,3): already so.
    return d
