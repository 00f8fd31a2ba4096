"""The cell lists and the two exact nearest-vertex searches (csrc/sample.hip) called directly on the CPU build of their source, against
brute force in the kernels' own arithmetic (d^2 = ((dx*dx) + (dy*dy)) + (dz*dz) in fp32 without fma, lowest id on ties):

  * sherf_sample_mask_nn: shell mask, vertex ids and the ray-major compaction order -- with S = 80 (two 64-sample chunks, the
    second ragged), rays that miss everything, rays along a dense cluster (more than 32 points in a ball);
  * sherf_warp_geom: the nearest T-vertex through nn_search_batched -- queries whose ball stays inside 3 x 3 rows of cells (the
    batched path) and queries whose same-index vertex is far away (the plain loops behind it).
"""
import ctypes

import numpy as np
import pytest
import torch

from sherf_amd import _lib
from tests.hipcpu import build_cpu

MAX_CELLS = 64 * 64 * 64


@pytest.fixture(scope='module')
def lib(tmp_path_factory):
    path = build_cpu.build('sherf_hipcpu_nn', ['sample.hip'], str(tmp_path_factory.mktemp('hipcpu_nn')),
                           extra_src='char g_sherf_err[256] = ""; int g_sherf_debug = 0;\n')
    lib = ctypes.CDLL(path)
    protos = _lib.parse_header()
    for name in ('sherf_build_cells', 'sherf_build_cells2', 'sherf_build_near_lists', 'sherf_sample_mask_nn', 'sherf_warp_geom'):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = protos[name][0], [a[0] for a in protos[name][1]]
    return lib


def _P(t):
    return ctypes.c_void_p(t.data_ptr())


def _d2(q, p):
    """[nq,3] x [np,3] -> [nq,np] squared distances exactly as dist2_exact evaluates them (fp32, no fma)."""
    d = (q[:, None, :].astype(np.float32) - p[None, :, :].astype(np.float32)).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)


def _points(rs, n):
    """A thin noisy shell (like a body surface) plus one dense cluster of 60 points inside a 2 cm cube."""
    u = rs.normal(size=(n - 60, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    shell = u * np.array([0.25, 0.45, 0.18]) + rs.normal(scale=0.004, size=(n - 60, 3))
    cluster = np.array([0.25, 0.0, 0.0]) + rs.uniform(-0.01, 0.01, size=(60, 3))
    return np.concatenate([shell, cluster]).astype(np.float32)


def test_shell_mask_and_vertex_ids_against_brute_force(lib):
    rs = np.random.RandomState(3)
    n = 900
    verts = torch.from_numpy(_points(rs, n))
    Rg, Th = torch.eye(3).contiguous(), torch.zeros(3)
    hdr = torch.zeros(2, 12); cell_start = torch.zeros(2, MAX_CELLS + 1, dtype=torch.int32); cell_pts = torch.zeros(2, n, 4)
    scratch = torch.zeros(2 * 5 * n, dtype=torch.int32); near_mask = torch.zeros(32768, dtype=torch.int32)
    assert lib.sherf_build_cells2(_P(verts), _P(Rg), _P(Th), _P(verts), n, 0.05, _P(hdr), _P(cell_start), _P(cell_pts), _P(scratch), _P(near_mask), None) == 0
    # the cell list is a permutation of the points, ids intact
    ids = cell_pts[0, :, 3].contiguous().view(torch.int32)
    assert sorted(ids.tolist()) == list(range(n)) and torch.equal(cell_pts[0, :, :3], verts[ids.long()])
    assert int(cell_start[0].max()) == n
    # rays: through the shell, through the cluster, past everything -- for both search paths (two passes over a dense candidate list /
    # one wave per ray: sherf_set_debug bit 9) and one, two and three 64-sample chunks per ray
    # the near lists (round 4): per sub-cell of the near mask exactly the vertices whose distance to the sub-cell's box is below the
    # radius (+ the mask's margin) -- checked against brute force; bit set <=> list not empty
    NSUB = 524288
    near_hdr = torch.zeros(2 * NSUB + 2, dtype=torch.int32); near_list = torch.zeros(125 * n + 3 * NSUB, dtype=torch.int16)
    mask2 = torch.full((32768,), -1, dtype=torch.int32)                # the mask as the list builder writes it == sherf_build_cells2's
    assert lib.sherf_build_near_lists(_P(hdr), _P(cell_pts), n, 0.05, _P(near_hdr), _P(near_list), near_list.numel(), _P(mask2), None) == 0
    g = hdr[0]
    o, cell = g[:3].numpy().astype(np.float64), float(g[3])
    nx, ny, nz, sub = [int(v) for v in g[5:9].view(torch.int32)]
    cs = cell / sub
    hq = near_hdr[:2 * NSUB].view(NSUB, 2).numpy()
    nsub = nx * ny * nz * sub ** 3
    assert not hq[nsub:].any() and (hq[:nsub, 0] % 4 == 0).all()
    bits = np.unpackbits(near_mask.numpy().view(np.uint8), bitorder='little')[:nsub].astype(bool)
    assert np.array_equal(bits, hq[:nsub, 1] > 0)
    assert torch.equal(mask2[:(nsub + 31) // 32], near_mask[:(nsub + 31) // 32])
    pts = cell_pts[0, :, :3].numpy().astype(np.float64)
    total = 0
    for q in rs.choice(np.flatnonzero(bits), 300, replace=False).tolist() + rs.randint(0, nsub, 100).tolist():
        qx, qy, qz = q % (nx * sub), (q // (nx * sub)) % (ny * sub), q // (nx * sub * ny * sub)
        b0 = o + np.array([qx, qy, qz]) * cs
        e = np.maximum(np.maximum(b0 - pts, pts - (b0 + cs)), 0.0)
        dist = np.sqrt((e ** 2).sum(1))
        lst = near_list[hq[q, 0]: hq[q, 0] + hq[q, 1]].numpy().astype(np.int64) & 0xFFFF
        assert len(set(lst.tolist())) == len(lst)
        must, may = set(np.flatnonzero(dist < 0.05).tolist()), set(np.flatnonzero(dist < 0.05 + 2e-3 * cs).tolist())
        assert must <= set(lst.tolist()) <= may, q
        total += len(lst)
    assert total > 2000
    # lists never overlap: the allocation cursor == the sum of the padded counts
    assert int(near_hdr[2 * NSUB]) == int(((hq[:nsub, 1] + 3) // 4 * 4).sum())
    import ctypes as _ct
    dbg = _ct.c_int.in_dll(lib, 'g_sherf_debug')
    import os as _os
    # (xp: SHERF_EXPERIMENT -- 2048 = round 6's experimental compaction with one lane per ray + whole waves for the hit rays; default: one wave per ray)
    # (8192 = sixteen lanes per ray, four rays per wave: one, two and four mask words per ray; three words fall back to the default kernel;
    #  16384 = the list search one pipeline stage deeper, eight list entries per lane and step; 24576 = both; 49152 = the deeper search held to 80 registers)
    for S, flag, lists, xp in ((80, 0, True, '0'), (80, 0, True, '2048'), (80, 0, True, '24576'), (80, 0, False, '0'), (80, 512, False, '0'), (40, 0, True, '0'),
                               (40, 0, True, '2048'), (40, 0, True, '24576'), (150, 0, True, '0'), (150, 0, True, '2048'), (150, 0, True, '24576'), (150, 0, False, '0'),
                               (150, 512, False, '0'), (200, 0, True, '0'), (200, 0, True, '24576'), (80, 0, True, '49152')):
        dbg.value = flag
        _os.environ['SHERF_EXPERIMENT'] = xp
        R = 96
        o = rs.uniform(-0.1, 0.1, size=(R, 3)).astype(np.float32); o[:, 2] -= 1.0
        tgt = rs.uniform(-0.3, 0.3, size=(R, 3)).astype(np.float32)
        tgt[:24] = np.array([0.25, 0.0, 0.0]) + rs.uniform(-0.015, 0.015, size=(24, 3))          # aimed at the cluster
        tgt[24:32] += np.array([3.0, 3.0, 0.0])                                                   # misses
        d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
        near, far = np.full(R, 0.55, np.float32), np.full(R, 1.45, np.float32)
        ray_o, ray_d = torch.from_numpy(o), torch.from_numpy(d.astype(np.float32))
        t_near, t_far = torch.from_numpy(near), torch.from_numpy(far)
        cap = R * S
        counters = torch.zeros(4, dtype=torch.int32); ray_base = torch.zeros(R, dtype=torch.int32); ray_cnt = torch.zeros(R, dtype=torch.int32)
        cs_idx = torch.zeros(cap, dtype=torch.int32); cs_vid = torch.zeros(cap, dtype=torch.int32); cs_xs = torch.zeros(cap, 4)
        dense_vid = torch.zeros(R * S, dtype=torch.int32); ray_mask = torch.zeros(R * ((S + 63) // 64), dtype=torch.int64); scan_ws = torch.zeros(R + R // 1024 + 2, dtype=torch.int32)
        assert lib.sherf_sample_mask_nn(_P(ray_o), _P(ray_d), _P(t_near), _P(t_far), R, S, _P(Rg), _P(Th), _P(hdr), _P(cell_start), _P(cell_pts),
                                        _P(near_mask), cap, _P(counters), _P(ray_base), _P(ray_cnt), _P(cs_idx), _P(cs_vid), _P(cs_xs),
                                        _P(dense_vid), _P(ray_mask), _P(scan_ws), _P(near_hdr) if lists else None,
                                        _P(near_list) if lists else None, None) == 0
        # brute force in the same arithmetic: depths of math_utils.py:101-118, positions o + t d with separate roundings
        k = np.arange(S, dtype=np.float32)
        step = (k / np.float32(S - 1)).astype(np.float32)
        t = (near[:, None] + (step[None, :] * (far - near)[:, None]).astype(np.float32)).astype(np.float32)
        x = (o[:, None, :] + (t[..., None] * d[:, None, :].astype(np.float32)).astype(np.float32)).astype(np.float32).reshape(-1, 3)
        d2 = _d2(x, verts.numpy())
        best = d2.min(1)
        vid = np.array([np.flatnonzero(row == m)[0] for row, m in zip(d2, best)])
        valid = np.flatnonzero(best < np.float32(0.05 * 0.05))
        nv = int(counters[0])
        assert nv == valid.size and nv > 150, (S, flag, lists, nv, valid.size)
        assert np.array_equal(cs_idx[:nv].numpy(), valid) and np.array_equal(cs_vid[:nv].numpy(), vid[valid])
        assert np.array_equal(cs_xs[:nv, :3].numpy(), x[valid])
        assert np.array_equal(ray_cnt.numpy(), np.bincount(valid // S, minlength=R))
        assert np.array_equal(ray_base.numpy(), np.concatenate([[0], np.cumsum(np.bincount(valid // S, minlength=R))[:-1]]))
        # some ball really holds more points than one 32-point step of a group
        assert ((d2[valid] < np.float32(0.0025)).sum(1) > 32).any()


    dbg.value = 0
    _os.environ['SHERF_EXPERIMENT'] = '0'

    # ---- the warp's T-vertex search on the same grid (grid 1 of the pair): near and far same-index vertices ----
    nq = 400
    q = (verts.numpy()[rs.randint(0, n, nq)] + rs.normal(scale=0.01, size=(nq, 3))).astype(np.float32)
    same = _d2(q, verts.numpy()).argmin(1).copy()
    far_ones = rs.rand(nq) < 0.3
    same[far_ones] = rs.randint(0, n, int(far_ones.sum()))                 # a wrong (far) same-index vertex: ball of up to 1 m
    ident = torch.zeros(n, 12); ident[:, 0] = ident[:, 4] = ident[:, 8] = 1.0
    counters_w = torch.tensor([nq, 0, 0, 0], dtype=torch.int32)
    w_idx = 2 * torch.arange(nq, dtype=torch.int32); w_vid = torch.from_numpy(same.astype(np.int32)); w_xs = torch.zeros(nq, 4); w_xs[:, :3] = torch.from_numpy(q)
    w_rd = torch.zeros(nq, 3); w_rd[:, 2] = 1.0
    geom = torch.zeros(nq, 8); tvid = torch.zeros(nq, dtype=torch.int32)
    assert lib.sherf_warp_geom(_P(counters_w), _P(w_idx), _P(w_vid), _P(w_xs), _P(w_rd), 2, _P(Rg), _P(ident), _P(ident), _P(verts),
                               _P(hdr[1]), _P(cell_start[1]), _P(cell_pts[1]), nq, _P(geom), _P(tvid), None) == 0
    dq = _d2(q, verts.numpy())
    ref = np.array([np.flatnonzero(row == row.min())[0] for row in dq])
    assert np.array_equal(tvid.numpy(), ref)
    assert np.array_equal(geom[:, :3].numpy(), q)
    # both paths were taken: balls within 3 x 3 rows of 5 cm cells, and balls far wider than that
    rad = np.sqrt(dq[np.arange(nq), same])
    assert (rad < 0.04).sum() > 100 and (rad > 0.2).sum() > 50
