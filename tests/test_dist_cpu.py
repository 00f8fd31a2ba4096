"""world_size-2 gloo tests (CPU) of the multi-GPU sharding logic in sherf_amd/dist.py: the interleaved ray-tile
partition + all_gather reconstructs exactly the frame a single process renders; view sharding; flat-grad all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sherf_amd import dist as sd


def _fake_render(idx):
    """A pure function of the ray index stands in for the renderer (rays are independent end to end)."""
    i = idx.double()
    return torch.stack([torch.sin(i), torch.cos(i * 0.5), i % 7, i / 10.0, (i % 3 == 0).double()], 1).float()


def _worker(rank, world, port, n_rays, tile, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mine = sd.ray_tiles(n_rays, tile=tile)
        full = sd.gather_rays(_fake_render(mine), n_rays, tile=tile)
        ok_rays = torch.equal(full, _fake_render(torch.arange(n_rays)))
        views = sd.shard_views(5)
        frames = sd.gather_views({v: torch.full((3, 2), float(v)) for v in views}, 5)
        ok_views = all(torch.equal(frames[v], torch.full((3, 2), float(v))) for v in range(5))
        p = torch.nn.Parameter(torch.ones(4)); p.grad = torch.full((4,), float(rank + 1))
        q_ = torch.nn.Parameter(torch.ones(2)); q_.grad = torch.tensor([float('nan'), 1.0])
        sd.allreduce_flat_grads([p, q_])
        ok_grad = torch.allclose(p.grad, torch.full((4,), 1.5)) and float(q_.grad[0]) == 0.0
        q.put((rank, ok_rays, ok_views, ok_grad, len(mine)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_ray_and_view_sharding():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    n_rays, tile = 5000, 256                      # ragged: last tile partial, shards of unequal size
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rays, tile, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    assert all(r[1] and r[2] and r[3] for r in res), res
    assert sum(r[4] for r in res) == n_rays


def test_partition_covers_every_ray_once():
    for n, w, t in ((262144, 8, 1024), (1000, 3, 64), (7, 4, 2)):
        seen = torch.cat([sd.ray_tiles(n, r, w, t) for r in range(w)])
        assert torch.equal(torch.sort(seen)[0], torch.arange(n))
