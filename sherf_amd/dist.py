"""Multi-GPU sharding of the render path: one process per GPU, `torch.distributed` over RCCL (backend "nccl" on
ROCm) / xGMI.  The reference has NO inference parallelism (eval runs on rank 0 only, training_loop.py:311-328);
rays are independent end to end, so the path shards with no data-path exchange and ONE collective per frame:
an all_gather of the rendered [rays, 5] (rgb, depth, acc) tiles -- 5.24 MB total at 512x512, latency bound.

Two partitions:
  * views  : a batch of target views of one subject, view v -> rank v % world  (BASELINE config 4)
  * rays   : one frame, interleaved tiles of `tile` consecutive rays round-robin over ranks, which balances the
             body's footprint (contiguous row blocks would not: the subject covers a fraction of the frame).
Per-frame shared state (planes, feature map, voxel tables, SMPL tables, weights, < 300 MB) is replicated.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_views(n_views, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_views, world_size))


def ray_tiles(n_rays, rank=None, world_size=None, tile=1024):
    """Indices (int64, ascending) of the rays this rank renders under the interleaved-tile partition."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    idx = torch.arange(n_rays)
    return idx[((idx // tile) % world_size) == rank]


def padded_shard_size(n_rays, world_size, tile=1024):
    return max(int((((torch.arange(n_rays) // tile) % world_size) == r).sum()) for r in range(world_size))


def gather_rays(local, n_rays, tile=1024):
    """local: [n_local, C] results for `ray_tiles(n_rays)` on every rank -> full [n_rays, C] on every rank (one all_gather)."""
    rank, w = world()
    if w == 1:
        return local
    m = padded_shard_size(n_rays, w, tile)
    buf = torch.zeros(m, local.shape[1], dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(w)]
    dist.all_gather(out, buf)
    full = torch.empty(n_rays, local.shape[1], dtype=local.dtype, device=local.device)
    for r in range(w):
        idx = ray_tiles(n_rays, r, w, tile).to(local.device)
        full[idx] = out[r][:idx.numel()]
    return full


def depth_range(near, far, reduce=False):
    """[lo, hi] (float32 tensor [2] on near's device) of ALL sample depths t_k = near + k/(S-1) (far - near) of a frame: what
    MipRayMarcher2 clamps the depth image with (ray_marcher.py:57: the min / max over the WHOLE tensor).  t is linear in k, so the
    extremes of a ray are its end points t_0 = near and t_{S-1} = near + (far - near) (evaluated exactly as the sampler does) -- in
    EITHER order: a ray with far < near contributes t_{S-1} to the minimum, as in the reference's torch.min / max over all depths and
    in the frame kernel's own depth_minmax.  Pass the FULL frame's near / far (every rank of a ray-tile sharded frame holds them: no
    collective needed) and hand the result to the renderer as rendering_options['depth_range'], so that a rank rendering a subset
    of the rays clamps with the frame's range, not its subset's.  reduce=True: for callers that only hold their own rays -- the
    per-rank ranges are MIN / MAX-reduced over the ranks (one 8-byte all_reduce)."""
    n, f = near.detach().float().reshape(-1), far.detach().float().reshape(-1)
    e = n + (f - n)
    v = torch.stack([-torch.minimum(n, e).min(), torch.maximum(n, e).max()])     # (one MAX reduction serves both ends)
    if reduce and world()[1] > 1:
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return torch.stack([-v[0], v[1]])


def render_ray_tiles(render, ray_origins, ray_directions, near, far, tile=1024):
    """One frame sharded over the ranks by interleaved ray tiles.  render(ray_o, ray_d, near, far, extra_options) ->
    (rgb [1,r,3], depth [1,r,1], acc [1,r,1]) renders a subset of rays (ImportanceRenderer.forward with the frame's other inputs
    bound); -> the full [R, 5] (rgb, depth, acc) frame on every rank, identical to the single-process frame."""
    R = ray_origins.shape[1]
    idx = ray_tiles(R, tile=tile).to(ray_origins.device)
    rng = depth_range(near, far)
    rgb, depth, acc = render(ray_origins[:, idx], ray_directions[:, idx], near[:, idx], far[:, idx], dict(depth_range=rng))
    return gather_rays(torch.cat([rgb[0], depth[0], acc[0]], 1), R, tile)


def gather_views(local_frames, n_views, frame_shape=None, dtype=torch.float32, device=None):
    """local_frames: {view index: [R, C]} rendered on this rank (view v lives on rank v % world) -> list of all frames.
    A rank that owns no view (n_views < world) must pass frame_shape / device so that it still takes part in the collective."""
    rank, w = world()
    if w == 1:
        return [local_frames[v] for v in range(n_views)]
    per = (n_views + w - 1) // w
    if local_frames:
        any_frame = next(iter(local_frames.values()))
        frame_shape, dtype, device = tuple(any_frame.shape), any_frame.dtype, any_frame.device
    elif frame_shape is None or device is None:
        raise ValueError('gather_views: this rank renders no view -- pass frame_shape and device (the other ranks are waiting in the all_gather)')
    buf = torch.zeros(per, *frame_shape, dtype=dtype, device=device)
    for i, v in enumerate(shard_views(n_views, rank, w)):
        buf[i] = local_frames[v]
    out = [torch.empty_like(buf) for _ in range(w)]
    dist.all_gather(out, buf)
    return [out[v % w][v // w] for v in range(n_views)]


def broadcast_params(modules, src=0, world_size=None):
    """The reference's start-up distribution of the weights (training_loop.py:231-236): every parameter and buffer of every module, one
    `broadcast(src=0)` each, so that all ranks start from rank 0's initialisation.  -> the number of tensors sent."""
    _, w = world()
    world_size = w if world_size is None else world_size
    n = 0
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            if t.numel() > 0 and world_size > 1:
                dist.broadcast(t.data if t.requires_grad else t, src=src)
                n += 1
    return n


def allreduce_flat_grads(params, world_size=None):
    """The reference's manual data-parallel gradient exchange (training_loop.py:374-383): flatten every existing
    grad, one all_reduce(SUM), divide by the number of GPUs, nan_to_num, scatter back."""
    _, w = world()
    world_size = w if world_size is None else world_size
    params = [p for p in params if p.grad is not None]
    if not params:
        return
    flat = torch.cat([p.grad.flatten() for p in params])
    if world_size > 1:
        dist.all_reduce(flat)
        flat /= world_size
    torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
    for p, g in zip(params, flat.split([p.numel() for p in params])):
        p.grad = g.reshape(p.shape)
