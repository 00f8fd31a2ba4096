"""Training-step timing for BASELINE config 5 (forward render + backward through the HIP kernels + the reference's flat-grad
all-reduce, training_loop.py:374-383 + Adam step) -- companion of bench.py (whose metric is the forward render), same launch
contract:

    python bench_train.py --gpus N --steps K --warmup W          (N > 1: launches its N ranks itself, or runs under torchrun; one rank per GPU)

The backward pipeline (sherf_amd/backward.py) is checked against the unmodified reference's gradients by
tests/test_gpu_backward.py (`pytest -m gpu`).  Rank 0 prints ONE JSON line; `roofline` is the step's dominant kernel
(sherf_gather_tokens_bwd: the scatter of d_tokens into the tri-planes, the feature map and the voxel rows with fp32 atomics), timed
with events on the launch stream inside the timed steps; `phases_ms` splits a step into forward / backward / all-reduce + Adam.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)       # two steps under rocprofv3 --pmc: nothing printed
    ap.add_argument('--no-pmc', action='store_true', help='skip the two rocprofv3 --pmc passes that count the tap scatter\'s HBM bytes')
    a = ap.parse_args()
    if a.pmc_child:
        a.steps, a.warmup = 2, 1
    import bench
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:        # no launcher in front: start the N ranks here (as bench.py does; sherf/train.py:98-103)
        sys.exit(bench.launch_ranks(a.gpus, script=__file__))
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); lrank = int(os.environ.get('LOCAL_RANK', 0))
    bench.check_world(a.gpus, world, 'bench_train.py')
    host_build = bool(os.environ.get('SHERF_HIPCPU_LIB'))            # TEST INFRASTRUCTURE (tests/test_dist_cpu.py): the script's plumbing on CPU tensors over gloo
    if host_build:
        bench._use_host_build()
        a.no_pmc = True
    dev = bench._device(lrank)
    if world > 1:
        backend = os.environ.get('SHERF_DIST_BACKEND', 'nccl')        # 'nccl' = RCCL; 'gloo' only for the CPU dry run in tests/
        torch.distributed.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))

    class _Ev:                                                        # a point on the stream: a HIP event; wall clock on the host build (launches are synchronous there)
        def __init__(self, enable_timing=True):
            self.e = None if host_build else torch.cuda.Event(enable_timing=True)
            self.t = 0.0

        def record(self):
            if self.e is not None:
                self.e.record()
            else:
                self.t = time.perf_counter()

        def elapsed_time(self, other):
            return self.e.elapsed_time(other.e) if self.e is not None else 1e3 * (other.t - self.t)
    from synthdata import fixtures, synth                    # seeded synthetic inputs (not the oracle)
    from sherf_amd import dist as sdist
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder, TriPlaneGenerator
    from sherf_amd.voxel import SparseConvTensor
    smpl = synth.make_synth_smpl(0)
    fx, d, to = bench.make_inputs(a.config, 0.4 + rank * 2 * np.pi / max(world, 1), dev)
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=smpl)
    dec = NeRFDecoder(32)
    variant = fixtures.variant_of(a.config)                  # ('_ri' configurations: the reference-init network, as in bench.py)
    fixtures.load_seeded_state(rend, 'renderer.', variant); fixtures.load_seeded_state(dec, 'decoder.', variant)
    rend.to(dev).train(); dec.to(dev).train()
    rend.enable_autograd = True
    gen = TriPlaneGenerator.__new__(TriPlaneGenerator)
    torch.nn.Module.__init__(gen); gen.renderer = rend
    sp_input, _ = gen.prepare_sp_input(d['t_vertices'].float(), gen.canonical_obs_vertices(d))
    planes = to(fx['planes']).requires_grad_(True)
    obs_feat = to(fx['obs_feat']).requires_grad_(True)
    vfeat = to(fx['vertex_feat']).requires_grad_(True)
    obs_img = d['obs_img_all'][:, 0]
    ro, rd, nr, fr = d['ray_o_all'][:, 0], d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0]
    opts = dict(fx['options'])
    R = ro.shape[1]
    # training_loop.py:231-236: rank 0's weights go to every rank before the first step.  (Ranks other than 0 are perturbed first, so that the broadcast is what
    # makes them equal: `dist.params_equal_after_broadcast` in the JSON line -- every rank's checksum of its parameters and buffers, compared on rank 0.)
    dist_info = None
    if world > 1:
        if rank != 0:
            with torch.no_grad():
                for p_ in list(rend.parameters()) + list(dec.parameters()):
                    p_.mul_(1.0 + 1e-3 * rank)
        n_sent = sdist.broadcast_params([rend, dec])
        cs = torch.stack([t_.detach().double().sum() for t_ in list(rend.parameters()) + list(rend.buffers()) + list(dec.parameters())]).sum().reshape(1).to(dev)
        allcs = [torch.empty_like(cs) for _ in range(world)]
        torch.distributed.all_gather(allcs, cs)
        dist_info = dict(broadcast_tensors=n_sent, params_equal_after_broadcast=bool(all(float(c) == float(allcs[0]) for c in allcs)))
    params = [p for p in list(rend.parameters()) + list(dec.parameters())]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.0, 0.99), eps=1e-8)          # train.py: G_opt_kwargs
    g = torch.Generator(device='cpu').manual_seed(11 + rank)
    t_rgb = (torch.rand(1, R, 3, generator=g) * 2 - 1).to(dev); t_acc = torch.rand(1, R, 1, generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        sp = SparseConvTensor(vfeat, sp_input['coord'], sp_input['out_sh'], 1)
        rgb, depth, acc = rend(planes, obs_img, obs_feat, sp, None, sp_input, dec, ro, rd, nr, fr, d, opts)
        loss = ((rgb - t_rgb) ** 2).mean() + ((acc - t_acc) ** 2).mean()
        loss.backward()
        sdist.allreduce_flat_grads(params)
        opt.step()
        return loss

    # the dominant kernel of the step, timed live: events on the stream the call is launched on (torch's current stream)
    from sherf_amd import _lib
    scatter_ev, call0 = [], _lib.call

    def timed_call(name, *args):
        if name != 'sherf_gather_tokens_bwd_binned':
            return call0(name, *args)
        e0, e1 = _Ev(), _Ev()
        e0.record(); call0(name, *args); e1.record()
        scatter_ev.append((e0, e1))
    _lib.call = timed_call
    phase_ev = []
    step0 = step

    host_t = []

    def step():                                             # the same step with three more events on the stream
        ev = [_Ev() for _ in range(4)]
        opt.zero_grad(set_to_none=True)
        h0 = time.perf_counter()
        ev[0].record()
        sp = SparseConvTensor(vfeat, sp_input['coord'], sp_input['out_sh'], 1)
        rgb, depth, acc = rend(planes, obs_img, obs_feat, sp, None, sp_input, dec, ro, rd, nr, fr, d, opts)
        loss = ((rgb - t_rgb) ** 2).mean() + ((acc - t_acc) ** 2).mean()
        ev[1].record()
        h1 = time.perf_counter()
        loss.backward()
        ev[2].record()
        h2 = time.perf_counter()
        sdist.allreduce_flat_grads(params)
        opt.step()
        ev[3].record()
        host_t.append((h1 - h0, h2 - h1, time.perf_counter() - h2))      # host time of each phase's enqueue (the backward's includes its one read-back)
        phase_ev.append(ev)
        return loss

    for _ in range(a.warmup):
        step()
    scatter_ev.clear(); phase_ev.clear(); host_t.clear()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    scat_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in scatter_ev])) if scatter_ev else None
    phases = {k: float(np.mean([ev[i].elapsed_time(ev[i + 1]) for ev in phase_ev])) for i, k in enumerate(('forward', 'backward', 'allreduce_adam'))}
    if a.pmc_child:
        return
    if rank == 0:
        n_valid = int(rend.last['ws']['counters'][0]) if rend.last else 0
        # algorithmic bytes of the scatter per valid sample: d_tokens 384 B + geometry 32 B read; fp32 read-modify-write of the taps:
        # tri-planes 3 slots x 4 taps x 128 B, feature map 2 slots x 4 taps x 128 B, voxel rows 3 levels x 8 taps x 3 slots x 128 B
        # Round 6: rounds 2-5 priced this kernel with the UNMERGED algorithm's read-modify-write bytes (23 968 B per
        # sample), which the run-length form no longer moves -- 6.8 TB/s "achieved" was above what the memory system can do.  Now: `achieved` =
        # the bytes the kernel cannot avoid (d_tokens 384 B + geometry 32 B + its place in the sorted order 8 B per valid sample, read once)
        # over its duration; `traffic` = what it actually moves (FETCH_SIZE x 2 + WRITE_SIZE of two rocprofv3 --pmc passes of this script,
        # gfx950 rule of MI355X_MICROARCH.md), per launch.  The kernel is bound by atomic adds, not by bandwidth: the honest fraction is small.
        bytes_per_sample = 384 + 32 + 8
        roofline = None
        if scat_ms and n_valid:
            ach = bytes_per_sample * n_valid / (scat_ms * 1e-3) / 1e9
            traffic = None
            under_profiler = any(k.startswith(('ROCPROF', 'ROCP_TOOL')) for k in os.environ)      # (never a counter pass inside somebody else's rocprofv3 run)
            if world == 1 and not a.no_pmc and not under_profiler:
                t = bench.pmc_traffic(a, lrank, timeout=240, child=[sys.executable, os.path.abspath(__file__), '--pmc-child', '--config', a.config],
                                      keys=('gather_tokens_bwd_runs_kernel',))
                traffic = t.get('hbm_bytes_per_launch') if isinstance(t, dict) and 'error' not in t else t
            roofline = dict(kernel='gather_tokens_bwd_runs_kernel (the tap scatter of sherf_gather_tokens_bwd_binned; events also cover its bin count / scans / fill)', bound='hbm',
                            achieved=ach, peak=bench.PEAK_HBM_GBS, unit='GB/s', frac=ach / bench.PEAK_HBM_GBS,
                            traffic=traffic, kernel_ms=scat_ms, bytes_per_sample=bytes_per_sample, valid_samples=n_valid,
                            traffic_gbs=(traffic / (scat_ms * 1e-3) / 1e9) if isinstance(traffic, (int, float)) else None,
                            note='largest single kernel of the step; `achieved` counts only the bytes it must read once (d_tokens, geometry, sort order); fp32 atomic adds of '
                                 'whole rows (lane = channel) over the samples sorted by their finest voxel cell, run-length sums in registers: bound by the number of atomically '
                                 'added elements, not by bandwidth (round 2 direct form: 20.3 ms, round 3 binned: 4.1 ms, round 5: 2.4 ms; profiles/r05_call_x_*)')
        print(json.dumps(dict(metric='training rays/sec at 512x512x64 (forward + backward + flat-grad all-reduce + Adam)', value=world * R * a.steps / dt,
                              unit='rays/s', n_gpus=world, rccl_ranks=torch.distributed.get_world_size() if world > 1 else 1, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * dt / a.steps, higher_is_better=True,
                              scaling='weak', vs_baseline=None, dtype='f32 (backward: fp32 kernels, MFMA GEMMs on a three-part bf16 split, MFMA sparse-conv input gradient on a range-scaled fp16 split; forward: f16x3 MFMA)',
                              data='synthetic', final_loss=float(loss), phases_ms=phases,
                              host_ms=dict(zip(('forward', 'backward', 'allreduce_adam'), (1e3 * np.mean(host_t, 0)).tolist())), roofline=roofline,
                              dist=dist_info, config=dict(workload=f'{a.config}: one view per GPU, stub loss MSE(rgb)+MSE(acc)', rays=R))))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
