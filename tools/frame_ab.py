"""A/B timings of whole frames under `sherf_set_debug` settings, interleaved in ONE process on one box (GPU box only):

    python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0x20000,0x30000,0x40000 [--names whole,2parts,...]
                             [--opts ";main_after_layer=2;near_lists=False"]      one `key=value,key=value` group per arm (rendering options)

Every arm renders the bench frame with the given debug word; rounds are interleaved after a clock warm-up, the first arm's output is the
reference every other arm's rgb / depth / acc is compared with bit for bit."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeline(w, frames=12):
    """HIP-event timeline of `frames` frames recorded by the native driver (sherf_profile_frames): mean ms since the frame's first event."""
    import ctypes as ct
    import numpy as np
    import bench
    from sherf_amd import _lib
    for _ in range(3):
        bench.render_frame(w)
    torch.cuda.synchronize()
    _lib.call('sherf_profile_frames', 1)
    for _ in range(frames):
        bench.render_frame(w)
    torch.cuda.synchronize()
    ms = (ct.c_float * (64 * 8))(); n = ct.c_int32(0)
    _lib.call('sherf_profile_frames_read', ms, 64, ct.byref(n))
    _lib.call('sherf_profile_frames', 0)
    prof = np.array(ms[:n.value * 8], dtype=np.float64).reshape(-1, 8)
    names = ('host_enqueue', 'smpl_tables_done', 'encoder_done', 'rays_at_encoder_join', 'gather_done', 'mlp_done', 'frame_done', 'mlp_ms')
    return {k: round(float(v), 4) for k, v in zip(names, prof.mean(0))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2_dense_ri')
    ap.add_argument('--arms', default='0')
    ap.add_argument('--names', default='')
    ap.add_argument('--opts', default='')
    ap.add_argument('--exps', default='', help='one SHERF_EXPERIMENT word per arm (launch order / stream placement experiments: csrc/common.h)')
    ap.add_argument('--timeline', action='store_true', help='print the HIP-event timeline of every arm (bench.frame_timeline)')
    ap.add_argument('--dump', default='', help='save the first arm\'s rendered frame (rgb, depth, acc) to this file: frames of different LIBRARIES (SHERF_HIP_LIB, one process each) compared afterwards')
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    import bench
    from sherf_amd import _lib
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    w = bench.make_workload(argparse.Namespace(config=a.config, precision='auto', bn_mode='train'), 0.4, dev)
    arms = [int(x, 0) for x in a.arms.split(',')]
    names = a.names.split(',') if a.names else [hex(x) for x in arms]
    import ast
    groups = a.opts.split(';') if a.opts else []
    groups += [''] * (len(arms) - len(groups))
    arm_opts = [{kv.split('=')[0]: ast.literal_eval(kv.split('=')[1]) for kv in g.split(',') if kv} for g in groups]
    arm_graph = [o.pop('frame_graph', True) for o in arm_opts]
    base_opts = dict(w['opts'])
    exps = [x for x in a.exps.split(',') if x] + ['0'] * len(arms)

    def select(i):
        lib.sherf_set_debug(arms[i])
        lib.sherf_frame_graphs(1 if arm_graph[i] else 0)              # (`frame_graph=False` in an arm's options: frames enqueued launch by launch)
        os.environ['SHERF_EXPERIMENT'] = str(int(exps[i], 0) | (int(os.environ.get('SHERF_EXPERIMENT_BASE', '0'), 0)))
        w['opts'] = dict(base_opts, **arm_opts[i])
    lib = _lib.lib()
    lib.sherf_frame_graphs(0)                                     # (the warm-up renders launch by launch; every arm sets its own mode)
    os.environ['SHERF_EXPERIMENT'] = str(int(os.environ.get('SHERF_EXPERIMENT_BASE', '0'), 0))     # (bit 2 is read at the first frame)
    for _ in range(3):
        bench.render_frame(w)                                   # calibration of `auto` + warm-up
    torch.cuda.synchronize()
    print('configuration:', {k: w['rend'].last.get(k) for k in ('mlp_precision', 'table_precision', 'encoder_precision')})
    ref = None
    outs = {}
    for i, n in enumerate(names):
        select(i)
        r = bench.render_frame(w)
        torch.cuda.synchronize()
        outs[n] = [t.clone() for t in r]
        if ref is None:
            ref = outs[n]
        print(f'[bits] {n}: identical to {names[0]}: {all(torch.equal(p, q) for p, q in zip(outs[n], ref))}')
    if a.dump:
        torch.save([t.cpu() for t in ref], a.dump)
    for _ in range(30):
        bench.render_frame(w)
    times = {n: [] for n in names}
    for _ in range(a.rounds):
        for i, n in enumerate(names):
            select(i)
            for _ in range(3):
                bench.render_frame(w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                bench.render_frame(w)
            e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / a.iters)
    if a.timeline:
        for i, n in enumerate(names):
            select(i)
            print(f'[timeline] {n}: {timeline(w)}')
    lib.sherf_set_debug(0)
    import ctypes as ct
    gs = (ct.c_int64 * 4)()
    lib.sherf_frame_graph_stats(gs, 4)
    print('[graphs] captured %d, replayed frames %d, enqueued frames %d, failed captures %d' % tuple(int(v) for v in gs))
    if w['rend'].__dict__.get('form_report'):
        print('[mlp_form auto]', w['rend'].__dict__['form_report'])            # (the tuner's back-to-back launches, beside the in-frame `mlp_ms` of the timelines)
    for n in names:
        print(f'[arm] {n:12s} ms/frame {" ".join(f"{t:.4f}" for t in times[n])}   min {min(times[n]):.4f}')


if __name__ == '__main__':
    main()
