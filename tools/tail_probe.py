"""Where do the ~25 us per frame between a natively looped frame (mlp_form='auto' tuner: 1.544 ms) and the same frame through ImportanceRenderer.forward (1.572 ms)
go?  The frame the bench renders, looped four ways in one process, interleaved: (a) forward(); (b) sherf_render_frame alone on the descriptor forward() left; (c) + the
encoder's running-statistics update (SparseConvNet.finish); (d) + the counters' read-back of the flag watch.
    python tools/tail_probe.py [--config cfg2_dense_ri] [--rounds 5] [--iters 40]"""
import argparse
import ctypes as ct
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from sherf_amd import _lib                                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2_dense_ri')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=40)
    a = ap.parse_args()
    dev = bench._device(0)
    w = bench.make_workload(argparse.Namespace(config=a.config, precision='auto', bn_mode='train'), 0.4, dev)
    rend = w['rend']
    for _ in range(6):
        bench.render_frame(w)                                    # calibration, form tuning, warm-up
    torch.cuda.synchronize()
    wsp = rend._workspace(dev)
    fr, last = wsp.desc[2], rend.last                           # (the cached frame descriptor, as the last forward() left it)
    levels, pl, ws = last['levels_struct'], last['plan'], last['ws']
    main_s = torch.cuda.current_stream(dev)
    s_main = ct.c_void_p(main_s.cuda_stream)
    s_side = ct.c_void_p(rend._side(dev).cuda_stream)
    s_aux = ct.c_void_p(rend._side(dev, 1).cuda_stream)
    native = lambda: _lib.call('sherf_render_frame', ct.byref(fr), 3, levels, s_main, s_side, s_aux)
    host = torch.zeros(8, dtype=torch.int32).pin_memory()

    def native_finish():
        native(); rend.encoder_3d.finish(pl)

    def native_finish_copy():
        native(); rend.encoder_3d.finish(pl); host.copy_(ws['counters'], non_blocking=True)

    real_watch, real_finish = rend._flag_watch, rend.encoder_3d.finish

    def forward_without(watch, finish):
        def f():
            if not watch:
                rend.__dict__['_flag_watch'] = lambda ws_, dev_: None
            if not finish:
                rend.encoder_3d.__dict__['finish'] = lambda pl_: None
            try:
                bench.render_frame(w)
            finally:
                rend.__dict__.pop('_flag_watch', None); rend.encoder_3d.__dict__.pop('finish', None)
        return f

    arms = dict(forward=lambda: bench.render_frame(w), forward_no_watch=forward_without(False, True), forward_no_watch_no_finish=forward_without(False, False),
                native=native, native_finish=native_finish, native_finish_copy=native_finish_copy)
    times = {k: [] for k in arms}
    for _ in range(a.rounds):
        for k, f in arms.items():
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / a.iters)
    for k, v in times.items():
        print(f'[arm] {k:20s} ms/frame {" ".join(f"{t:.4f}" for t in v)}   min {min(v):.4f}')


if __name__ == '__main__':
    main()
