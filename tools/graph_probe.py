"""Where does capturing a frame into a hipGraph break?  (GPU box only; round 6: hipStreamEndCapture crashed inside sherf_render_frame's own capture
while tools/ubench/graph_probe.hip -- the same fork / join shape on the system runtime -- works.)  One experiment per child process:

  torch_all     torch.cuda.graph() around a whole frame (the library's own capture off): torch's capture stream, begin / end by torch
  torch_noaux   the same without the third stream (rendering option aux_stream=False)
  native        the library's own capture (SHERF_FRAME_GRAPH_DEBUG trace)
  native_noaux  the same without the third stream
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode):
    import ctypes as ct
    import torch
    import bench
    from sherf_amd import _lib
    dev = torch.device('cuda', 0)
    w = bench.make_workload(argparse.Namespace(config='cfg1_ri', precision='f16', bn_mode='train'), 0.4, dev)
    if 'noaux' in mode:
        w['opts']['aux_stream'] = False
    native = mode.startswith('native')
    _lib.call('sherf_frame_graphs', 1 if native else 0)
    for _ in range(3 if not native else 1):
        r = bench.render_frame(w)
    torch.cuda.synchronize()
    ref = [t.clone() for t in r]
    print(f'[{mode}] eager frames done', flush=True)
    if native:
        for i in range(4):
            r = bench.render_frame(w)
            torch.cuda.synchronize()
            print(f'[{mode}] frame {i}: identical {all(torch.equal(a, b) for a, b in zip(r, ref))}', flush=True)
        s = (ct.c_int64 * 4)(); _lib.call('sherf_frame_graph_stats', s, 4)
        print(f'[{mode}] stats captured {s[0]} replayed {s[1]} eager {s[2]} failed {s[3]}', flush=True)
        return
    rend = w['rend']
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        bench.render_frame(w)                                  # the workspace of THIS caller stream exists before the capture
        torch.cuda.synchronize()
        print(f'[{mode}] begin capture', flush=True)
        with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
            out = bench.render_frame(w)
        print(f'[{mode}] captured', flush=True)
        g.replay(); torch.cuda.synchronize()
        print(f'[{mode}] replayed: identical {all(torch.equal(a, b) for a, b in zip(out, ref))}', flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--child', default='')
    ap.add_argument('--modes', default='torch_all,torch_noaux,native,native_noaux')
    a = ap.parse_args()
    if a.child:
        return child(a.child)
    for m in a.modes.split(','):
        env = dict(os.environ, SHERF_FRAME_GRAPH_DEBUG='1')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', m], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith('[') or 'Error' in l or 'error' in l]
        print('\n'.join(lines[-12:]))
        print(f'== {m}: rc {r.returncode}', flush=True)


if __name__ == '__main__':
    main()
