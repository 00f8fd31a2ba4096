"""Where is the HOST while the GPU renders?  (GPU box only; SHERF_EXPERIMENT bit 4: csrc/common.h)

    python tools/host_stamps.py [--config cfg2_dense_ri] 2> stamps.txt

Renders a few back-to-back frames with the native driver printing the host clock at its enqueue points, then prints, per frame and relative
to the moment the host's wait for the frame's sample count ENDS (= the compaction's end on the GPU, ~0.55 ms into the frame's GPU time):
when the host entered the frame, queued the encoder's first launch, reached scatter_rows, finished the encoder, began to wait."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import torch
    import bench
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    w = bench.make_workload(argparse.Namespace(config=a.config, precision='auto', bn_mode='train'), 0.4, dev)
    for _ in range(20):
        bench.render_frame(w)
    torch.cuda.synchronize()
    for _ in range(10):
        bench.render_frame(w)
    os.environ['SHERF_EXPERIMENT'] = str(a.exp)
    for _ in range(a.frames):
        bench.render_frame(w)
    os.environ['SHERF_EXPERIMENT'] = '0'
    for _ in range(3):
        bench.render_frame(w)
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2_dense_ri')
    ap.add_argument('--frames', type=int, default=6)
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--exp', type=int, default=16, help='SHERF_EXPERIMENT word of the stamped frames (16 host clock + in-stream host functions; 32 event trail along the encoder stream)')
    a = ap.parse_args()
    if a.child:
        return child(a)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--config', a.config, '--frames', str(a.frames), '--exp', str(a.exp)], capture_output=True, text=True)
    rows = [l.split(None, 2) for l in r.stderr.splitlines() if l.startswith('[host]')]
    for l in r.stderr.splitlines():
        if l.startswith('[trail]'):
            print(l)
    frames, cur = [], None
    for _, t, what in rows:
        if what == 'frame enter':
            cur = []
            frames.append(cur)
        if cur is not None:
            cur.append((float(t), what))
    prev_end = None
    # the in-stream stamps ("gpu: ...") arrive from a runtime thread, later than the frame that queued them: listed separately, absolute order kept
    gpu = [(float(t), w) for _, t, w in rows if w.startswith('gpu:')]
    frames = [[(t, w) for t, w in fr if not w.startswith('gpu:')] for fr in frames]
    t00 = frames[0][0][0] if frames else 0.0
    for t, w in gpu:
        print(f'[gpu] {t - t00:9.0f} us  {w}')
    for fr in frames:
        print(f'[host] {fr[0][0] - t00:9.0f} us  frame enter;  scatter_rows queued {dict((w, t) for t, w in fr).get("scatter_rows next", 0) - t00:9.0f} us')
    for k, fr in enumerate(frames):
        d = dict((w, t) for t, w in fr)
        ref = d.get('count wait ends', fr[0][0])          # (no host wait in the frame: relative to its entry)
        line = '  '.join(f'{w} {t - ref:+.0f}' for t, w in fr)
        print(f'[frame {k}] us relative to the end of the count wait: {line}' + (f'   (period {ref - prev_end:.0f} us)' if prev_end else ''))
        prev_end = ref
    if not frames and '[trail]' not in r.stderr:
        print(r.stderr[-2000:])


if __name__ == '__main__':
    main()
