"""Power / clock telemetry of GPU 0 while a command runs (GPU box only): samples the amdgpu hwmon files (socket power, shader clock) every ~20 ms
in a thread around a child command and prints their distribution -- the evidence behind "the network kernel is power-bound" (DESIGN section 5.1):

    python tools/power_probe.py -- python tools/mlp_ab.py --config cfg2_dense_ri --forms pp --rounds 6
"""
import glob
import os
import subprocess
import sys
import threading
import time


def find_files():
    out = {}
    for hw in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
        for name in ('power1_average', 'power1_input', 'freq1_input', 'power1_cap'):
            p = os.path.join(hw, name)
            if os.path.exists(p) and name not in out:
                out[name] = p
        if out:
            break
    return out


def main():
    cmd = sys.argv[sys.argv.index('--') + 1:]
    files = find_files()
    print('[power_probe] files:', files, flush=True)
    samples, stop = [], [False]

    def read(p):
        try:
            return int(open(p).read().strip())
        except Exception:
            return None

    def loop():
        while not stop[0]:
            samples.append((time.perf_counter(), {k: read(p) for k, p in files.items() if k != 'power1_cap'}))
            time.sleep(0.02)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.perf_counter()
    rc = subprocess.run(cmd).returncode
    stop[0] = True
    th.join(timeout=1)
    # only the samples inside the command's [sustain] window, when it prints one (a sustained load of seconds: the hwmon averages are slow)
    win = None
    try:
        txt = open(os.environ.get('POWER_PROBE_LOG', '/dev/null')).read()
        import re
        a_, b_ = re.search(r'\[sustain\] start ([0-9.]+)', txt), re.search(r'\[sustain\] end ([0-9.]+)', txt)
        if a_ and b_:
            win = (float(a_.group(1)) + 0.5, float(b_.group(1)))
    except Exception:
        pass
    if win:
        off = time.time() - time.perf_counter()
        samples = [s for s in samples if win[0] <= s[0] + off <= win[1]]
        print(f'[power_probe] {len(samples)} samples inside the sustained window ({win[1] - win[0]:.1f} s)')
    cap = read(files['power1_cap']) if 'power1_cap' in files else None
    print(f'[power_probe] command rc={rc}, {time.perf_counter() - t0:.1f} s, {len(samples)} samples; power cap {cap / 1e6 if cap else None} W')
    for key, unit, scale in (('power1_average', 'W', 1e-6), ('power1_input', 'W', 1e-6), ('freq1_input', 'MHz', 1e-6)):
        v = sorted(s[1][key] * scale for s in samples if s[1].get(key) is not None)
        if v:
            q = lambda f: v[min(len(v) - 1, int(f * len(v)))]
            print(f'[power_probe] {key:15s} {unit}: min {v[0]:.0f}  p10 {q(0.1):.0f}  p50 {q(0.5):.0f}  p90 {q(0.9):.0f}  max {v[-1]:.0f}')
    # the busiest stretch: the 20 % of the samples with the highest power
    key = 'power1_average' if any(s[1].get('power1_average') for s in samples) else 'power1_input'
    hot = sorted((s for s in samples if s[1].get(key)), key=lambda s: -s[1][key])[:max(1, len(samples) // 5)]
    if hot and hot[0][1].get('freq1_input') is not None:
        f = sorted(s[1]['freq1_input'] * 1e-6 for s in hot)
        print(f'[power_probe] shader clock during the top-20 % power samples: median {f[len(f) // 2]:.0f} MHz (min {f[0]:.0f}, max {f[-1]:.0f}); '
              f'their power: median {sorted(s[1][key] for s in hot)[len(hot) // 2] * 1e-6:.0f} W')


if __name__ == '__main__':
    main()
