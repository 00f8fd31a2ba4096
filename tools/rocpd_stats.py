"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output): per-kernel totals, sorted by time.
    python tools/rocpd_stats.py trace_results.db [n_steps] [top] [tail_n] > profiles/<name>.txt
tail_n: a second table over every kernel's LAST tail_n dispatches only -- the timed frames of a `bench.py --steps tail_n` run, without the calibration frames' and the
form-tuning launches (back to back, i.e. at another operating point of the power cap) that the first table averages in."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = 'name' if 'name' in cols else 'kernel_name'
rows = db.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
span = db.execute("select min(start), max(end) from kernels").fetchone()
print(f'# kernels: {sum(r[1] for r in rows)} dispatches, total kernel time {tot/1e6:.3f} ms, first->last span {(span[1]-span[0])/1e6:.3f} ms')
if steps:
    print(f'# per step (/{steps} incl. warmup): {tot/1e6/steps:.3f} ms of kernel time')
print(f'{"total_ms":>10} {"calls":>7} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"pct":>6}  name')
for n, c, t, a, mn, mx in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f'{t/1e6:10.3f} {c:7d} {a/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*t/tot:6.2f}  {n[:120]}')

tail_n = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if tail_n:
    print(f'\n# the last {tail_n} dispatches of each kernel (the timed frames)')
    print(f'{"total_ms":>10} {"calls":>7} {"avg_us":>10} {"min_us":>9} {"max_us":>9}  name')
    out = []
    for n, c, *_ in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
        d = [r[0] for r in db.execute(f"select end-start from kernels where {name} = ? order by start desc limit ?", (n, tail_n))]
        if len(d) == tail_n:                                   # (kernels that ran at least once per timed frame)
            out.append((sum(d), len(d), sum(d) / len(d), min(d), max(d), n))
    for t, c, a, mn, mx, n in sorted(out, reverse=True):
        print(f'{t/1e6:10.3f} {c:7d} {a/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f}  {n[:120]}')
