"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output): per-kernel totals, sorted by time.
    python tools/rocpd_stats.py trace_results.db [n_steps] > profiles/<name>.txt"""
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
