"""Timeline of the last bench step from a rocprofv3 rocpd database: kernel name, queue, start/end (us, relative).
    python tools/rocpd_timeline.py trace.db > timeline.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = next((c for c in ('queue_id', 'queue', 'stream_id', 'stream') if c in cols), None)
rows = db.execute(f"select name, {qcol or '0'}, start, end from kernels order by start").fetchall()
# last step = from the last 'smpl_bones_kernel' launch (first kernel of a frame on the side stream) / build_cells2
starts = [i for i, r in enumerate(rows) if 'smpl_bones' in r[0]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1]
# include side-stream kernels that started slightly before
t0 = min(r[2] for r in rows[max(0, i0 - 8):i0 + 1])
sel = [r for r in rows if r[2] >= t0 and r[2] < rows[i1][2] - 1]
print(f'# columns: cols={cols}')
print(f'# step window {(rows[i1][2]-t0)/1e3:.1f} us, {len(sel)} kernels')
prev_end = {}
for name, q, s, e in sel:
    short = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:44]
    gap = (s - prev_end.get(q, s)) / 1e3
    print(f'{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  q={q}  gap={gap:7.1f}  {short}')
    prev_end[q] = e
