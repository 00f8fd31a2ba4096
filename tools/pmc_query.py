"""Print per-kernel averages of every counter in a rocprofv3 --pmc rocpd database.
    python tools/pmc_query.py results.db [kernel-substring ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")] if 'pmc_events' in tabs else []
if not cols:
    print('no pmc_events view; tables:', tabs)
    sys.exit(0)
print('# pmc_events columns:', cols)
# heuristics for column names across rocprofiler-sdk versions
kname = next((c for c in cols if c in ('name', 'kernel_name')), None)
cname = next((c for c in cols if c in ('counter_name', 'pmc_name', 'symbol')), None)
vname = next((c for c in cols if c in ('value', 'counter_value')), None)
if not (kname and cname and vname):
    for r in db.execute("select * from pmc_events limit 3"):
        print(r)
    sys.exit(0)
q = f"select {kname}, {cname}, count(*), avg({vname}) from pmc_events group by {kname}, {cname}"
rows = db.execute(q).fetchall()
out = {}
for k, c, n, v in rows:
    if pats and not any(p in k for p in pats):
        continue
    out.setdefault(k, {})[c] = (n, v)
for k, d in out.items():
    print(k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:60])
    for c, (n, v) in sorted(d.items()):
        print(f'    {c:34s} n={n:4d} avg={v:.4g}')
