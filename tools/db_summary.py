#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 results.db (kernel-trace): ms per step and launches per step.
usage: db_summary.py results.db steps [top]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    n = n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
    tot[n][0] += 1; tot[n][1] += e - s
T = sum(v[1] for v in tot.values())
print(f"launches {len(rows)}  kernel time {T/1e6/steps:.2f} ms/step over {steps} steps (incl. warm-up)")
for n, v in sorted(tot.items(), key=lambda x: -x[1][1])[:top]:
    print(f"{v[1]/1e6/steps:8.3f} ms/step {v[0]/steps:7.1f} x  {n[:90]}")
