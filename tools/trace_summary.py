#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-launch durations and inter-kernel gaps of the last step."""
import csv
import sys

def _wgs(r) -> int:
    """workgroups of a launch (all three grid dimensions)"""
    n = 1
    for d in "XYZ":
        n *= max(int(r.get(f"Grid_Size_{d}", 1) or 1), 1) // max(int(r.get(f"Workgroup_Size_{d}", 1) or 1), 1) or 1
    return n


path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows]
# one step = from one stem launch to the next
idx = [i for i, r in enumerate(ours) if "stem_conv7x7" in r["Kernel_Name"] or "stem_pool7x7" in r["Kernel_Name"]]
if len(idx) >= 2:
    lo, hi = idx[-2], idx[-1]
else:
    lo, hi = 0, len(ours)
step = ours[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    print(f"{(s - t0) / 1e3:9.1f}us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  grid {_wgs(r):6d}  {name}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step wall {(prev_end - t0) / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, launches {len(step)}")
