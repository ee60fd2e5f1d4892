#!/usr/bin/env python3
"""Copies the summaries of tools/refresh_profiles.sh from gpurun_out/prof into profiles/ (tracked).
usage: collect_profiles.py r01"""
import glob, os, shutil, subprocess, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles")
def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert hits, pattern
    return hits[0]
for B in (1, 8):
    shutil.copy(one(f"trace_b{B}/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_config2_b{B}.csv"))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trace_summary.py"), one(f"trace_b{B}/**/*kernel_trace.csv")],
                         capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{tag}_step_timeline_config2_b{B}.txt"), "w").write(out)
js = os.path.join(dst, f"{tag}_pmc_traffic_config2_b8.json")
out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), one("pmc_fetch/**/*counter_collection.csv"),
                      one("pmc_write/**/*counter_collection.csv"), js], capture_output=True, text=True, check=True).stdout
open(os.path.join(dst, f"{tag}_pmc_traffic_config2_b8.txt"), "w").write(out)
shutil.copy(one("trace_train/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_config4_train_b8.csv"))
print(open(os.path.join(dst, f"{tag}_step_timeline_config2_b8.txt")).read()[-400:])
