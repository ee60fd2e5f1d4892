#!/usr/bin/env python3
"""Copies the summaries of tools/refresh_profiles_bf16.sh from gpurun_out/prof_bf16 into profiles/ (tracked).
usage: collect_profiles_bf16.py r03"""
import glob, os, shutil, subprocess, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_bf16"), os.path.join(root, "profiles")
def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert hits, pattern
    return hits[0]
for c, b in ((3, 8), (5, 2)):
    name = f"config{c}_bf16_b{b}"
    shutil.copy(one(f"trace_c{c}/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_{name}.csv"))
    shutil.copy(os.path.join(src, f"timeline_c{c}.txt"), os.path.join(dst, f"{tag}_step_timeline_{name}.txt"))
    shutil.copy(os.path.join(src, f"pmc_mfma_c{c}.txt"), os.path.join(dst, f"{tag}_pmc_mfma_{name}.txt"))
    js = os.path.join(dst, f"{tag}_pmc_traffic_{name}.json")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), one(f"pmc_fetch_c{c}/**/*counter_collection.csv"),
                          one(f"pmc_write_c{c}/**/*counter_collection.csv"), js], capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{tag}_pmc_traffic_{name}.txt"), "w").write(out)
    print(open(os.path.join(dst, f"{tag}_step_timeline_{name}.txt")).read()[-120:])
