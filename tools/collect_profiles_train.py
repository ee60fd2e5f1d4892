#!/usr/bin/env python3
"""Copies the summaries of tools/refresh_profiles_train.sh from gpurun_out/prof_train into profiles/ (tracked).  usage: collect_profiles_train.py r03"""
import glob, os, shutil, subprocess, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_train"), os.path.join(root, "profiles")
one = lambda pattern: glob.glob(os.path.join(src, pattern), recursive=True)[0]
shutil.copy(os.path.join(src, "pmc_mfma.txt"), os.path.join(dst, f"{tag}_pmc_mfma_config4_train_b8.txt"))
js = os.path.join(dst, f"{tag}_pmc_traffic_config4_train_b8.json")
out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), one("pmc_fetch/**/*counter_collection.csv"),
                      one("pmc_write/**/*counter_collection.csv"), js], capture_output=True, text=True, check=True).stdout
open(os.path.join(dst, f"{tag}_pmc_traffic_config4_train_b8.txt"), "w").write(out)
print(out[:600])
