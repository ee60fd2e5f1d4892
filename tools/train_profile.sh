#!/bin/bash
# rocprofv3 kernel stats of the training step (run through gpurun): prints the top kernels of 1 warm-up + 3 steps.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_train
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer --extras none > $O/bench.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms over 4 steps", round(tot / 1e6, 2))
for r in rows[:18]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(5), "%8.2f" % (float(r["TotalDurationNs"]) / 1e6), "%8.1f" % (float(r["AverageNs"]) / 1e3))
PY
