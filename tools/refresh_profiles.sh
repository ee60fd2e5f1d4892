#!/bin/bash
# Runs on the GPU box (through gpurun): kernel traces, PMC passes and the training trace behind profiles/.
# Usage: gpurun -- 'bash tools/refresh_profiles.sh'   then   python tools/collect_profiles.py <round>
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
for B in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b$B -o t -- python3 bench.py --steps 5 --warmup 2 --batch $B --no-cpu-baseline --no-kernel-timer --extras none > $O/bench_b$B.log 2>&1
  echo "trace b$B done"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o t -- python3 bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-kernel-timer --extras none > $O/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o t -- python3 bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-kernel-timer --extras none > $O/pmc_write.log 2>&1
echo "pmc write done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o t -- python3 bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer --extras none > $O/bench_train.log 2>&1
echo "train trace done"
find $O -name "*.csv" | head -30
