#!/bin/bash
# MFMA-pipe and wait counters per kernel for the default bench (one separate --pmc pass; run through gpurun), summarised by
# tools/pmc_mfma_summary.py into profiles/rNN_pmc_mfma_config2_b8.txt
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_mfma
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -o t -- python3 bench.py --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-kernel-timer --extras none > $O/bench.log 2>&1
python3 tools/pmc_mfma_summary.py $(find $O -name "t_counter_collection.csv") > $O/summary.txt
cat $O/summary.txt
