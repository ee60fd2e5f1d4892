#!/bin/bash
# Runs on the GPU box (through gpurun): one MFMA / LDS --pmc pass and the FETCH / WRITE passes of the training leg (BASELINE config 4).
# Usage: gpurun -- 'bash tools/refresh_profiles_train.sh'; then python tools/collect_profiles_train.py rNN
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_train
rm -rf $O && mkdir -p $O
ARGS="bench.py --mode train --no-cpu-baseline --no-kernel-timer --extras none --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o t -- python3 $ARGS > $O/pmc_mfma.log 2>&1
python3 tools/pmc_mfma_summary.py $(find $O/pmc_mfma -name "t_counter_collection.csv") > $O/pmc_mfma.txt
echo "pmc mfma done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o t -- python3 $ARGS > $O/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o t -- python3 $ARGS > $O/pmc_write.log 2>&1
echo "pmc write done"
head -12 $O/pmc_mfma.txt
