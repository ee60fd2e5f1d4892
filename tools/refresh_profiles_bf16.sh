#!/bin/bash
# Runs on the GPU box (through gpurun): kernel traces + one MFMA/LDS --pmc pass + FETCH/WRITE passes for the bf16 legs
# (BASELINE configs 3 and 5).  Usage: gpurun -- 'bash tools/refresh_profiles_bf16.sh'; then tools/collect_profiles_bf16.py rNN
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_bf16
rm -rf $O && mkdir -p $O
for CB in "3 8" "5 2"; do
  set -- $CB; C=$1; B=$2
  ARGS="bench.py --config $C --dtype bf16 --batch $B --no-cpu-baseline --no-kernel-timer --extras none"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c$C -o t -- python3 $ARGS --steps 5 --warmup 2 > $O/bench_c$C.log 2>&1
  echo "trace config $C done"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_c$C -o t -- python3 $ARGS --steps 2 --warmup 1 > $O/pmc_mfma_c$C.log 2>&1
  python3 tools/pmc_mfma_summary.py $(find $O/pmc_mfma_c$C -name "t_counter_collection.csv") > $O/pmc_mfma_c$C.txt
  echo "pmc mfma config $C done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c$C -o t -- python3 $ARGS --steps 2 --warmup 1 > $O/pmc_fetch_c$C.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_c$C -o t -- python3 $ARGS --steps 2 --warmup 1 > $O/pmc_write_c$C.log 2>&1
  echo "pmc traffic config $C done"
  python3 tools/trace_summary.py $(find $O/trace_c$C -name "t_kernel_trace.csv") > $O/timeline_c$C.txt
  tail -3 $O/timeline_c$C.txt
done
