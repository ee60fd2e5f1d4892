#!/bin/bash
# PMC passes around tools/conv3x3_bench.py (run through gpurun): MFMA / LDS / wait counters of the bf16 3x3 kernel per layer shape.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_c3
rm -rf $O && mkdir -p $O
LAYERS=${1:-layer1,layer3,fusion1_c3,head}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o t -- python3 tools/conv3x3_bench.py $LAYERS 1 > $O/p1.log 2>&1
python3 tools/pmc_mfma_summary.py $(find $O/p1 -name "t_counter_collection.csv") > $O/p1.txt
cat $O/p1.txt | head -20
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p2 -o t -- python3 tools/conv3x3_bench.py $LAYERS 1 > $O/p2.log 2>&1 || tail -5 $O/p2.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_c3/p2/**/t_counter_collection.csv", recursive=True)
if f:
    per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], name) not in seen:
            seen.add((r["Dispatch_Id"], name)); cnt[name] += 1
    for name, c in per.items():
        if "conv" in name:
            print(name, cnt[name], {k: f"{v / cnt[name]:.3g}" for k, v in c.items()})
PY
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -c . $O/counters.txt
