set -o pipefail
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "default rc=$?"
for spec in "3 fp32 8" "5 fp32 2" "1 fp32 8" "2 bf16 8"; do set -- $spec; python bench.py --config $1 --dtype $2 --batch $3 --extras none --no-cpu-baseline --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg', '$1', '$2', 'B$3', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms')"; done
for b in 1 4; do python bench.py --batch $b --extras none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg 2 fp32 B$b', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms')"; done
python bench.py --conv f32 --extras none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg 2 conv f32 B8', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms', d['roofline']['frac'])"
python bench.py --mode train --extras none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms', d['roofline']['frac'])"
