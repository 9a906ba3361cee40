#!/bin/bash
# other BASELINE.json configs at their per-GPU sizes + large batches (documentation lines, not the headline)
TAG=${1:-r01g}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_bench_configs.jsonl; : > $OUT
B="python bench.py --steps 500 --warmup 1500 --no-cpu-baseline"
$B --task gallop --signal ol --envs-per-gpu 8192 >> $OUT
$B --task turn --signal ik --terrain random --envs-per-gpu 4096 >> $OUT
$B --mixed --mark arm --envs-per-gpu 2048 >> $OUT
$B --mark arm >> $OUT
$B --task standup --signal ol >> $OUT
$B --task poses >> $OUT
$B --envs-per-gpu 16384 >> $OUT
$B --envs-per-gpu 65536 --steps 200 --warmup 600 >> $OUT
$B --envs-per-gpu 262144 --steps 100 --warmup 300 >> $OUT
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_configs.jsonl"):
    d=json.loads(l); print("%-58s %8.3f ms/step %8.2f M env-steps/s"%(d["metric"][30:]+" | "+d["config"]["workload"][:22], d["ms_per_step"], d["value"]/1e6))
PY
