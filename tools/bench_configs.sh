#!/bin/bash
# Run on the GPU box (gpurun): the bench line of every configuration the DESIGN table lists.  usage: tools/bench_configs.sh TAG [quick]
TAG=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_bench_configs.jsonl
: > $OUT
STEPS=${2:-1000}
run() { python bench.py --steps $STEPS --warmup 100 --no-cpu-baseline --no-walking-workload "$@" 2>> gpurun_out/${TAG}_bench.err | tail -1 >> $OUT; }
run
run --task gallop --signal ol --envs-per-gpu 8192
run --task turn --terrain random
run --mark arm
run --mixed --mark arm --envs-per-gpu 2048
run --mixed --mark arm --envs-per-gpu 16384
run --task standup --signal ol
run --task poses
run --envs-per-gpu 8192                                      # north_star's shard: 65 536 / 8
run --task gallop --signal ol --envs-per-gpu 65536           # the WHOLE of configs[2] on one GPU (strong-scaling expectation)
run --task turn --terrain random --envs-per-gpu 32768        # the whole of configs[3]
run --envs-per-gpu 16384
run --envs-per-gpu 65536
run --envs-per-gpu 262144
python - <<PY
import json
for l in open("$OUT"):
    try:
        d = json.loads(l)
        print("%-110s %8.2f M env-steps/s  %.4f ms/step  kernel %.4f ms" % (d["config"]["workload"][:110], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"]))
    except Exception as e:
        print("bad line", e)
PY
