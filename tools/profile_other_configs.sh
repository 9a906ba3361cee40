#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE.json configurations at their per-GPU sizes (one pass each)
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$name -o prof -- python bench.py --steps 300 --warmup 1500 --no-cpu-baseline "$@" > gpurun_out/prof_${TAG}_$name.log 2>&1
  grep -h "rex_step_kernel\|rex_settle" gpurun_out/prof_${TAG}_$name/prof_kernel_stats.csv | sed "s/^/$name,/"; }
run gallop_ol_8192 --task gallop --signal ol --envs-per-gpu 8192
run turn_ik_terrain_4096 --task turn --signal ik --terrain random --envs-per-gpu 4096
run arm_walk_ik_2048 --mark arm --envs-per-gpu 2048
run walk_ik_65536 --envs-per-gpu 65536 --steps 100 --warmup 600
