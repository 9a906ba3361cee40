#!/bin/bash
# Run on the GPU box (gpurun): the round's closing measurements.  usage: tools/final_round.sh TAG   -> gpurun_out/TAG_*
TAG=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2>> gpurun_out/${TAG}_bench.err
OUT=gpurun_out/${TAG}_bench_configs.jsonl
: > $OUT
run() { python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-walking-workload "$@" 2>> gpurun_out/${TAG}_bench.err | tail -1 >> $OUT; }
run --task gallop --signal ol --envs-per-gpu 8192
run --task turn --terrain random
run --mark arm
run --mixed --mark arm --envs-per-gpu 2048
run --mixed --mark arm --envs-per-gpu 16384
run --task standup --signal ol
run --task poses
run --envs-per-gpu 16384
run --envs-per-gpu 65536
run --envs-per-gpu 262144
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}_profile.txt 2>&1
