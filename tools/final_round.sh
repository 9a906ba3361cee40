#!/bin/bash
# Run on the GPU box (gpurun): the round's closing measurements.  usage: tools/final_round.sh TAG   -> gpurun_out/TAG_*
# Every step runs under its own `timeout` and stamps gpurun_out/TAG_progress.txt: round 4's closing call sat in one of these steps
# until gpurun's 40-minute limit killed it and nothing came back.  Cheap essentials first.
TAG=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=gpurun_out/${TAG}_progress.txt; : > $P
step() { echo "$(date +%s) start $1" >> $P; shift; "$@"; echo "$(date +%s) rc=$?" >> $P; }
step bench_default   timeout 240 bash -c "python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err"
step bench_driver    timeout 120 bash -c "python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2>> gpurun_out/${TAG}_bench.err"
# the multi-rank launch path on this one-GPU box: 2 ranks (gloo, sharing the GPU), a strong-scaling shard of BASELINE configs[2]
# each, rollout segments all-gathered every 25 steps -- through the plain spelling (bench.py starts its own ranks)
step bench_2rank     timeout 240 bash -c "python bench.py --gpus 2 --steps 200 --warmup 50 --config 3 --backend gloo --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_gloo_1gpu.json 2>> gpurun_out/${TAG}_bench.err"
step bench_configs   timeout 600 bash -c "bash tools/bench_configs.sh ${TAG} 1000 > gpurun_out/${TAG}_bench_configs.txt 2>&1"
step profile_all     timeout 900 bash tools/profile_all.sh ${TAG}
cat $P
