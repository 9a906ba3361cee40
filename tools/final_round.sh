#!/bin/bash
# Run on the GPU box (gpurun): the round's closing measurements.  usage: tools/final_round.sh TAG   -> gpurun_out/TAG_*
TAG=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2>> gpurun_out/${TAG}_bench.err
# the multi-rank launch path on this one-GPU box: 2 ranks (gloo, sharing the GPU), a strong-scaling shard of BASELINE configs[2]
# each, rollout segments all-gathered every 25 steps
python bench.py --gpus 2 \
  --steps 200 --warmup 50 --config 3 --backend gloo --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_gloo_1gpu.json 2>> gpurun_out/${TAG}_bench.err
bash tools/bench_configs.sh ${TAG} 1000 > gpurun_out/${TAG}_bench_configs.txt 2>&1
bash tools/profile_all.sh ${TAG}
