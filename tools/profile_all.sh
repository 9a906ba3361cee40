#!/bin/bash
# tools/profile_all.sh TAG: profile_round.sh for the workloads DESIGN.md quotes -> gpurun_out/TAG_profile_<workload>.txt
TAG=$1
cd $GRAFT_REPO_ROOT
# (each workload under its own timeout; a killed one leaves its rocprofv3 databases behind: removed, gpurun copies back <= 64 MiB)
run() { name=$1; shift; timeout 240 bash tools/profile_round.sh ${TAG}_$name "$@" > gpurun_out/${TAG}_profile_$name.txt 2>&1; rm -rf gpurun_out/${TAG}_${name}_prof gpurun_out/${TAG}_${name}_pmc_*; }
run walk4096
run walk262144 --envs-per-gpu 262144
run arm4096 --mark arm
run mixedarm2048 --mixed --mark arm --envs-per-gpu 2048
run gallop8192 --task gallop --signal ol --envs-per-gpu 8192
run turnhf4096 --task turn --terrain random
run poses4096 --task poses
