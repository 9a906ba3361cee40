#!/bin/bash
# tools/profile_all.sh TAG: profile_round.sh for the workloads DESIGN.md quotes -> gpurun_out/TAG_profile_<workload>.txt
TAG=$1
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; bash tools/profile_round.sh ${TAG}_$name "$@" > gpurun_out/${TAG}_profile_$name.txt 2>&1; }
run walk4096
run walk262144 --envs-per-gpu 262144
run arm4096 --mark arm
run mixedarm2048 --mixed --mark arm --envs-per-gpu 2048
run gallop8192 --task gallop --signal ol --envs-per-gpu 8192
run turnhf4096 --task turn --terrain random
run poses4096 --task poses
