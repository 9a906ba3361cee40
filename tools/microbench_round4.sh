#!/bin/bash
# Run on the GPU box (gpurun): round 4's micro-benchmarks -> gpurun_out/r04_row_ilp.txt, r04_calib.md
#   row_ilp.hip            issue rate of independent VGPR-operand fmas / DPP adds, the contact row with one and two envs interleaved
#   hbm_counter_calib.hip  FETCH_SIZE / WRITE_SIZE against a known byte count in the step kernel's access pattern
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/row_ilp tools/microbench/row_ilp.hip && /tmp/row_ilp > gpurun_out/r04_row_ilp.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_counter_calib tools/microbench/hbm_counter_calib.hip
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/calib_F -- /tmp/hbm_counter_calib 4096 16384 262144 > $GRAFT_REPO_ROOT/gpurun_out/r04_calib_known.txt 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/calib_W -- /tmp/hbm_counter_calib 4096 16384 262144 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/hbm_calib.py $(find gpurun_out/calib_F -name "*_results.db" | head -1) $(find gpurun_out/calib_W -name "*_results.db" | head -1) > gpurun_out/r04_calib.md 2>&1
rm -rf gpurun_out/calib_F gpurun_out/calib_W
cat gpurun_out/r04_row_ilp.txt gpurun_out/r04_calib_known.txt gpurun_out/r04_calib.md
