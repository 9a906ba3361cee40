#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + two PMC passes (HBM read / write bytes) of bench.py for one
# workload.  usage: tools/profile_round.sh TAG [bench.py workload flags ...]   -> gpurun_out/TAG_*
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-walking-workload "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 50 --no-cpu-baseline --no-walking-workload "$@" > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocprof_stats.py $(find gpurun_out/${TAG}_prof -name "*_results.db" | head -1) --skip 1550 | head -4
python tools/rocprof_pmc.py $(find gpurun_out/${TAG}_pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/${TAG}_pmc_WRITE_SIZE -name "*_results.db" | head -1)
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE   # the databases are tens of MB: gpurun copies back <= 64 MiB
