#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + three PMC passes (HBM read bytes, HBM write bytes, SQ issue / wait
# cycles next to GRBM_GUI_ACTIVE) of bench.py for one workload.  Counters are collected in their own passes, with
# --kernel-trace only.   usage: tools/profile_round.sh TAG [bench.py workload flags ...]   -> gpurun_out/TAG_*
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
B="--no-cpu-baseline --no-walking-workload --no-gather --no-segment-launch --no-closed-loop"   # the per-step rex_step launches alone
TO=${REX_PROFILE_TIMEOUT:-150}
timeout $TO rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 50 $B "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
timeout $TO rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_FETCH_SIZE -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 50 $B "$@" > /dev/null 2>&1
timeout $TO rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_WRITE_SIZE -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 50 $B "$@" > /dev/null 2>&1
timeout $TO rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_SQ -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 50 $B "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
# (launches 1550 .. 1949 of the step kernel = bench.py's 400 timed launches behind its 1 500-step pre-roll and the 50 warm-up steps:
#  the window bench.py's own device-side kernel_ms covers -- the two figures of one run, side by side)
python tools/rocprof_stats.py $(find gpurun_out/${TAG}_prof -name "*_results.db" | head -1) --window 1550:1950 --csv gpurun_out/${TAG}_kernel_stats.csv | head -4
python - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_under_rocprof.json"):
    if l.startswith("{"):
        d = json.loads(l); print(json.dumps({"bench_kernel_ms_same_launches": d["roofline"]["kernel_ms"], "bench_ms_per_step": d["ms_per_step"]}))
PY
python tools/rocprof_pmc.py $(find gpurun_out/${TAG}_pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find gpurun_out/${TAG}_pmc_WRITE_SIZE -name "*_results.db" | head -1) $(find gpurun_out/${TAG}_pmc_SQ -name "*_results.db" | head -1)
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE gpurun_out/${TAG}_pmc_SQ   # the databases are tens of MB: gpurun copies back <= 64 MiB
