#!/bin/bash
# run on the GPU box: full GPU tests, default bench, rocprofv3 kernel stats, two PMC passes (HBM read / write bytes)
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; tail -3 gpurun_out/tests_$TAG.log
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o prof -- python bench.py --steps 300 --warmup 1500 --no-cpu-baseline > gpurun_out/bench_prof_$TAG.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o pmc -- python bench.py --steps 100 --warmup 1500 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$C.log 2>&1
done
find gpurun_out/prof_$TAG -name "*stats*" | head; 
