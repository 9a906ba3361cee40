#!/bin/bash
# Run on the GPU box (gpurun): the same bench lines with several builds of the library, back to back on one box
# (boxes differ by a few per cent, so only numbers of one call compare).  usage: tools/ab_libs.sh TAG STEPS lib1.so lib2.so ...
TAG=$1; STEPS=$2; shift 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab.txt
: > $OUT
one() {   # lib, bench args...
  local lib=$1; shift
  REX_LIB_PATH=$PWD/$lib python bench.py --steps $STEPS --warmup 100 --no-cpu-baseline --no-walking-workload "$@" 2>> gpurun_out/${TAG}_ab.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %-24s %8.2f M env-steps/s  kernel %.4f ms' % ('$lib'.split('/')[-1], d['value']/1e6, d['roofline']['kernel_ms']))" >> $OUT
}
cfg() { echo "== $*" >> $OUT; for lib in "${LIBS[@]}"; do one $lib "$@"; done; }
LIBS=("$@")
cfg --envs-per-gpu 4096
cfg --task gallop --signal ol --envs-per-gpu 8192
cfg --mark arm
cfg --mixed --mark arm --envs-per-gpu 2048
cfg --task poses
cfg --envs-per-gpu 16384
cfg --envs-per-gpu 65536
cfg --envs-per-gpu 262144
cat $OUT
