#!/bin/bash
# Run on the GPU box (gpurun): four bench lines (4 096 walk, 8 192 gallop, 16 384, 262 144 envs; 600-step windows) for each of
# several builds of the library, back to back on one box -- under a minute per build.  The fast parity subset to run first:
#   python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "settled_snapshot or single_step_parity_from_common or joint_limit_rows or kernel_variants_agree or ragged or regrouped"
# usage: tools/ab_quick.sh TAG lib1.so lib2.so ...   -> gpurun_out/TAG_ab.txt
TAG=$1; shift
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG}_ab.txt; : > $OUT
one() { local lib=$1; shift
  REX_LIB_PATH=$PWD/$lib python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-walking-workload "$@" 2>> gpurun_out/${TAG}_ab.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %-24s %8.2f M env-steps/s  kernel %.4f ms (min %.4f)' % ('$lib'.split('/')[-1], d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min']))" >> $OUT; }
cfg() { echo "== $*" >> $OUT; for lib in "${LIBS[@]}"; do one $lib "$@"; done; }
LIBS=("$@")
cfg --envs-per-gpu 4096
cfg --task gallop --signal ol --envs-per-gpu 8192
cfg --envs-per-gpu 16384
cfg --envs-per-gpu 262144
cat $OUT
