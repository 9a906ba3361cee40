#!/bin/bash
# Reproduces the round-3 "link-box kernels with the leg factor in registers give wrong states" build:
# the tree of commit 007fcba with REX_LEG_F4_OF forced to 1 for <4, base, BODY>, built as it is (fails) and with each of the
# compiler switches that were tried in round 4; then runs the poses tests on each library.  Build part: any host with hipcc;
# test part: on the GPU box (gpurun).     usage: tools/repro_body_miscompile.sh build | test
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$ROOT/scratch/t007
D='-DREX_LEG_F4_OF(EPW,ARM,BODY)=((((EPW)<=8)||((EPW)<=16&&(ARM)))?1:REX_LEG_F4)'
if [ "$1" = build ]; then
  rm -rf $T && mkdir -p $T && git -C $ROOT archive 007fcba | tar -x -C $T
  make -C $T/oracle > /dev/null
  # (that commit defines the macro unconditionally: make it overridable)
  sed -i 's|^#define REX_LEG_F4_OF(EPW, ARM, BODY) \(.*\)$|#ifndef REX_LEG_F4_OF\n#define REX_LEG_F4_OF(EPW, ARM, BODY) \1\n#endif|' $T/rex_gym_amd/csrc/rex_device.h
  b() { name=$1; shift; (cd $T && python -c "
import sys; sys.path.insert(0, '$T')
from rex_gym_amd import build as hb
print(hb.build(force=True, lib_path='$T/$name.so', defines=sys.argv[1:], only='body,base', jobs=4))" "$@"); }
  b parked                                             # the workaround of that commit: factor parked in LDS        -> passes
  b registers "$D"                                     # factor in registers                                        -> FAILS (epw 4)
  b no_strict_aliasing "$D" -fno-strict-aliasing       #                                                            -> passes
  b auto_var_init "$D" -ftrivial-auto-var-init=pattern #                                                            -> passes
  b no_agpr_spill "$D" -mllvm -amdgpu-spill-vgpr-to-agpr=0      #                                                   -> FAILS
  b no_liverange_opt "$D" -mllvm -amdgpu-opt-vgpr-liverange=false   #                                               -> FAILS
  b waitcnt_zero "$D" -mllvm -amdgpu-waitcnt-forcezero #                                                            -> FAILS
  b no_misched "$D" -mllvm -enable-misched=false       # no pre-RA machine scheduler                                -> passes
  b no_post_sched "$D" -mllvm -enable-post-misched=false   #                                                        -> FAILS
else
  cd $T
  for L in parked registers no_strict_aliasing auto_var_init no_agpr_spill no_liverange_opt waitcnt_zero no_misched no_post_sched; do
    echo "== $L"
    REX_LIB_PATH=$T/$L.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "self_collision_rows_of_the_rolled_pose" 2>&1 | tail -3
  done
fi
