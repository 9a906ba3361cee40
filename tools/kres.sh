#!/bin/bash
# resource usage (AGPRs, scratch) of every kernel of one variant group's translation unit, with extra compiler flags:
#   tools/kres.sh step_arm '-DREX_TARGET_BY_DIVISION(EPW,ARM)=0' ...
cd "$(dirname "$0")/.."
G=$1; shift
FLAGS=$(python -c "from rex_gym_amd.build import COMPILE_FLAGS; print(' '.join(COMPILE_FLAGS))")   # the library's own compile flags
D=scratch/isa_$$_$RANDOM
mkdir -p $D && cd $D
hipcc $FLAGS -Rpass-analysis=kernel-resource-usage "$@" -I ../../rex_gym_amd/csrc -c ../../rex_gym_amd/csrc/rex_${G}.hip -o /dev/null 2>&1 \
  | grep -E "Function Name: .*rex_(step|settle)_kernel|AGPRs:|ScratchSize" | sed 's/.*Function Name: _ZN3rex15rex_step_kernel/K /; s/.*Function Name: _ZN3rex17rex_settle_kernel/S /; s/EEEv.*//; s/.*remark: *//; s/\[-Rpass.*//' | paste - - -
cd ../..; rm -rf $D
