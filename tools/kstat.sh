#!/bin/bash
# compile one variant group's translation unit with -save-temps into scratch/isa, print the resource usage of every kernel
# instantiation in it and the instruction/loop statistics of one of them:
#   tools/kstat.sh [GROUP=step_base] [EPW=4] [ARM=0]     GROUP: step_base step_arm step_mixed_base step_mixed_arm step_body settle_base settle_arm
set -e
cd "$(dirname "$0")/.."
GROUP=${1:-step_base}; EPW=${2:-4}; ARM=${3:-0}
FLAGS=$(python -c "from rex_gym_amd.build import COMPILE_FLAGS; print(' '.join(COMPILE_FLAGS))")   # the library's own compile flags
mkdir -p scratch/isa && cd scratch/isa
hipcc $FLAGS -save-temps -Rpass-analysis=kernel-resource-usage -I ../../rex_gym_amd/csrc -c ../../rex_gym_amd/csrc/rex_${GROUP}.hip -o /dev/null 2>&1 \
  | grep -A9 "Function Name: .*rex_\(step\|settle\)_kernel" | grep -E "Name|VGPRs:|AGPRs|Scratch|VGPRs Spill|LDS Size" | sed 's/.*remark: //' || true
cd ../..
python tools/isa_stats.py scratch/isa/rex_${GROUP}-hip-amdgcn-amd-amdhsa-gfx950.s "rex_step_kernelILi${EPW}ELb${ARM}E" 12 || true
