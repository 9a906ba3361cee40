#!/bin/bash
# compile rexsim.hip with -save-temps into scratch/isa, print the resource usage of every rex_step_kernel
# instantiation and the instruction/loop statistics of one of them: tools/kstat.sh [EPW=4] [ARM=0]
set -e
cd "$(dirname "$0")/.."
EPW=${1:-4}; ARM=${2:-0}
mkdir -p scratch/isa && cd scratch/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -Rpass-analysis=kernel-resource-usage -c ../../rex_gym_amd/csrc/rexsim.hip -o /dev/null 2>&1 \
  | grep -A9 "Function Name: .*rex_step_kernel" | grep -E "Name|VGPRs:|Scratch|VGPRs Spill" | sed 's/.*remark: //' || true
cd ../..
python tools/isa_stats.py scratch/isa/rexsim-hip-amdgcn-amd-amdhsa-gfx950.s "rex_step_kernelILi${EPW}ELb${ARM}E" 12
