// rex_step_arm.hip -- instantiates the kernels of one variant group (rex_kernels.h): mark arm, single task, toes only.
#include "rex_kernels.h"

void REX_STEP_LAUNCHER(arm)(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
  REX_LAUNCH_BY_EPW(true, false, false);
}
