// rex_step_mixed_arm.hip -- instantiates the kernels of one variant group (rex_kernels.h): mark arm, REX_TASK_MIXED (BASELINE.json configs[4]).
#include "rex_kernels.h"

void REX_STEP_LAUNCHER(mixed_arm)(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
  REX_LAUNCH_BY_EPW(true, true, false);
}
