// rex_step_mixed_base.hip -- instantiates the kernels of one variant group (rex_kernels.h): mark base, REX_TASK_MIXED.
#include "rex_kernels.h"

void REX_STEP_LAUNCHER(mixed_base)(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
  REX_LAUNCH_BY_EPW(false, true, false);
}
