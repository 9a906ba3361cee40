// rex_step_base.hip -- instantiates the kernels of one variant group (rex_kernels.h): mark base, single task, toes only: 4 / 8 / 16 envs per wave (lane groups) and 64 (one env per lane).
#include "rex_kernels.h"

void REX_STEP_LAUNCHER(base)(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
#if !REX_TU_POL   /* (the fused actor spreads its neurons over the lanes of an env group: lane-group kernels only) */
  if (s->epw == 64) REX_LAUNCH_STEP(64, false, false, false);
  else
#endif
  REX_LAUNCH_BY_EPW(false, false, false);
}
