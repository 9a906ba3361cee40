// rex_step_body.hip -- instantiates the kernels of one variant group (rex_kernels.h): link-box contact rows (RexConfig.body_contacts): 4 or 8 envs per wave, mark arm 4 (rex_create caps it).
#include "rex_kernels.h"

void REX_STEP_LAUNCHER(body)(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
  if (s->cfg.mark == REX_MARK_ARM) REX_LAUNCH_STEP(4, true, false, true);
  else if (s->epw == 4) REX_LAUNCH_STEP(4, false, false, true);
  else REX_LAUNCH_STEP(8, false, false, true);
}
