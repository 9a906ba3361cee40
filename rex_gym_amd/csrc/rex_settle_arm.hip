// rex_settle_arm.hip -- instantiates the kernels of one variant group (rex_kernels.h): the reset motion, mark arm (16 envs per wave).
#include "rex_kernels.h"

void rex_launch_settle_arm(RexSim* s, int nrec, hipStream_t st, float* snap) {
  if (s->cfg.body_contacts) hipLaunchKernelGGL((rex::rex_settle_kernel<true, true>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
  else hipLaunchKernelGGL((rex::rex_settle_kernel<true, false>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
}
