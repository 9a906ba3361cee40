// rex_settle_base.hip -- instantiates the kernels of one variant group (rex_kernels.h): the reset motion, mark base (toes only: one env per lane; link-box rows: 16 envs per wave).
#include "rex_kernels.h"

void rex_launch_settle_base(RexSim* s, int nrec, hipStream_t st, float* snap) {
  if (s->cfg.body_contacts) hipLaunchKernelGGL((rex::rex_settle_kernel<false, true>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
  else hipLaunchKernelGGL((rex::rex_settle_kernel<false, false>), dim3((nrec + REX_WAVE - 1) / REX_WAVE), dim3(REX_WAVE), 0, st, s->dev, snap);
}
