// libaddhip: error channel + version.  Kernels live in env_step.hip / gemm.hip / learn.hip.
#include <cstdarg>
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}
namespace addhip {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace addhip

extern "C" const char* addhip_last_error(void) { return g_err; }
extern "C" int addhip_version(void) { return 6; }
// sizeof() of the parameter blocks, in the order they are declared in include/addhip.h: lets a binding check its own layout
extern "C" int addhip_abi_sizes(int32_t* out, int32_t count) {
  const int32_t sizes[] = {(int32_t)sizeof(addhip_motion_t), (int32_t)sizeof(addhip_task_t), (int32_t)sizeof(addhip_env_t), (int32_t)sizeof(addhip_step_out_t),
                           (int32_t)sizeof(addhip_sampler_t), (int32_t)sizeof(addhip_gemm_t), (int32_t)sizeof(addhip_gather_t), (int32_t)sizeof(addhip_rigid_model_t),
                           (int32_t)sizeof(addhip_rigid_dr_t), (int32_t)sizeof(addhip_optimizer_t), (int32_t)sizeof(addhip_section_t),
                           (int32_t)sizeof(addhip_mlp_t), (int32_t)sizeof(addhip_extra_dw_t), (int32_t)sizeof(addhip_mlp_marks_t), (int32_t)sizeof(addhip_ppo_loss_t),
                           (int32_t)sizeof(addhip_ppo_marks_t), (int32_t)sizeof(addhip_disc_loss_t), (int32_t)sizeof(addhip_disc_marks_t),
                           (int32_t)sizeof(addhip_actor_head_t)};
  const int32_t n = (int32_t)(sizeof(sizes) / sizeof(sizes[0]));
  if (!out || count < n) return -1;
  for (int32_t i = 0; i < n; ++i) out[i] = sizes[i];
  return n;
}
