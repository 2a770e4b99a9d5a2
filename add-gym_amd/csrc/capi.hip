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
extern "C" int addhip_version(void) { return 1; }
