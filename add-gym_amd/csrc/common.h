// Shared host-side plumbing of libaddhip: error text, launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "addhip.h"

namespace addhip {
void set_error(const char* fmt, ...);
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}
}  // namespace addhip

#define ADDHIP_REQUIRE(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      addhip::set_error(__VA_ARGS__);    \
      return -1;                         \
    }                                    \
  } while (0)

#define ADDHIP_HIP(call)                                              \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      addhip::set_error("%s: %s", #call, hipGetErrorString(e_));      \
      return -2;                                                      \
    }                                                                 \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
