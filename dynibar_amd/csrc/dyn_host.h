// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/dynibar_hip.h"

void dyn_set_error(const char* fmt, ...);

#define DYN_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      dyn_set_error(__VA_ARGS__);             \
      return DYN_E_INVALID;                   \
    }                                         \
  } while (0)

#define DYN_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      dyn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return DYN_E_LAUNCH;                                                  \
    }                                                                       \
  } while (0)

static inline int dyn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
