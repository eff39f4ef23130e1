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

// ---- optional per-kernel timing (dyn_profile_*): HIP events recorded on the launch stream around every kernel ----
enum DynKernelSlot {
  DYN_K_PREPARE_CAMERAS, DYN_K_NCHW_TO_NHWC, DYN_K_SAMPLE, DYN_K_POINTS_FROM_Z, DYN_K_PROJECT_GATHER, DYN_K_SAMPLE_MASK, DYN_K_COMPOSITE,
  DYN_K_FINE_SAMPLES, DYN_K_STATIC_REF, DYN_K_STATIC_VIEWS, DYN_K_STATIC_POINTS, DYN_K_STATIC_BLEND, DYN_K_SELFTEST, DYN_K_DYNAMIC_TIME, DYN_K_DYNAMIC_VIEWS, DYN_K_DYNAMIC_POINTS, DYN_K_MOTION_MLP,
  DYN_K_TRAJECTORY, DYN_K_RENDER_FLOWS, DYN_K_SCENE_FLOW, DYN_K_IMAGE_RAYS, DYN_K_STATIC_POINTS_QKV, DYN_K_DYNAMIC_POINTS_QKV, DYN_K_COUNT
};
void dyn_prof_begin(int slot, hipStream_t stream);
void dyn_prof_end(int slot, hipStream_t stream);

#define DYN_LAUNCH(slot, name, kernel, grid, block, shmem, stream, ...)              \
  do {                                                                              \
    if ((size_t)(shmem) > 65536) { /* dynamic LDS beyond 64 KiB must be opted into, once per kernel */ \
      static bool attr_done_ = false;                                               \
      if (!attr_done_) {                                                            \
        hipError_t ea_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(shmem)); \
        if (ea_ != hipSuccess) {                                                    \
          dyn_set_error("%s: cannot reserve %zu bytes of LDS: %s", name, (size_t)(shmem), hipGetErrorString(ea_)); \
          return DYN_E_LAUNCH;                                                      \
        }                                                                           \
        attr_done_ = true;                                                          \
      }                                                                             \
    }                                                                               \
    dyn_prof_begin(slot, stream);                                                   \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);            \
    dyn_prof_end(slot, stream);                                                     \
    DYN_CHECK_LAUNCH(name);                                                         \
  } while (0)
