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
  DYN_K_TRAJECTORY, DYN_K_RENDER_FLOWS, DYN_K_SCENE_FLOW, DYN_K_IMAGE_RAYS, DYN_K_STATIC_POINTS_QKV, DYN_K_DYNAMIC_POINTS_QKV,
  DYN_K_ENC_CONV7, DYN_K_ENC_CONV3, DYN_K_ENC_CONV1, DYN_K_ENC_BLOCK,
  DYN_K_TRAIN_GEMM, DYN_K_TRAIN_ROWS, DYN_K_TRAIN_ATTN, DYN_K_TRAIN_GATHER_BWD, DYN_K_MOTION_TAIL, DYN_K_STATIC_PLAN, DYN_K_COUNT
};
void dyn_prof_begin(int slot, hipStream_t stream);
void dyn_prof_end(int slot, hipStream_t stream);

// Dynamic LDS beyond 64 KiB must be opted into per kernel AND per device, and the opt-in is a maximum: a call site whose LDS size
// varies (k_fine_samples: 72-144 KiB with S + N) raises it again whenever a launch needs more than any earlier one on that device.
// The launch goes to the CURRENT device; a stream that belongs to another device is refused instead of launched wrongly.
#define DYN_MAX_DEVICES 16
#define DYN_LAUNCH(slot, name, kernel, grid, block, shmem, stream, ...)              \
  do {                                                                              \
    int dev_ = 0, sdev_ = -1;                                                       \
    (void)hipGetDevice(&dev_);                                                      \
    if ((stream) != nullptr && hipStreamGetDevice((stream), &sdev_) == hipSuccess && sdev_ != dev_) { \
      dyn_set_error("%s: the stream belongs to device %d but the current device is %d (select the tensors' device first)", name, sdev_, dev_); \
      return DYN_E_INVALID;                                                         \
    }                                                                               \
    if ((size_t)(shmem) > 65536) {                                                  \
      static int attr_max_[DYN_MAX_DEVICES] = {0};                                  \
      const int slot_ = (dev_ >= 0 && dev_ < DYN_MAX_DEVICES) ? dev_ : 0;           \
      if ((int)(shmem) > attr_max_[slot_] || dev_ >= DYN_MAX_DEVICES) {             \
        hipError_t ea_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(shmem)); \
        if (ea_ != hipSuccess) {                                                    \
          dyn_set_error("%s: cannot reserve %zu bytes of LDS: %s", name, (size_t)(shmem), hipGetErrorString(ea_)); \
          return DYN_E_LAUNCH;                                                      \
        }                                                                           \
        attr_max_[slot_] = (int)(shmem);                                            \
      }                                                                             \
    }                                                                               \
    dyn_prof_begin(slot, stream);                                                   \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);            \
    dyn_prof_end(slot, stream);                                                     \
    DYN_CHECK_LAUNCH(name);                                                         \
  } while (0)
