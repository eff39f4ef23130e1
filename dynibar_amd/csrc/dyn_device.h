// Common device helpers for the gfx950 kernels of dynibar_amd.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// All LDS of every kernel lives in ONE dynamic array whose base is 16-byte aligned
// (cdna_hip_programming.md guideline 17: no static __shared__ in front of the dynamic region).
extern __shared__ __attribute__((aligned(16))) float4 dyn_smem[];

#define DYN_WAVE 64

__device__ __forceinline__ int dyn_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int dyn_wave() { return threadIdx.x >> 6; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// Inclusive product scan across the 64 lanes (Hillis-Steele on shuffles).
__device__ __forceinline__ float wave_inclusive_prod(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d);
    if (lane >= d) v *= o;
  }
  return v;
}
