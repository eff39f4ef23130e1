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

// Streaming accesses: data written once for a later kernel, or read exactly once, can go around the 4 MiB L2 slices (non-temporal) so
// that the re-read working set (source maps, the packed weight stream) stays resident.
typedef float dyn_f32x4 __attribute__((ext_vector_type(4)));
// Measured inside the bench pipeline: the view kernel's stores of the parked feature / point records and the point kernel's loads of
// those records (group 4): view kernel -1 %, point kernel -6 %.  The gather's outputs (group 1): 335 MB of plain stores push the source
// maps out of the L2s and the memory-side cache while the taps still need them: with non-temporal stores (and the map prefetch of
// k_project_gather_tile) the gather runs 86 us instead of 102 us inside the pipeline; its consumer then finds less of them in the
// 256 MB Infinity Cache (view kernel +2 %).  Not the view kernel's gathered inputs (+6 %), not the blend kernel's loads of the parked
// feature (+2 %).
#ifndef DYN_NT
#define DYN_NT 5  /* bit 0: gather outputs, bit 1: view kernel's gathered inputs, bit 2: parked features / point records, bit 3: blend loads */
#endif
template <int GROUP>
__device__ __forceinline__ void nt_store4(float4* p, float4 v) {
  if (DYN_NT & GROUP) {
    const dyn_f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<dyn_f32x4*>(p));
  } else {
    *p = v;
  }
}
template <int GROUP>
__device__ __forceinline__ void nt_store1(float* p, float v) {
  if (DYN_NT & GROUP) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int GROUP>
__device__ __forceinline__ float4 nt_load4(const float4* p) {
  if (DYN_NT & GROUP) {
    const dyn_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const dyn_f32x4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
  }
  return *p;
}
template <int GROUP>
__device__ __forceinline__ float nt_load1(const float* p) {
  return (DYN_NT & GROUP) ? __builtin_nontemporal_load(p) : *p;
}

