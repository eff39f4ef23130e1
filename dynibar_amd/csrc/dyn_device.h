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


// -------------------------------------------------------------------------------------------------------------------
// The Pluecker moment the way the reference forms it.  render_ray.py:375 and :392 call torch.cross WITHOUT `dim`, and torch then
// crosses over the FIRST axis of size 3 -- not necessarily xyz: with exactly 3 source views the moments of compute_src_plucker_coordinate
// ([V, R, S, 3] operands) are products over the VIEW axis, with a chunk of exactly 3 rays over the rays (also compute_ref_plucker_coordinate's
// [R, 3] operands), with exactly 3 samples per ray over the samples.  A DynibarStatic trained with 3 source views has learnt from those
// values, so a drop-in has to produce them: every kernel that forms a moment takes the axis from the call's shape exactly like torch does
// (rounds 1-5 crossed over xyz always and warned).  c_i = a_{i+1} b_{i+2} - a_{i+2} b_{i+1} along the axis, per xyz component.
// -------------------------------------------------------------------------------------------------------------------
enum { DYN_CROSS_XYZ = 0, DYN_CROSS_VIEW = 1, DYN_CROSS_RAY = 2, DYN_CROSS_SAMPLE = 3 };
__host__ __device__ inline int dyn_src_cross_axis(int V, int R, int S) {
  return V == 3 ? DYN_CROSS_VIEW : (R == 3 ? DYN_CROSS_RAY : (S == 3 ? DYN_CROSS_SAMPLE : DYN_CROSS_XYZ));
}
__host__ __device__ inline int dyn_ref_cross_axis(int R) { return R == 3 ? DYN_CROSS_RAY : DYN_CROSS_XYZ; }

__device__ __forceinline__ void dyn_unit3(float x, float y, float z, float& ox, float& oy, float& oz) {
  const float d = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);  // F.normalize(eps=1e-12)
  ox = x / d; oy = y / d; oz = z / d;
}

// moment of the source ray (view v, ray r, sample s) when the product runs over `axis` (not xyz).  pts_at(v, r, s, q[3]): the point that
// view sees (the same point for every view unless the caller has per-view points); ctr_at(v, c[3]): the view's camera centre.
template <class PtsAt, class CtrAt>
__device__ __forceinline__ void dyn_src_moment_over_axis(int axis, int v, int r, int s, PtsAt pts_at, CtrAt ctr_at, float (&m)[3]) {
  const int i = axis == DYN_CROSS_VIEW ? v : (axis == DYN_CROSS_RAY ? r : s);
  float a[2][3], b[2][3];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int jj = (i + 1 + e) % 3;
    const int vv = axis == DYN_CROSS_VIEW ? jj : v, rr = axis == DYN_CROSS_RAY ? jj : r, ss = axis == DYN_CROSS_SAMPLE ? jj : s;
    float q[3];
    pts_at(vv, rr, ss, q);
    ctr_at(vv, a[e]);
    dyn_unit3(q[0] - a[e][0], q[1] - a[e][1], q[2] - a[e][2], b[e][0], b[e][1], b[e][2]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) m[k] = a[0][k] * b[1][k] - a[1][k] * b[0][k];
}

// moment of target ray r of a chunk of exactly 3 rays (compute_ref_plucker_coordinate crossing over the rays)
__device__ __forceinline__ void dyn_ref_moment_over_rays(const float* __restrict__ ray_o, const float* __restrict__ ray_d, int r, float (&m)[3]) {
  const int r1 = (r + 1) % 3, r2 = (r + 2) % 3;
  float d1[3], d2[3];
  dyn_unit3(ray_d[r1 * 3], ray_d[r1 * 3 + 1], ray_d[r1 * 3 + 2], d1[0], d1[1], d1[2]);
  dyn_unit3(ray_d[r2 * 3], ray_d[r2 * 3 + 1], ray_d[r2 * 3 + 2], d2[0], d2[1], d2[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) m[k] = ray_o[r1 * 3 + k] * d2[k] - ray_o[r2 * 3 + k] * d1[k];
}
