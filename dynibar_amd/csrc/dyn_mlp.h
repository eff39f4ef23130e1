// Register-resident MLP chain engine for gfx950 (fp32-in / fp32-accumulate MFMA, v_mfma_f32_32x32x2_f32).
//
// Every Linear layer of the per-point networks (reference ibrnet/mlp_network.py) is evaluated TRANSPOSED:
//     out^T [features x rows] = W [features x K] . act^T [K x rows]
// so the weights are the MFMA A operand and the activations the B operand.  One wavefront owns a tile of 32 rows
// (point-views or points): lane l = (j = l & 31 : the row, h = l >> 5 : the half).  The MFMA result layout
// ("D layout": register r of lane (j,h) holds feature (r&3) + 8*(r>>2) + 4*h of the 32-feature output tile, for row j) is
// exactly what the next layer's B operand wants when its k-steps are enumerated in that same order, so activations never
// leave the register file between layers: no LDS round trip, no transposes.  The summation order over K is a
// pack-time permutation of the reference's (results agree to fp32 round-off, not bitwise; tolerance 1e-4 per north_star).
//
// Weights are pre-packed on the host (dyn_nets.hip: pack_layer) into a stream of 16 KiB chunks in consumption order and
// DMA'd global->LDS (global_load_lds_dwordx4) into a 2-deep ring shared by the 4 waves of a workgroup: chunk c+1 is in
// flight while chunk c feeds the MFMAs; one workgroup barrier per chunk (4096 MFMA-cycles per wave).
// Bias is folded into K as one extra k-slot whose activation is the constant 1.
#pragma once
#include "dyn_device.h"

#define DYN_CHUNK 4096        // floats per weight chunk (16 KiB)
#define DYN_NET_THREADS 256   // 4 waves per workgroup share one weight ring

// feature index (within a 32-feature tile) held in register r by a lane of half h
__host__ __device__ constexpr int dyn_fi(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct WeightRing {
  const float* gsrc;  // stream + tid*4 (per-thread source of the DMA)
  float* buf;         // LDS: 2 chunks
  int next;           // next chunk to consume
  int total;          // chunks in the stream
};

__device__ __forceinline__ void ring_issue(const WeightRing& R, int chunk) {
  const float* g = R.gsrc + (long)chunk * DYN_CHUNK;
  // wave-uniform LDS base; the hardware adds lane*16 bytes
  float* l = R.buf + (chunk & 1) * DYN_CHUNK + (threadIdx.x >> 6) * 256;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 1024),
                                     (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
}

__device__ __forceinline__ void ring_init(WeightRing& R, const float* stream, int total, float* lds) {
  R.gsrc = stream + threadIdx.x * 4;
  R.buf = lds;
  R.next = 0;
  R.total = total;
  ring_issue(R, 0);
}

// Returns the LDS image of the next chunk.  The barrier both publishes every wave's part of that chunk (each wave first
// drains its own DMA: s_waitcnt vmcnt(0)) and retires all reads of the other buffer, which is then refilled.
__device__ __forceinline__ const float* ring_acquire(WeightRing& R) {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  __syncthreads();
  const int c = R.next++;
  if (c + 1 < R.total) ring_issue(R, c + 1);
  return R.buf + (c & 1) * DYN_CHUNK;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// One Linear layer: NT output tiles of 32 features, NSTEPS k-steps (2 input features each).  feed(s) returns the lane's B
// operand (its activation for k-step s); s is a compile-time constant after unrolling.
template <int NT, int NSTEPS, class Feed>
__device__ __forceinline__ void mlp_layer(WeightRing& R, f32x16 (&acc)[NT], Feed&& feed) {
  constexpr int NSG = (NSTEPS + 3) / 4;
  constexpr int SGC = 16 / NT;
  constexpr int NCH = (NSG + SGC - 1) / SGC;
  static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8 || NT == 16, "tiles per layer must divide 16");
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float* buf = ring_acquire(R);
#pragma unroll
    for (int g = 0; g < SGC; ++g) {
      const int sg = c * SGC + g;
      if (sg < NSG) {
        // the lane's B operands of this group of four k-steps (evaluated once, shared by all output tiles)
        const float b0 = (sg * 4 + 0 < NSTEPS) ? feed(sg * 4 + 0) : 0.f;
        const float b1 = (sg * 4 + 1 < NSTEPS) ? feed(sg * 4 + 1) : 0.f;
        const float b2 = (sg * 4 + 2 < NSTEPS) ? feed(sg * 4 + 2) : 0.f;
        const float b3 = (sg * 4 + 3 < NSTEPS) ? feed(sg * 4 + 3) : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 a = *reinterpret_cast<const float4*>(buf + ((g * NT + t) * 64 + lane) * 4);
          if (sg * 4 + 0 < NSTEPS) acc[t] = mfma32(a.x, b0, acc[t]);
          if (sg * 4 + 1 < NSTEPS) acc[t] = mfma32(a.y, b1, acc[t]);
          if (sg * 4 + 2 < NSTEPS) acc[t] = mfma32(a.z, b2, acc[t]);
          if (sg * 4 + 3 < NSTEPS) acc[t] = mfma32(a.w, b3, acc[t]);
        }
      }
    }
  }
}

__host__ __device__ constexpr int dyn_layer_chunks(int NT, int NSTEPS) {
  return (((NSTEPS + 3) / 4) + (16 / NT) - 1) / (16 / NT);
}

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : (__expf(v) - 1.0f); }
__device__ __forceinline__ float sigmoid1(float v) { return 1.0f / (1.0f + __expf(-v)); }

template <int NT>
__device__ __forceinline__ void acc_elu(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = elu1(acc[t][r]);
}

// ---- reductions over the VSEG consecutive lanes (views, padded to a power of two) of one point: xor butterflies, so every lane
// of the segment ends with the bit-identical result.  (V, seg_base are unused; kept so call sites read like the maths.)
template <int VSEG>
__device__ __forceinline__ float seg_sum(float v, int, int) {
#pragma unroll
  for (int m = 1; m < VSEG; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
template <int VSEG>
__device__ __forceinline__ float seg_min(float v, int, int) {
#pragma unroll
  for (int m = 1; m < VSEG; m <<= 1) v = fminf(v, __shfl_xor(v, m));
  return v;
}
template <int VSEG>
__device__ __forceinline__ float seg_max(float v, int, int) {
#pragma unroll
  for (int m = 1; m < VSEG; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// dot product of the lane's 16*NTI activation registers with a [2][16*NTI] table in LDS (row h), summed over both halves
template <int NTI>
__device__ __forceinline__ float row_dot(const f32x16 (&act)[NTI], const float* tab) {
  const int h = (threadIdx.x & 63) >> 5;
  const float4* t4 = reinterpret_cast<const float4*>(tab + h * 16 * NTI);
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NTI; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = t4[t * 4 + q];
      s = fmaf(act[t][q * 4 + 0], w.x, s);
      s = fmaf(act[t][q * 4 + 1], w.y, s);
      s = fmaf(act[t][q * 4 + 2], w.z, s);
      s = fmaf(act[t][q * 4 + 3], w.w, s);
    }
  return s + __shfl_xor(s, 32);
}
