// Microbenchmark: throughput of "3 MFMAs + K VALU instructions" units in different instruction orders, two waves per SIMD
// (developer tool; decides the layout of the MLP engine's inner loop).
//   order 0: K VALU, then MMM              (same accumulator)
//   order 1: M K/3 VALU  M K/3 VALU  M K/3 VALU   (same accumulator)
//   order 2: as 1, three different accumulators
//   order 3: K VALU, then MMM on three different accumulators
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
#define VA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i) & 7]) : "v"(c));

template <int ORDER, int K3>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
  f32x16 acc0, acc1, acc2;
  for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 1e-3f; acc1[r] = r; acc2[r] = 2 * r; }
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float c = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (ORDER == 0 || ORDER == 3) {
#pragma unroll
        for (int i = 0; i < 3 * K3; ++i) VA(i)
        if (ORDER == 0) { MF(acc0) MF(acc0) MF(acc0) } else { MF(acc0) MF(acc1) MF(acc2) }
      } else {
        if (ORDER == 1) { MF(acc0) } else { MF(acc0) }
#pragma unroll
        for (int i = 0; i < K3; ++i) VA(i)
        if (ORDER == 1) { MF(acc0) } else { MF(acc1) }
#pragma unroll
        for (int i = 0; i < K3; ++i) VA(i + 3)
        if (ORDER == 1) { MF(acc0) } else { MF(acc2) }
#pragma unroll
        for (int i = 0; i < K3; ++i) VA(i + 6)
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int ORDER, int K3>
void run(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ORDER, K3>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<ORDER, K3>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 2 waves x 8 units per iteration
  printf("order %d  K=%2d VALU per 3 MFMA : %7.1f ns per unit-pair (two waves' units; 3+3 MFMAs alone = ~100 ns)\n", ORDER, 3 * K3,
         ms * 1e6 / (8.0 * iters));
}
template <int K3>
void row(float* out, int iters) {
  run<0, K3>(out, iters); run<1, K3>(out, iters); run<2, K3>(out, iters); run<3, K3>(out, iters);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  row<0>(out, iters); row<2>(out, iters); row<4>(out, iters); row<6>(out, iters); row<8>(out, iters); row<12>(out, iters); row<16>(out, iters);
  return 0;
}
