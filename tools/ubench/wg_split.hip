// Microbenchmark: one 8-wave workgroup per CU against two independent 4-wave workgroups per CU, same total waves per SIMD (2), for a kernel
// shaped like the view chain: layer phases (3 dependent MFMAs + KV fillers per pair, a workgroup barrier every 24 pairs) alternating with
// matrix-idle phases (KI VALU instructions, a barrier).  Prints the time per wave-iteration for both shapes (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ float smem[];
template <int THREADS, int KV, int KI>
__global__ void __launch_bounds__(THREADS, 2) k(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = threadIdx.x * 1e-3f + r + t;
  u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float c = 1.0001f;
  smem[threadIdx.x] = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int chunk = 0; chunk < 4; ++chunk) {       // a "layer": 4 chunks of 24 pairs, barrier per chunk
      __syncthreads();
#pragma unroll
      for (int pr = 0; pr < 24; ++pr) {
#pragma unroll
        for (int i = 0; i < KV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(c));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[pr & 3]) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[pr & 3]) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[pr & 3]) : "v"(a), "v"(b));
      }
    }
    __syncthreads();                                  // a matrix-idle phase (statistics, exchanges)
#pragma unroll 8
    for (int i = 0; i < KI; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(c));
  }
  float s = smem[threadIdx.x];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int THREADS, int KV, int KI>
float run(float* out, int iters, int lds) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k<THREADS, KV, KI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256 * 8 * (512 / THREADS);  // the same number of waves in both shapes, 8 rounds per CU
  hipLaunchKernelGGL((k<THREADS, KV, KI>), dim3(grid), dim3(THREADS), lds, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<THREADS, KV, KI>), dim3(grid), dim3(THREADS), lds, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / (8.0 * iters);  // us per iteration of one workgroup round
}
template <int KV, int KI>
void row(float* out) {
  const int iters = 40;
  const float one = run<512, KV, KI>(out, iters, 150 * 1024), two = run<256, KV, KI>(out, iters, 76 * 1024);
  printf("fillers per pair %2d, idle-phase VALU %4d:  1 x 8 waves %7.2f us   2 x 4 waves %7.2f us   ratio %.3f\n", KV, KI, one, two, two / one);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 512 * 4);
  row<0, 0>(out); row<8, 0>(out); row<16, 0>(out); row<24, 0>(out);
  row<8, 400>(out); row<16, 400>(out); row<24, 400>(out); row<16, 1200>(out); row<24, 1200>(out);
  return 0;
}
