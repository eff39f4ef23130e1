// Microbenchmark: does the register file of the MFMA ACCUMULATOR (ArchVGPR vs AccVGPR) change how much VALU work hides under the matrix pipe?
// Unit per wave: 3 dependent MFMAs on one accumulator (the split engine's triple) + K fillers (v_fma_f32 on independent chains), two waves per
// SIMD, for K = 0 .. 24.  Prints ns per unit for both files (developer tool).   hipcc --offload-arch=gfx950 -O3 mfma_acc_file.hip -o mfma_acc_file
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int AGPR, int K, int SPREAD>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = threadIdx.x * 1e-3f + r + t;
  u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float c = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // SPREAD = 0: fillers first, then the three MFMAs back to back (the engine's form); 1: a third of the fillers after each MFMA
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (SPREAD == 0 && m == 0) {
#pragma unroll
          for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(c));
        }
        if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
        if (SPREAD == 1) {
#pragma unroll
          for (int i = 0; i < K / 3; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i + 3 * m) & 7]) : "v"(c));
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int AGPR, int K, int SPREAD>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<AGPR, K, SPREAD>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<AGPR, K, SPREAD>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (4.0 * iters);  // ns per unit (3 MFMAs + K fillers) per wave; two waves share a SIMD
}
template <int K>
void row(float* out, int iters) {
  printf("K = %2d fillers per 3 MFMAs:  acc in VGPR  %6.1f ns (spread %6.1f)   acc in AGPR  %6.1f ns (spread %6.1f)\n", K, run<0, K, 0>(out, iters), run<0, K, 1>(out, iters),
         run<1, K, 0>(out, iters), run<1, K, 1>(out, iters));
}
// the same 12 MFMAs per loop body, term-major: consecutive MFMAs write DIFFERENT accumulators (no dependent pair back to back)
template <int K>
__global__ void __launch_bounds__(512, 2) k_indep(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = threadIdx.x * 1e-3f + r + t;
  u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float c = 1.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < K / 3; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i + 3 * m) & 7]) : "v"(c));
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int K>
float run_indep(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_indep<K>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_indep<K>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (4.0 * iters);
}
// effective shader clock: s_memtime ticks at 100 MHz? use wall-clock of a pure-VALU loop of known cycle count instead
__global__ void k_clock(float* out, int iters) {
  float v = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v));
  }
  out[blockIdx.x * 64 + threadIdx.x] = v;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;
  {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_clock, dim3(256), dim3(64), 0, 0, out, 20000);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_clock, dim3(256), dim3(64), 0, 0, out, 20000);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("dependent v_fma chain, one wave per CU: %.2f ns per instruction (4 cycles dependent-issue => %.2f GHz)\n", ms * 1e6 / (20000.0 * 64), 4.0 / (ms * 1e6 / (20000.0 * 64)));
  }
  printf("term-major (independent consecutive MFMAs): K=0 %6.1f ns   K=12 %6.1f   K=18 %6.1f   K=24 %6.1f   K=36 %6.1f\n", run_indep<0>(out, iters), run_indep<12>(out, iters),
         run_indep<18>(out, iters), run_indep<24>(out, iters), run_indep<36>(out, iters));
  row<0>(out, iters); row<6>(out, iters); row<12>(out, iters); row<18>(out, iters); row<24>(out, iters); row<36>(out, iters);
  printf("(3 MFMAs at full rate: 2 waves x 96 cycles = 192 cycles per unit pair = 80 ns per unit at 2.4 GHz; a v_fma_f32 alone: 2 cycles)\n");
  return 0;
}
