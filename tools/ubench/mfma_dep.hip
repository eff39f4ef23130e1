// Microbenchmark: issue interval of v_mfma_f32_32x32x16_bf16 chains by accumulator reuse distance, one and two waves per SIMD
// (developer tool: decides the MFMA order of the engine in the one-wave-per-SIMD point kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
template <int DIST, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) k(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = threadIdx.x * 1e-3f + t;
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (DIST == 1) { MF(acc[0]) MF(acc[0]) MF(acc[0]) MF(acc[1]) MF(acc[1]) MF(acc[1]) MF(acc[2]) MF(acc[2]) MF(acc[2]) MF(acc[3]) MF(acc[3]) MF(acc[3]) }
      if (DIST == 2) { MF(acc[0]) MF(acc[1]) MF(acc[0]) MF(acc[1]) MF(acc[0]) MF(acc[1]) MF(acc[2]) MF(acc[3]) MF(acc[2]) MF(acc[3]) MF(acc[2]) MF(acc[3]) }
      if (DIST == 4) { MF(acc[0]) MF(acc[1]) MF(acc[2]) MF(acc[3]) MF(acc[0]) MF(acc[1]) MF(acc[2]) MF(acc[3]) MF(acc[0]) MF(acc[1]) MF(acc[2]) MF(acc[3]) }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int DIST, int THREADS>
void run(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<DIST, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<DIST, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%d wave(s)/SIMD, same accumulator every %d MFMAs: %6.2f ns per MFMA per SIMD\n", THREADS / 256, DIST, ms * 1e6 / (48.0 * iters * (THREADS / 256)));
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;
  run<1, 256>(out, iters); run<2, 256>(out, iters); run<4, 256>(out, iters);
  run<1, 512>(out, iters); run<2, 512>(out, iters); run<4, 512>(out, iters);
  return 0;
}
