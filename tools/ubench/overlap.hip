// Microbenchmark: do an MFMA-bound wave and a VALU-bound wave on the same SIMD overlap?  (developer tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA, 4-7 VALU; 3: every wave alternates MFMA block / VALU block
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 1e-3f; acc1[r] = r; }
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float v0 = threadIdx.x, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
  const bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 5) && wave < 4) || MODE == 3 || (MODE == 6 && (wave & 1) == 0);
  const bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 4) && wave >= 4) || MODE == 3 || (MODE == 6 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
      }
    }
    if (MODE == 7 || MODE == 8) {  // fine interleave inside every wave: 1 MFMA, then 16 (MODE 7) / 8 (MODE 8) fmas, 16 times per iteration
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < (MODE == 7 ? 4 : 2); ++q) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < (MODE == 7 ? 4 : 2); ++q) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
      }
    }
    if (do_valu) {
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1);
      }
    }
  }
  float s = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  printf("all MFMA (16 per iter per wave)   : %.3f ms\n", run<0>(out, iters));
  printf("all VALU (256 fma per iter)       : %.3f ms\n", run<1>(out, iters));
  printf("half waves MFMA, half VALU        : %.3f ms\n", run<2>(out, iters));
  printf("every wave MFMA block + VALU block: %.3f ms\n", run<3>(out, iters));
  printf("waves 4-7 VALU, waves 0-3 idle    : %.3f ms\n", run<4>(out, iters));
  printf("waves 0-3 MFMA, waves 4-7 idle    : %.3f ms\n", run<5>(out, iters));
  printf("even waves MFMA, odd waves VALU   : %.3f ms\n", run<6>(out, iters));
  printf("per wave 16 x (MFMA + 16 fma)      : %.3f ms   (same work as 'MFMA block + VALU block')\n", run<7>(out, iters));
  printf("per wave 16 x (MFMA + 8 fma)       : %.3f ms   (half the VALU work)\n", run<8>(out, iters));
  return 0;
}
