// Microbenchmark: how much of a VALU instruction kind's issue time hides under the matrix pipe?  Unit per wave: 3 MFMAs (3 accumulators),
// each followed by K3 instructions of one kind on 8 independent chains; two waves per SIMD (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(ACC) if (WITH_MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
template <int KIND>
__device__ __forceinline__ void op(float& v, float c) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c));
  if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(c));
  if (KIND == 2) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(v));
  if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
  if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(c));
  if (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(c) : "vcc");
  if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v));
  if (KIND == 7) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v));
  if (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(v) : "v"(c));
  if (KIND == 9) asm volatile("v_add_f32 %0, -1.0, %0" : "+v"(v));
  if (KIND == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(c));
}
template <int KIND, int K3, int WITH_MFMA>
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
      MF(acc0)
#pragma unroll
      for (int i = 0; i < K3; ++i) op<KIND>(v[i & 7], c);
      MF(acc1)
#pragma unroll
      for (int i = 0; i < K3; ++i) op<KIND>(v[(i + 3) & 7], c);
      MF(acc2)
#pragma unroll
      for (int i = 0; i < K3; ++i) op<KIND>(v[(i + 6) & 7], c);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND, int K3, int WITH_MFMA>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, K3, WITH_MFMA>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<KIND, K3, WITH_MFMA>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / (8.0 * iters);
}
template <int KIND>
void row(const char* name, float* out, int iters) {
  const float valu = run<KIND, 4, 0>(out, iters), both = run<KIND, 4, 1>(out, iters), mf = run<KIND, 0, 1>(out, iters);
  printf("%-22s 12 instr/wave alone %6.1f ns   6 MFMAs alone %6.1f   together %6.1f   hidden %3.0f %%\n", name, valu, mf, both,
         100.0 * (valu + mf - both) / (valu < mf ? valu : mf));
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  row<0>("v_fma_f32", out, iters);
  row<1>("v_add_f32 (v,v)", out, iters);
  row<9>("v_add_f32 (-1.0,v)", out, iters);
  row<2>("v_mul_f32 (lit,v)", out, iters);
  row<3>("v_exp_f32", out, iters);
  row<4>("v_cvt_pk_bf16_f32", out, iters);
  row<5>("v_cmp + v_cndmask", out, iters);
  row<6>("v_lshlrev_b32", out, iters);
  row<7>("v_and_b32 (lit)", out, iters);
  row<8>("v_med3_f32", out, iters);
  row<10>("v_max_f32", out, iters);
  return 0;
}
