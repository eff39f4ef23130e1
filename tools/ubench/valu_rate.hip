// Microbenchmark: issue cost of the VALU instructions the MLP engine's epilogues are made of (developer tool).
// 512-thread workgroups, one per CU, two waves per SIMD, 8 independent dependency chains per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
  float v[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = f2{v[i], v[i] + 1.f}; }
  const float c = 1.0001f;
  const f2 pc = {1.0001f, 0.9999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 1) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        REP8(X)
#undef X
      } else if (MODE == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        REP8(X)
#undef X
      } else if (MODE == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
        REP8(X)
#undef X
      } else if (MODE == 4) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (MODE == 5) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 6) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (MODE == 7) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 8) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 9) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (MODE == 10) {
#define X(i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (MODE == 11) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 13) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[8:9]" : "+v"(v[i]) : "v"(c) : "s8", "s9");
        REP8(X)
#undef X
      } else if (MODE == 14) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc");
        REP8(X)
#undef X
      } else if (MODE == 15) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(c) : "vcc");
        REP8(X)
#undef X
      } else if (MODE == 16) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 17) {
#define X(i) asm volatile("v_cmp_gt_f32 s[8:9], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[8:9]" : "+v"(v[i]) : "v"(c) : "s8", "s9");
        REP8(X)
#undef X
      } else if (MODE == 18) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 19) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (MODE == 20) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (MODE == 12) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // two waves per SIMD, 64 instructions per iteration each
  printf("%-22s %.3f ms   %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (2.0 * 64 * iters));
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;
  run<0>("v_fma_f32", out, iters);
  run<12>("v_add_f32", out, iters);
  run<1>("v_pk_add_f32", out, iters);
  run<2>("v_pk_mul_f32", out, iters);
  run<3>("v_pk_fma_f32", out, iters);
  run<4>("v_exp_f32", out, iters);
  run<5>("v_cvt_pk_bf16_f32", out, iters);
  run<6>("v_add_f32_dpp", out, iters);
  run<9>("v_mov_b32_dpp", out, iters);
  run<7>("v_med3_f32", out, iters);
  run<8>("v_cndmask_b32", out, iters);
  run<13>("v_cndmask_e64 sgpr", out, iters);
  run<14>("v_cmp+v_cndmask vcc (2)", out, iters);
  run<17>("v_cmp+v_cndmask sgpr (2)", out, iters);
  run<15>("v_cmp_gt_f32 vcc", out, iters);
  run<16>("v_max_f32", out, iters);
  run<18>("v_mul_f32", out, iters);
  run<19>("v_lshlrev_b32", out, iters);
  run<20>("v_sub_f32", out, iters);
  run<10>("v_and_b32", out, iters);
  run<11>("v_perm_b32", out, iters);
  return 0;
}
