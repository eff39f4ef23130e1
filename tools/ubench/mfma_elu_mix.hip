// Microbenchmark: which ingredient of the MLP engine's inner loop stops VALU work from overlapping the matrix pipe?
// Unit = 3 MFMAs (one accumulator) + the VALU work of one B-operand slice (2 values: ELU + bf16 split), two waves per SIMD.
// Variants add the real instruction kinds one at a time (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));

// VARIANT bits: 1 = v_exp, 2 = v_cmp + v_cndmask (vcc), 4 = cvt/shift/and split, 8 = two ds_read_b128 + wait, 16 = write the MFMA's B operand
template <int VARIANT>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
  __shared__ u32x4 lds[64 * 64];
  lds[threadIdx.x] = u32x4{threadIdx.x, 1, 2, 3};
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x * 1e-3f; acc1[r] = r; }
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  float x0 = threadIdx.x * 1e-3f - 0.2f, x1 = 0.3f - threadIdx.x * 1e-3f;
  float s0 = 0.f, s1 = 0.f;
  unsigned hi = 0, mid = 0;
  const u32x4* lp = lds + (threadIdx.x & 63);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      u32x4 l0, l1;
      if (VARIANT & 8) {
        asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024" : "=v"(l0), "=v"(l1) : "v"((unsigned)(size_t)lp + u * 2048));
      }
      float e0, e1;
      // ELU on two values
      asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %2\n v_mul_f32 %1, 0x3fb8aa3b, %3" : "=v"(e0), "=v"(e1) : "v"(x0), "v"(x1));
      if (VARIANT & 1) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(e0), "+v"(e1));
      else asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1" : "+v"(e0), "+v"(e1));
      MF(acc0)
      asm volatile("v_add_f32 %0, -1.0, %0\n v_add_f32 %1, -1.0, %1" : "+v"(e0), "+v"(e1));
      if (VARIANT & 2)
        asm volatile("v_cmp_lt_f32 vcc, 0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_lt_f32 vcc, 0, %3\n v_cndmask_b32 %1, %1, %3, vcc"
                     : "+v"(e0), "+v"(e1) : "v"(x0), "v"(x1) : "vcc");
      else
        asm volatile("v_med3_f32 %0, %0, %2, 0\n v_med3_f32 %1, %1, %3, 0" : "+v"(e0), "+v"(e1) : "v"(x0), "v"(x1));
      MF(acc0)
      if (VARIANT & 4) {
        float r0, r1;
        asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n v_lshlrev_b32 %1, 16, %0\n v_and_b32 %2, 0xffff0000, %0" : "=&v"(hi), "=&v"(r0), "=&v"(r1) : "v"(e0), "v"(e1));
        asm volatile("v_sub_f32 %0, %2, %0\n v_sub_f32 %1, %3, %1" : "+v"(r0), "+v"(r1) : "v"(e0), "v"(e1));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(mid) : "v"(r0), "v"(r1));
      } else {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(e0));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(e1));
      }
      if (VARIANT & 8) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a = l0; 
        b[1] = l1[1];
      }
      if (VARIANT & 16) { b[0] = hi; b[2] = mid; }
      MF(acc0)
      x0 += 1e-3f; x1 -= 1e-3f;
    }
  }
  float s = s0 + s1 + hi + mid;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int VARIANT>
void run(const char* name, float* out, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VARIANT>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<VARIANT>), dim3(256), dim3(512), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %7.1f ns per unit-pair (6 MFMAs alone ~95)\n", name, ms * 1e6 / (8.0 * iters));
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  run<0>("plain VALU stand-ins (mul/add/med3)", out, iters);
  run<1>("+ v_exp", out, iters);
  run<2>("+ v_cmp/v_cndmask via vcc", out, iters);
  run<4>("+ bf16 split (cvt_pk, shift, and, sub, cvt_pk)", out, iters);
  run<8>("+ 2 ds_read_b128 + lgkmcnt(0) per unit", out, iters);
  run<7>("exp + cmp/cndmask + split", out, iters);
  run<15>("exp + cmp/cndmask + split + ds_read", out, iters);
  run<31>("all + split result feeds the MFMA's B operand", out, iters);
  run<23>("all but ds_read, split result feeds B", out, iters);
  return 0;
}
