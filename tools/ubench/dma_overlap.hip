// Microbenchmark: does an HBM -> LDS stream (LDS-DMA, global_load_lds_dwordx4) overlap with matrix / vector / LDS work of the same
// workgroups?  Persistent workgroups (2 per CU, 256 threads, two 32 KiB slots): per step wait for the request of the step before,
// barrier, request the next 32 KiB, then "compute".  (developer tool; hipcc --offload-arch=gfx950 -O3 dma_overlap.hip -o dma_overlap)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) float smem[];

__device__ __forceinline__ void dma16(const float* gp, float* wave_base) {
  const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)wave_base);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(off) : "m0", "memory");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void barrier_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- second experiment: what keeps the stream fast?  QH quads per thread and step from the workgroup's own HBM stream, QS quads from a
// small region every workgroup shares (a weight tile: L2 hits), NSLOT ring slots (request distance NSLOT - 1), compute as above ----
template <int QH, int QS, int NSLOT, int NM, int NV, int NL>
__global__ void __launch_bounds__(256) k2(const float* __restrict__ src, long floats_per_wg, const float* __restrict__ shared_src, float* out, int steps) {
  const int tid = threadIdx.x, wave = tid >> 6;
  constexpr int SLOT = (QH + QS) * 1024;  // floats
  const float* base = src + (long)blockIdx.x * floats_per_wg;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = tid * 1e-3f + r;
  u32x4 a = {0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float v0 = tid, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
  f32x4 lsum = {0.f, 0.f, 0.f, 0.f};
  auto request = [&](int step) __attribute__((always_inline)) {
    float* dst = smem + (step % NSLOT) * SLOT;
    const float* g = base + (long)step * (QH * 1024);
#pragma unroll
    for (int j = 0; j < QH; ++j) dma16(g + 4 * (j * 256 + tid), dst + 4 * (j * 256 + wave * 64));
    const float* gs = shared_src + (step & 7) * (QS * 1024);
#pragma unroll
    for (int j = 0; j < QS; ++j) dma16(gs + 4 * (j * 256 + tid), dst + QH * 1024 + 4 * (j * 256 + wave * 64));
  };
  for (int s0 = 0; s0 < NSLOT - 1; ++s0) request(s0);
  for (int s = 0; s < steps; ++s) {
    if (NSLOT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * (QH + QS)) : "memory");
    barrier_lds();
    if (s + NSLOT - 1 < steps) request(s + NSLOT - 1);
    else {  // keep the count of newer operations constant
#pragma unroll
      for (int j = 0; j < QH + QS; ++j) dma16(base + 4 * tid, smem + ((s + NSLOT - 1) % NSLOT) * SLOT + 4 * (j * 256 + wave * 64));
    }
    const float* slot = smem + (s % NSLOT) * SLOT;
#pragma unroll
    for (int i = 0; i < NL; ++i) lsum += *reinterpret_cast<const f32x4*>(slot + 4 * ((tid + 64 * i) & (SLOT / 4 - 1)));
#pragma unroll
    for (int i = 0; i < NV; ++i) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
#pragma unroll
    for (int i = 0; i < NM; ++i)
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i & 3], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = v0 + v1 + v2 + v3 + lsum[0] + lsum[1] + lsum[2] + lsum[3];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 16; ++q) r += acc[i][q];
  out[blockIdx.x * 256 + tid] = r;
}
template <int QH, int QS, int NSLOT, int NM, int NV, int NL>
void run2(const float* src, const float* shared_src, float* out, long total_floats, int wgs, const char* what) {
  const int steps = (int)(total_floats / wgs / (QH * 1024));
  const long per_wg = (long)steps * QH * 1024;
  const int lds = NSLOT * (QH + QS) * 4096;
  hipFuncSetAttribute((const void*)k2<QH, QS, NSLOT, NM, NV, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<QH, QS, NSLOT, NM, NV, NL>), dim3(wgs), dim3(256), lds, 0, src, per_wg, shared_src, out, steps);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k2<QH, QS, NSLOT, NM, NV, NL>), dim3(wgs), dim3(256), lds, 0, src, per_wg, shared_src, out, steps);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double gb = (double)wgs * per_wg * 4 / 1e9;
  printf("%-78s %4d WGs %5d steps %8.1f us  HBM stream %5.2f TB/s  (LDS %3d KiB)\n", what, wgs, steps, ms * 1e3, gb / ms, lds / 1024);
}

// LOAD: 0 none, 1 LDS-DMA, 2 loads into registers (consumed after the wait);  NM MFMAs, NV x 4 fmas, NL ds_read_b128 per step
template <int LOAD, int NM, int NV, int NL>
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ src, long floats_per_wg, float* out, int steps) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const float* base = src + (long)blockIdx.x * floats_per_wg;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = tid * 1e-3f + r;
  u32x4 a = {0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float v0 = tid, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
  f32x4 regs[8];
  f32x4 lsum = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < 8; ++j) regs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto request = [&](int step) __attribute__((always_inline)) {
    float* dst = smem + (step & 1) * 8192;
    const float* g = base + (long)step * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (LOAD == 1) dma16(g + 4 * (j * 256 + tid), dst + 4 * (j * 256 + wave * 64));
      if (LOAD == 2) regs[j] = *reinterpret_cast<const f32x4*>(g + 4 * (j * 256 + tid));
    }
  };
  request(0);
  for (int s = 0; s < steps; ++s) {
    if (LOAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (LOAD == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) lsum += regs[j];
    }
    barrier_lds();
    if (s + 1 < steps) request(s + 1);
    const float* slot = smem + (s & 1) * 8192;
#pragma unroll
    for (int i = 0; i < NL; ++i) lsum += *reinterpret_cast<const f32x4*>(slot + 4 * ((tid + 64 * i) & 2047));
#pragma unroll
    for (int i = 0; i < NV; ++i) { v0 = fmaf(v0, v1, v2); v1 = fmaf(v1, v2, v3); v2 = fmaf(v2, v3, v0); v3 = fmaf(v3, v0, v1); }
#pragma unroll
    for (int i = 0; i < NM; ++i)
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i & 3], 0, 0, 0);
  }
  float r = v0 + v1 + v2 + v3 + lsum[0] + lsum[1] + lsum[2] + lsum[3];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 16; ++q) r += acc[i][q];
  out[blockIdx.x * 256 + tid] = r;
}

template <int LOAD, int NM, int NV, int NL>
float run(const float* src, long per_wg, float* out, int steps, int wgs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<LOAD, NM, NV, NL>), dim3(wgs), dim3(256), 65536, 0, src, per_wg, out, steps);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<LOAD, NM, NV, NL>), dim3(wgs), dim3(256), 65536, 0, src, per_wg, out, steps);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
int main() {
  const int wgs = 512, steps = 180;
  const long per_wg = (long)steps * 8192;
  float *src, *out;
  hipMalloc(&src, wgs * per_wg * 4); hipMalloc(&out, wgs * 256 * 4);
  hipMemset(src, 0, wgs * per_wg * 4);
  const double gb = (double)wgs * per_wg * 4 / 1e9;
  printf("%d workgroups x %d steps x 32 KiB = %.2f GB\n", wgs, steps, gb);
#define ROW(L, M, V, D, what) { float us = run<L, M, V, D>(src, per_wg, out, steps, wgs); if (L) printf("%-62s %8.1f us  %5.2f TB/s\n", what, us, gb / us * 1e-3); else printf("%-62s %8.1f us\n", what, us); }
  ROW(1, 0, 0, 0, "LDS-DMA only")
  ROW(2, 0, 0, 0, "register loads only")
  ROW(0, 24, 0, 0, "24 MFMA per step, no loads")
  ROW(1, 24, 0, 0, "24 MFMA per step + LDS-DMA")
  ROW(2, 24, 0, 0, "24 MFMA per step + register loads")
  ROW(0, 0, 48, 0, "192 fma per step, no loads")
  ROW(1, 0, 48, 0, "192 fma per step + LDS-DMA")
  ROW(0, 0, 0, 16, "16 ds_read_b128 per step, no loads")
  ROW(1, 0, 0, 16, "16 ds_read_b128 per step + LDS-DMA")
  ROW(0, 24, 48, 16, "24 MFMA + 192 fma + 16 ds_read, no loads")
  ROW(1, 24, 48, 16, "24 MFMA + 192 fma + 16 ds_read + LDS-DMA")
  ROW(2, 24, 48, 16, "24 MFMA + 192 fma + 16 ds_read + register loads")
  ROW(0, 48, 0, 0, "48 MFMA per step, no loads")
  ROW(1, 48, 0, 0, "48 MFMA per step + LDS-DMA")
  const long total = (long)wgs * per_wg;
  float* sh; hipMalloc(&sh, 8 * 8 * 4096); hipMemset(sh, 0, 8 * 8 * 4096);
  printf("-- compute per step = 24 MFMA + 192 fma + 16 ds_read_b128 unless noted --\n");
  run2<8, 0, 2, 24, 48, 16>(src, sh, out, total, 512, "8 HBM quads / step, 2 slots (the first experiment's shape)");
  run2<4, 0, 2, 24, 48, 16>(src, sh, out, total, 512, "4 HBM quads / step, 2 slots (16 KiB requests)");
  run2<4, 4, 2, 24, 48, 16>(src, sh, out, total, 512, "4 HBM + 4 shared quads / step, 2 slots (the GEMM's forward shape)");
  run2<4, 4, 2, 0, 0, 0>(src, sh, out, total, 512, "4 HBM + 4 shared quads / step, 2 slots, no compute");
  run2<4, 0, 4, 24, 48, 16>(src, sh, out, total, 512, "4 HBM quads / step, 4 slots");
  run2<4, 4, 3, 24, 48, 16>(src, sh, out, total, 256, "4 HBM + 4 shared quads / step, 3 slots, ONE workgroup per CU");
  run2<4, 4, 4, 24, 48, 16>(src, sh, out, total, 256, "4 HBM + 4 shared quads / step, 4 slots, ONE workgroup per CU");
  run2<4, 2, 3, 24, 48, 16>(src, sh, out, total, 512, "4 HBM + 2 shared quads / step, 3 slots (72 KiB: two workgroups)");
  run2<2, 2, 4, 12, 24, 8>(src, sh, out, total, 512, "2 HBM + 2 shared quads / step, 4 slots, half the compute per step (BK = 16)");
  run2<2, 2, 5, 12, 24, 8>(src, sh, out, total, 512, "2 HBM + 2 shared quads / step, 5 slots, half the compute per step (BK = 16)");
  run2<4, 4, 2, 24, 24, 16>(src, sh, out, total, 512, "4 HBM + 4 shared quads / step, 2 slots, HALF the fma work (24 MFMA + 96 fma + 16 ds_read)");
  run2<4, 4, 2, 24, 0, 16>(src, sh, out, total, 512, "4 HBM + 4 shared quads / step, 2 slots, NO fma work (24 MFMA + 16 ds_read)");
  run2<4, 4, 2, 0, 48, 16>(src, sh, out, total, 512, "4 HBM + 4 shared quads / step, 2 slots, NO MFMA (192 fma + 16 ds_read)");
  run2<2, 2, 2, 12, 24, 8>(src, sh, out, total, 768, "2 HBM + 2 shared quads / step, 2 slots, BK = 16, THREE workgroups per CU");
  run2<2, 2, 2, 12, 24, 8>(src, sh, out, total, 1024, "2 HBM + 2 shared quads / step, 2 slots, BK = 16, FOUR workgroups per CU");
  run2<2, 2, 3, 12, 24, 8>(src, sh, out, total, 768, "2 HBM + 2 shared quads / step, 3 slots, BK = 16, THREE workgroups per CU");
  run2<4, 4, 2, 24, 48, 16>(src, sh, out, total, 768, "4 HBM + 4 shared quads / step, 2 slots, BK = 32, three workgroups requested (LDS allows two)");
  return 0;
}
