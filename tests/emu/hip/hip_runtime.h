// Wave-level CPU emulator of the HIP subset used by dynibar_amd/csrc  --  TEST INFRASTRUCTURE ONLY.
//
// The product kernels are written for gfx950 only and contain no host/device dual paths.  To debug their
// indexing (MFMA fragment maps, LDS layouts, cross-lane reductions) without a GPU in the build container,
// tests/emu builds the *same, unmodified* csrc sources with the host clang++ against this fake
// <hip/hip_runtime.h>.  One OS thread per GPU thread, blocks run one at a time, __syncthreads() and the
// wave-level builtins (shuffles, ballot, MFMA) rendezvous on barriers.  Nothing in dynibar_amd/ can load the
// resulting library: it is built into tests/emu/_build and opened only by tests/emu/*.py.
#pragma once
#include <pthread.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define DYN_PIN(x) ((void)0)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ /* all LDS is the single dynamic array `dyn_smem` (see csrc/dyn_device.h) */
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t*) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
constexpr int hipMemcpyHostToDevice = 1;
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; int clockRate; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 7; p->clockRate = 1000000; return hipSuccess; }
static inline hipError_t hipStreamGetDevice(hipStream_t, int* d) { *d = 0; return hipSuccess; }
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }  // 7 "CUs" x 1: persistent kernels walk several tiles per workgroup
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;

namespace emu {
constexpr int WAVE = 64;
constexpr int MAX_THREADS = 1024;
constexpr size_t LDS_BYTES = 160 * 1024;

struct WaveCtx {
  pthread_barrier_t bar;
  int n;
  alignas(64) uint64_t scratch[2][WAVE][4];  // double-buffered exchange slots (four 64-bit words per lane)
  int phase = 0;
};
struct BlockCtx {
  pthread_barrier_t bar;
  std::vector<WaveCtx> waves;
};
struct ThreadCtx {
  dim3 tid, bid, bdim, gdim;
  int lane, wave;
  WaveCtx* w;
  BlockCtx* b;
  int phase = 0;
};
extern thread_local ThreadCtx tc;
extern "C" char* emu_lds_base();
}  // namespace emu

#define threadIdx (emu::tc.tid)
#define blockIdx (emu::tc.bid)
#define blockDim (emu::tc.bdim)
#define gridDim (emu::tc.gdim)
constexpr int warpSize = 64;

alignas(16) extern float4 dyn_smem[];

static inline void __syncthreads() { pthread_barrier_wait(&emu::tc.b->bar); }
static inline void __builtin_amdgcn_s_barrier() { pthread_barrier_wait(&emu::tc.b->bar); }
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __builtin_amdgcn_wave_barrier() { pthread_barrier_wait(&emu::tc.w->bar); }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}

namespace emu {
// publish two 64-bit words per lane, rendezvous, return the slot array of this exchange
static inline uint64_t (*exchange(uint64_t a, uint64_t b, uint64_t c = 0, uint64_t d = 0))[4] {
  ThreadCtx& t = tc;
  WaveCtx* w = t.w;
  int ph = t.phase;
  t.phase ^= 1;
  w->scratch[ph][t.lane][0] = a;
  w->scratch[ph][t.lane][1] = b;
  w->scratch[ph][t.lane][2] = c;
  w->scratch[ph][t.lane][3] = d;
  pthread_barrier_wait(&w->bar);
  return w->scratch[ph];
}
template <class T> static inline uint64_t bits(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8); memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T unbits(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }
}  // namespace emu

template <class T> static inline T __shfl(T v, int src, int width = 64) {
  auto s = emu::exchange(emu::bits(v), 0);
  int lane = emu::tc.lane;
  int base = lane & ~(width - 1);
  return emu::unbits<T>(s[base + (src & (width - 1))][0]);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  auto s = emu::exchange(emu::bits(v), 0);
  int lane = emu::tc.lane;
  int src = lane ^ mask;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return emu::unbits<T>(s[src][0]);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  auto s = emu::exchange(emu::bits(v), 0);
  int lane = emu::tc.lane;
  int src = lane + (int)d;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return emu::unbits<T>(s[src][0]);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  auto s = emu::exchange(emu::bits(v), 0);
  int lane = emu::tc.lane;
  int src = lane - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return emu::unbits<T>(s[src][0]);
}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline unsigned long long __ballot(int pred) {
  auto s = emu::exchange(pred ? 1 : 0, 0);
  unsigned long long m = 0;
  for (int i = 0; i < emu::tc.w->n; ++i) m |= (unsigned long long)(s[i][0] & 1) << i;
  return m;
}
static inline int __all(int pred) { return __ballot(pred) == __ballot(1); }
static inline int __any(int pred) { return __ballot(pred) != 0; }
template <class T> static inline T __builtin_amdgcn_readfirstlane(T v) {
  auto s = emu::exchange(emu::bits(v), 0);
  return emu::unbits<T>(s[0][0]);
}
static inline int __builtin_amdgcn_readlane(int v, int l) { return __shfl(v, l); }
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {a (bytes 7..4), b (bytes 3..0)} (selectors 0..7 only)
static inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {
  const unsigned long long v = ((unsigned long long)a << 32) | b;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xffu) << (8 * i);
  return r;
}

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
// Exact f32 fma chain in k order (cdna_hip_programming.md section 3).
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
  auto s = emu::exchange(emu::bits(a), emu::bits(b));
  int l = emu::tc.lane, j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(emu::unbits<float>(s[row + 32 * k][0]), emu::unbits<float>(s[j + 32 * k][1]), acc);
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r.
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  auto s = emu::exchange(emu::bits(a), emu::bits(b));
  int l = emu::tc.lane, j = l & 15, g = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = g * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(emu::unbits<float>(s[row + 16 * k][0]), emu::unbits<float>(s[j + 16 * k][1]), acc);
    c[r] = acc;
  }
  return c;
}


#define __expf(x) expf(x)
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline void __builtin_amdgcn_s_waitcnt(int) {}
// global_load_lds: the LDS destination is the wave-uniform base + lane * size (cdna_hip_programming.md section 5)
static inline void __builtin_amdgcn_global_load_lds(const __attribute__((address_space(1))) void* g, __attribute__((address_space(3))) void* l,
                                                    int size, int offset, int) {
  const char* src = (const char*)(uintptr_t)g + offset;
  char* dst = (char*)(uintptr_t)l + offset + emu::tc.lane * size;
  memcpy(dst, src, size);
}

static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
// v_mfma_f32_32x32x16_bf16: A[i = l&31][k = 8*(l>>5) + e], B[k = 8*(l>>5) + e][j = l&31], e = 0..7; D as the 32x32 f32 form.
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
  uint64_t aw[2], bw[2];
  memcpy(aw, &a, 16);
  memcpy(bw, &b, 16);
  auto s = emu::exchange(aw[0], aw[1], bw[0], bw[1]);
  auto elem = [&](int lane, int which, int e) -> float {  // which: 0 = a, 1 = b
    uint64_t w = s[lane][which * 2 + (e >> 2)];
    unsigned bits = (unsigned)((w >> (16 * (e & 3))) & 0xffffu) << 16;
    float f; memcpy(&f, &bits, 4); return f;
  };
  int l = emu::tc.lane, j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) acc = fmaf(elem(row + 32 * (k >> 3), 0, k & 7), elem(j + 32 * (k >> 3), 1, k & 7), acc);
    c[r] = acc;
  }
  return c;
}

// v_mfma_f32_32x32x16_f16: same operand layout as the bf16 form, IEEE half elements
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 emu_f16x2 __attribute__((ext_vector_type(2)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c, int, int, int) {
  uint64_t aw[2], bw[2];
  memcpy(aw, &a, 16);
  memcpy(bw, &b, 16);
  auto s = emu::exchange(aw[0], aw[1], bw[0], bw[1]);
  auto elem = [&](int lane, int which, int e) -> float {
    uint64_t w = s[lane][which * 2 + (e >> 2)];
    unsigned short bits = (unsigned short)((w >> (16 * (e & 3))) & 0xffffu);
    _Float16 h; memcpy(&h, &bits, 2); return (float)h;
  };
  int l = emu::tc.lane, j = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 16; ++k) acc = fmaf(elem(row + 32 * (k >> 3), 0, k & 7), elem(j + 32 * (k >> 3), 1, k & 7), acc);
    c[r] = acc;
  }
  return c;
}
// v_cvt_pkrtz_f16_f32: two fp32 -> packed halves, rounded toward zero (finite overflow saturates at 65504)
static inline _Float16 emu_f16_rtz(float x) {
  _Float16 h = (_Float16)x;                       // round to nearest even
  if (__builtin_isnan(x)) return h;
  if (__builtin_isinf((float)h) && !__builtin_isinf(x)) { h = (_Float16)65504.0f; return x < 0 ? -h : h; }
  if (__builtin_fabsf((float)h) > __builtin_fabsf(x)) {  // step one ulp toward zero
    unsigned short b; memcpy(&b, &h, 2); b -= 1; memcpy(&h, &b, 2);
  }
  return h;
}
static inline emu_f16x2 __builtin_amdgcn_cvt_pkrtz(float a, float b) {
  emu_f16x2 r; r[0] = emu_f16_rtz(a); r[1] = emu_f16_rtz(b); return r;
}

// v_mov_b32_dpp for the lane selects used by csrc: quad_perm (ctrl < 0x100), row_half_mirror (0x141), row_mirror (0x140)
static inline int __builtin_amdgcn_mov_dpp(int v, int ctrl, int, int, bool) {
  auto s = emu::exchange(emu::bits(v), 0);
  int lane = emu::tc.lane, src;
  if (ctrl < 0x100) src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl == 0x141) src = (lane & ~7) | (7 - (lane & 7));
  else if (ctrl == 0x140) src = (lane & ~15) | (15 - (lane & 15));
  else { fprintf(stderr, "emu: unsupported dpp ctrl 0x%x\n", ctrl); abort(); }
  return emu::unbits<int>(s[src][0]);
}
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline void sincosf_(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
static inline float atomicAdd(float* p, float v) {
  float old = *p, nw;
  do { nw = old + v; } while (!__atomic_compare_exchange(p, &old, &nw, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
  return old;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline double atomicAdd(double* p, double v) {
  double old = *p, want;
  do { want = old + v; } while (!__atomic_compare_exchange(p, &old, &want, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
  return old;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}

namespace emu {
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
template <class K, class... Args>
static inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  run_grid(grid, block, shmem, [&]() { kernel(args...); });
}
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu::launch(kernel, grid, block, shmem, stream, __VA_ARGS__)
