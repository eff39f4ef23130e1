// Training kernels (first slice): forward-with-saved-activations and backward of the static branch, i.e. the
// step the reference's static bootstrap stage trains on (train.py:116-199: loss on ret['outputs_coarse_st']['rgb'], gradients to
// DynibarStatic's parameters and to the static feature maps).
//
// Design, as opposed to the inference kernels of dyn_nets.hip (register-resident chains, nothing saved): a training step needs every
// layer's output again in the backward pass, so the step is a sequence of
//   * ONE GEMM entry on the matrix pipe (dyn_train_gemm) used for forward (Y = act(diag(s) X W^T + b + P[row / V])), data gradients
//     (dX = dZ W) and weight gradients (dW += dZ^T X, split over the rows with fp32 atomics), in two kernel forms: k_train_gemm (tiles
//     through registers; the forward products) and k_train_gemm_ring (persistent workgroups, operand tiles by LDS-DMA: the backward
//     products -- see the comment in front of it), and
//   * small HBM-bound row / per-point kernels for everything between the Linear layers (Fourier features, pooling over views,
//     sigmoids, softmaxes, ray attention, LayerNorm, compositing) with their hand-derived backward forms,
// over row-major fp32 activation matrices that stay in HBM between kernels (the full iteration at 3072 rays x 64 samples x 10 + 10 + 15
// views peaks at 46 GB: sized for the 288 GB of an MI355X; the host side can recompute the 256-wide hidden layers instead, 37 GB).
//
// Arithmetic of the GEMM: fp32 in, fp32 accumulate; every fp32 operand is split into two IEEE half parts x = hi + mid (22 mantissa
// bits, like the inference engine) and the three partial products hi.hi + hi.mid + mid.hi run on v_mfma_f32_32x32x16_f16: fp32-class
// products at 2500/3 = 833 TFLOP/s peak.  Halves have 5 exponent bits, and gradients live at 1e-3 .. 1e-9: the gradient operand of the
// backward GEMMs is therefore multiplied, on its way into the split, by a power of two that brings the tensor's largest magnitude to
// 2^14 (the magnitude is a by-product of the kernel that produced the tensor; the result is scaled back in the epilogue; exact).
// The kernel is issue-bound on MFMAs + conversions, so two parts / three products instead of three bf16 parts / six
// products halve both.  The split happens once per element when a tile is written to LDS, not per use.
#include "dyn_device.h"
#include "dyn_host.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef unsigned tr_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short tr_u16;
typedef unsigned tr_u32x2 __attribute__((ext_vector_type(2)));

#define TG_BM 128
#define TG_BN 128
#define TG_BK 32
#ifndef TR_EXP
#define TR_EXP 0  /* developer decomposition builds (tools/gemm_decompose.sh): 1 no MFMA, 2 no loads inside the k loop, 4 no result stores, 8 no conversions */
#endif
#define TG_ROW 40                        // half elements per LDS row (32 used + 8 pad: 80-byte stride, conflict-free b128 reads)
#define TG_PART (TG_BM * TG_ROW)         // elements of one part image
#define TG_LDS_BYTES (2 * 2 * TG_PART * 2)  // (A | B) x two parts x 2 bytes = 40 KiB

typedef _Float16 tr_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tr_f16x8 __attribute__((ext_vector_type(8)));
// two fp32 values (times a power of two when SCALED) -> packed half pairs hi and mid: truncating pack-convert, then the exact residual
// written as a half by the mixed-precision fma (three instructions per pair, the inference engine's form; dyn_mlp.h)
template <bool SCALED>
__device__ __forceinline__ void tr_split2_pair(float x0, float x1, float scale, unsigned& h, unsigned& m) {
#if defined(__AMDGCN__)
  unsigned hh, mm;
  if (SCALED) {  // both parts straight from the mixed-precision fma: hi = half(x s), mid = half(x s - hi) -- two instructions per value
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(x0), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(x1), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(mm) : "v"(x0), "v"(scale), "v"(hh));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(mm) : "v"(x1), "v"(scale), "v"(hh));
  } else {       // truncating pack-convert for the pair, then the exact residuals: three instructions per pair
    hh = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(mm) : "v"(hh), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(mm) : "v"(hh), "v"(x1));
  }
  h = hh; m = mm;
#else  /* the wave-level emulator of tests/emu: the same two residuals, converted like the first part */
  if (SCALED) { x0 *= scale; x1 *= scale; }
  const auto hh = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  h = __builtin_bit_cast(unsigned, hh);
  m = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hh[0], x1 - (float)hh[1]));
#endif
}

struct TrOperand {
  const float* p;
  long rs, ks;   // element (r, k) at p[r * rs + k * ks]; exactly one of rs, ks is 1 (or both for a vector)
  int nrows;     // bound of r
  int aligned;   // base pointer 16-byte aligned
};

// global -> registers: the thread's 16 elements of a [128 rows x 32 k] operand tile (zero beyond the bounds).  Branch-free: every
// address is clamped into the operand and the out-of-range elements are zeroed by selects, so that all loads of a tile issue back
// to back (a loader with bounds branches waits for each load in turn: measured 10x slower on the 3 M-row layers).
// MODE 0: k-minor, rows 16-byte aligned and padded to a multiple of four floats (one dwordx4 per four k);
// MODE 1: k-minor, any alignment (four dword loads); MODE 2: k-major (rows contiguous; two dword loads per k for a row pair).
template <int MODE>
__device__ __forceinline__ void tr_load_tile(const TrOperand& o, int row0, int k0, int kend, float4 (&st)[4], int tid) {
  // RAW values only: the bounds masks are applied when the tile is converted (tr_mask_tile, a k-step or two later) -- a select right
  // here makes the compiler wait for every load where it is issued (measured: the loads of this kernel were effectively synchronous)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 v;
    if (MODE == 0 || MODE == 1) {
      const int r = row0 + (tid >> 3) + 32 * i, k = k0 + (tid & 7) * 4;
      const int rc = r < o.nrows ? r : o.nrows - 1;
      if (MODE == 0) {
        const int kc = k < kend ? k : 0;  // a row holds at least round_up4(kend) floats (checked by the host wrapper)
        v = *reinterpret_cast<const float4*>(o.p + (long)rc * o.rs + kc);
      } else {
        const float* g = o.p + (long)rc * o.rs;
        const int km = kend - 1;
        v.x = g[k < km ? k : km];
        v.y = g[k + 1 < km ? k + 1 : km];
        v.z = g[k + 2 < km ? k + 2 : km];
        v.w = g[k + 3 < km ? k + 3 : km];
      }
    } else {
      // the thread takes rows 2 rp, 2 rp + 1 of the eight k of its k-group; st[i] = (k = 2 i: r0, r1 | k = 2 i + 1: r0, r1)
      const int r = row0 + (tid & 63) * 2, kb = k0 + (tid >> 6) * 8 + 2 * i;
      const int rm = o.nrows - 1, km = kend - 1;
      const int r0 = r < rm ? r : rm, r1 = r + 1 < rm ? r + 1 : rm;
      const float* g0 = o.p + (long)(kb < km ? kb : km) * o.ks;
      const float* g1 = o.p + (long)(kb + 1 < km ? kb + 1 : km) * o.ks;
      v.x = g0[r0]; v.y = g0[r1]; v.z = g1[r0]; v.w = g1[r1];
    }
    st[i] = v;
  }
}
// zero the elements of a raw tile that lie beyond the operand's bounds (same index maps as tr_load_tile)
template <int MODE>
__device__ __forceinline__ void tr_mask_tile(const TrOperand& o, int row0, int k0, int kend, float4 (&st)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 v = st[i];
    if (MODE == 0 || MODE == 1) {
      const int r = row0 + (tid >> 3) + 32 * i, k = k0 + (tid & 7) * 4;
      const bool rok = r < o.nrows;
      v.x = (rok && k < kend) ? v.x : 0.f;
      v.y = (rok && k + 1 < kend) ? v.y : 0.f;
      v.z = (rok && k + 2 < kend) ? v.z : 0.f;
      v.w = (rok && k + 3 < kend) ? v.w : 0.f;
    } else {
      const int r = row0 + (tid & 63) * 2, kb = k0 + (tid >> 6) * 8 + 2 * i;
      v.x = (kb < kend && r < o.nrows) ? v.x : 0.f;
      v.y = (kb < kend && r + 1 < o.nrows) ? v.y : 0.f;
      v.z = (kb + 1 < kend && r < o.nrows) ? v.z : 0.f;
      v.w = (kb + 1 < kend && r + 1 < o.nrows) ? v.w : 0.f;
    }
    st[i] = v;
  }
}

// registers -> LDS part images [part][row][k] (k contiguous: what the MFMA operand reads want), splitting on the way
template <bool KMINOR, bool SCALED>
__device__ __forceinline__ void tr_store_tile(tr_u16* img, const float4 (&st)[4], float scale, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (KMINOR) {
      unsigned h0, m0, h1, m1;
      tr_split2_pair<SCALED>(st[i].x, st[i].y, scale, h0, m0);
      tr_split2_pair<SCALED>(st[i].z, st[i].w, scale, h1, m1);
      const int r = (tid >> 3) + 32 * i, k = (tid & 7) * 4;
      tr_u16* d = img + r * TG_ROW + k;
      *reinterpret_cast<tr_u32x2*>(d) = tr_u32x2{h0, h1};
      *reinterpret_cast<tr_u32x2*>(d + TG_PART) = tr_u32x2{m0, m1};
    }
  }
  if (!KMINOR) {
    // two rows x eight consecutive k per thread: one 16-byte write per (part, row)
    const int r = (tid & 63) * 2, k = (tid >> 6) * 8;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      tr_u32x4 ph, pm;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float a = q == 0 ? st[i].x : st[i].y, b = q == 0 ? st[i].z : st[i].w;
        unsigned th, tm;
        tr_split2_pair<SCALED>(a, b, scale, th, tm);
        ph[i] = th; pm[i] = tm;
      }
      tr_u16* d = img + (r + q) * TG_ROW + k;
      *reinterpret_cast<tr_u32x4*>(d) = ph;
      *reinterpret_cast<tr_u32x4*>(d + TG_PART) = pm;
    }
  }
}

struct TrGemmArgs {
  TrOperand a, b;  // a: rows = m, b: rows = n
  float* c;
  long ldc;
  int M, N, K;
  int k_chunk;           // k range per blockIdx.z / reduction chunk (multiple of TG_BK)
  int mt, nt, nz;        // ring form: row tiles, column tiles, reduction chunks (mt * nz units of nt tiles each)
  const float* bias;     // [N] or null
  const float* addend;   // [(M / add_div), ld_add] or null
  long ld_add;
  int add_div;
  int act;               // 0 none, 1 ELU
  int accumulate;        // 0 store, 1 c += result, 2 atomicAdd
  const float* a_absmax; // device: largest |a| (a gradient tensor: scaled into the half range), or null (operands of order one)
  const float* act_y;    // saved output of the layer whose activation derivative multiplies the result (data gradient), or null
  long ld_y;
  int act_y_kind;        // 1 ELU, 2 ReLU
  int act_y_vec;         // act_y rows 16-byte aligned and N a multiple of four: the tile is staged through LDS
  int c_vec;             // plain stores (accumulate 0) of 16-byte-aligned result rows, N a multiple of four: the tile leaves through LDS
  float* colsum_part;    // [row tiles, ld_part] per-workgroup column sums of the result (the bias gradient's partial sums), or null (needs c_vec)
  long ld_part;
  float* amax_part;      // [row tiles * column tiles] per-workgroup largest |result| (with colsum_part)
  const float* rowscale; // [M] or null: row m of the product is multiplied by rowscale[m] (before addend, bias, activation)
  const float* kscale;   // [K] or null (ring form only): operand b's element (n, k) is multiplied by kscale[k]
};

// Workgroup barrier that publishes this wave's LDS accesses but leaves its global loads in flight: __syncthreads() carries a
// workgroup-scope fence, i.e. s_waitcnt vmcnt(0), which would drain the prefetched tiles at every k-step.  The empty asm statements keep
// the compiler from moving LDS accesses across.
__device__ __forceinline__ void tr_barrier_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ELU as straight-line code: exp(v) - 1 (6e-8 absolute error) away from zero, the cubic Taylor polynomial (4e-8 relative) for -0.01 < v <= 0;
// both computed, the exponential pinned in front of the selects (left to itself the compiler turns the selects into a branch per element
// around v_exp_f32 -- 64 branches per result tile and wave)
__device__ __forceinline__ float tr_elu_flat(float v) {
  float e = __expf(v) - 1.0f;
#if defined(__AMDGCN__)
  asm volatile("" : "+v"(e));
#endif
  const float q = v * (1.0f + v * (0.5f + v * (1.0f / 6.0f)));
  return v > 0.f ? v : (v > -0.01f ? q : e);
}

__device__ __forceinline__ f32x16 tr_mfma(tr_u32x4 a, tr_u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tr_f16x8, a), __builtin_bit_cast(tr_f16x8, b), c, 0, 0, 0);
}

// waves per SIMD the tile kernel is compiled for: 214 registers at 2 (two workgroups per CU -- the registers, not the 40 KiB of LDS, set
// that); 3 (168 registers, 68 B of spills) measured 14-17 % slower, 4 spills 448 B
#ifndef TG_TILE_WAVES
#define TG_TILE_WAVES 2
#endif
template <int A_MODE, int B_MODE>
__global__ void __launch_bounds__(256, TG_TILE_WAVES) k_train_gemm(TrGemmArgs g) {
  constexpr bool A_KMINOR = A_MODE != 2, B_KMINOR = B_MODE != 2;
  tr_u16* As = reinterpret_cast<tr_u16*>(dyn_smem);
  tr_u16* Bs = As + 2 * TG_PART;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * TG_BM, n0 = blockIdx.y * TG_BN;  // the row tiles on grid.x: more than 65535 of them beyond 8 M rows
  const int kbeg = blockIdx.z * g.k_chunk;
  const int kend = g.K < kbeg + g.k_chunk ? g.K : kbeg + g.k_chunk;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // power-of-two scale of the gradient operand: its largest magnitude goes to [2^14, 2^15) (exact; undone in the epilogue)
  float a_scale = 1.0f, a_unscale = 1.0f;
  if (g.a_absmax != nullptr) {
    const unsigned mx = __float_as_uint(g.a_absmax[0]);
    const int e = (int)((mx >> 23) & 0xff);              // biased exponent of the largest magnitude (0: zero / subnormal tensor)
    if (e > 0 && e < 255) {
      const int sh = (127 + 14) - e;                       // multiply by 2^sh
      const int shc = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
      a_scale = __uint_as_float((unsigned)(127 + shc) << 23);
      a_unscale = __uint_as_float((unsigned)(127 - shc) << 23);
    }
  }
  // Two register stages of raw global loads: the tile of k-step t + 2 is requested while step t feeds the matrix pipe; the barrier of
  // a step publishes LDS writes only (tr_barrier_lds), so those loads stay in flight across it.
  const bool a_edge = m0 + TG_BM > g.a.nrows, b_edge = n0 + TG_BM > g.b.nrows, scaled = g.a_absmax != nullptr;
  float4 sa0[4], sb0[4], sa1[4], sb1[4];
  tr_load_tile<A_MODE>(g.a, m0, kbeg, kend, sa0, tid);
  tr_load_tile<B_MODE>(g.b, n0, kbeg, kend, sb0, tid);
  if (kbeg + TG_BK < kend) {
    tr_load_tile<A_MODE>(g.a, m0, kbeg + TG_BK, kend, sa1, tid);
    tr_load_tile<B_MODE>(g.b, n0, kbeg + TG_BK, kend, sb1, tid);
  }
  auto body = [&](float4 (&sa)[4], float4 (&sb)[4], int k0) {
    // interior tiles (nearly all of them) need no bounds masks: a uniform branch skips the 32 selects of a k-step
    const bool k_edge = k0 + TG_BK > kend;
    if (a_edge || k_edge) tr_mask_tile<A_MODE>(g.a, m0, k0, kend, sa, tid);
    if (b_edge || k_edge) tr_mask_tile<B_MODE>(g.b, n0, k0, kend, sb, tid);
    if (scaled) tr_store_tile<A_KMINOR, true>(As, sa, a_scale, tid);
    else tr_store_tile<A_KMINOR, false>(As, sa, 1.0f, tid);
    tr_store_tile<B_KMINOR, false>(Bs, sb, 1.0f, tid);
    tr_barrier_lds();
    if (!(TR_EXP & 2) && k0 + 2 * TG_BK < kend) {
      tr_load_tile<A_MODE>(g.a, m0, k0 + 2 * TG_BK, kend, sa, tid);
      tr_load_tile<B_MODE>(g.b, n0, k0 + 2 * TG_BK, kend, sb, tid);
    }
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      tr_u32x4 a[2][2], b[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          a[i][part] = *reinterpret_cast<const tr_u32x4*>(As + part * TG_PART + (wm * 64 + i * 32 + (lane & 31)) * TG_ROW + k16 * 16 + (lane >> 5) * 8);
          b[i][part] = *reinterpret_cast<const tr_u32x4*>(Bs + part * TG_PART + (wn * 64 + i * 32 + (lane & 31)) * TG_ROW + k16 * 16 + (lane >> 5) * 8);
        }
      // the three partial products of a tile form a dependent chain through its accumulator: issue them term by term ACROSS the four
      // tiles so that consecutive MFMAs are independent (smallest partial products first: mid.hi, hi.mid, hi.hi)
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (TR_EXP & 1) acc[i][j][term] += __uint_as_float(a[i][pa][0] ^ b[j][pb][0]);
            else acc[i][j] = tr_mfma(a[i][pa], b[j][pb], acc[i][j]);
          }
      }
    }
    tr_barrier_lds();
  };
  for (int k0 = kbeg; k0 < kend; k0 += 2 * TG_BK) {
    body(sa0, sb0, k0);
    if (k0 + TG_BK < kend) body(sa1, sb1, k0 + TG_BK);
  }
  // epilogue: D layout -- lane (j = lane & 31: column n, h = lane >> 5), register r: row (r & 3) + 8 (r >> 2) + 4 h.
  // Loads first (bias, per-point addend: clamped addresses, no branches, so they issue back to back), then arithmetic, then stores.
  const int nA = n0 + wn * 64 + (lane & 31), nB = nA + 32;
  const int nAc = nA < g.N ? nA : g.N - 1, nBc = nB < g.N ? nB : g.N - 1;
  const int mbase = m0 + wm * 64 + 4 * (lane >> 5);
  if (g.a_absmax != nullptr) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= a_unscale;
  }
  if (g.act_y != nullptr) {  // dZ = dX * act'(Y): the activation-derivative pass folded into the data gradient that feeds it
    const bool elu = g.act_y_kind == 1;
    if (g.act_y_vec) {
      // the Y tile through LDS (free after the k loop): 16-byte coalesced global reads, all of a thread's eight in flight, 64 rows per pass;
      // the waves of row half `pass` then read their elements in accumulator layout (row stride 132 floats)
      float* Yt = reinterpret_cast<float*>(dyn_smem);
      const int n4max = (g.N >> 2) - 1;
      for (int pass = 0; pass < 2; ++pass) {
        float4 yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = tid + 256 * j, row = idx >> 5, c4 = idx & 31;
          const int m = m0 + pass * 64 + row, n4 = (n0 >> 2) + c4;
          yv[j] = *reinterpret_cast<const float4*>(g.act_y + (long)(m < g.M ? m : g.M - 1) * g.ld_y + 4 * (n4 < n4max ? n4 : n4max));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = tid + 256 * j, row = idx >> 5, c4 = idx & 31;
          *reinterpret_cast<float4*>(Yt + row * 132 + 4 * c4) = yv[j];
        }
        __syncthreads();
        if (wm == pass) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float* yrow = Yt + (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 132 + wn * 64 + (lane & 31);
              const float yA = yrow[0], yB = yrow[32];
              acc[i][0][r] = yA > 0.f ? acc[i][0][r] : (elu ? acc[i][0][r] * (yA + 1.0f) : 0.f);
              acc[i][1][r] = yB > 0.f ? acc[i][1][r] : (elu ? acc[i][1][r] * (yB + 1.0f) : 0.f);
            }
        }
        __syncthreads();
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2);
          const float* yrow = g.act_y + (long)(m < g.M ? m : g.M - 1) * g.ld_y;
          const float yA = yrow[nAc], yB = yrow[nBc];
          acc[i][0][r] = yA > 0.f ? acc[i][0][r] : (elu ? acc[i][0][r] * (yA + 1.0f) : 0.f);
          acc[i][1][r] = yB > 0.f ? acc[i][1][r] : (elu ? acc[i][1][r] * (yB + 1.0f) : 0.f);
        }
    }
  }
  if (g.rowscale != nullptr) {  // Y = act(diag(s) (X W^T) + b): the forward of a Linear on x * s[row] without materialising x * s
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2);
        const float f = g.rowscale[m < g.M ? m : g.M - 1];
        acc[i][0][r] *= f;
        acc[i][1][r] *= f;
      }
  }
  if (g.addend != nullptr) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2);
        const float* add = g.addend + (long)((m < g.M ? m : g.M - 1) / g.add_div) * g.ld_add;
        acc[i][0][r] += add[nAc];
        acc[i][1][r] += add[nBc];
      }
  }
  if (g.bias != nullptr && blockIdx.z == 0) {
    const float bA = g.bias[nAc], bB = g.bias[nBc];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][0][r] += bA;
        acc[i][1][r] += bB;
      }
  }
  if (g.act == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = tr_elu_flat(acc[i][j][r]);
  }
  if (g.act == 2) {  // ReLU (MotionMLP)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
  }
  const bool okA = nA < g.N, okB = nB < g.N;
  if (g.c_vec && !(TR_EXP & 4)) {
    // plain stores of 16-byte-aligned rows: the tile leaves through LDS (64 rows per pass, accumulator layout in, row-major float4 out), so a
    // store instruction writes four complete 512-byte rows instead of two 128-byte pieces
    float* Ct = reinterpret_cast<float*>(dyn_smem);
    const int n4lim = g.N >> 2;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    float cmax = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* crow = Ct + (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 132 + wn * 64 + (lane & 31);
            crow[0] = acc[i][0][r];
            crow[32] = acc[i][1][r];
          }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j, row = idx >> 5, c4 = idx & 31;
        const int m = m0 + pass * 64 + row, n4 = (n0 >> 2) + c4;
        if (m < g.M && n4 < n4lim) {
          const float4 v = *reinterpret_cast<const float4*>(Ct + row * 132 + 4 * c4);
          *reinterpret_cast<float4*>(g.c + (long)m * g.ldc + 4 * n4) = v;
          csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;  // a thread keeps its four columns (c4 = tid & 31) over all its rows
          cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
      }
      __syncthreads();
    }
    if (g.colsum_part != nullptr) {
      // the bias gradient's share of this tile and the largest |dZ| (the next GEMMs' scale): the eight row groups meet in LDS and the workgroup
      // writes ONE partial row / value (no atomics; k_train_colsum_reduce adds the partials)
      *reinterpret_cast<float4*>(Ct + (tid >> 5) * 128 + 4 * (tid & 31)) = csum;
      cmax = wave_max(cmax);
      if (lane == 0) Ct[1024 + wave] = cmax;
      __syncthreads();
      if (tid < 128 && n0 + tid < g.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += Ct[k * 128 + tid];
        g.colsum_part[(long)blockIdx.x * g.ld_part + n0 + tid] = t;
      }
      if (tid == 0) g.amax_part[(long)blockIdx.x * gridDim.y + blockIdx.y] = fmaxf(fmaxf(Ct[1024], Ct[1025]), fmaxf(Ct[1026], Ct[1027]));
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2);
      if (m >= g.M) continue;
      if ((TR_EXP & 4) && acc[i][0][r] != 12345.678f) continue;
      float* crow = g.c + (long)m * g.ldc;
      if (g.accumulate == 0) {
        if (okA) crow[nA] = acc[i][0][r];
        if (okB) crow[nB] = acc[i][1][r];
      } else {  // += : fire-and-forget fp32 atomics (no read latency in the epilogue)
        if (okA) atomicAdd(crow + nA, acc[i][0][r]);
        if (okB) atomicAdd(crow + nB, acc[i][1][r]);
      }
    }
}

// =====================================================================================================================
// Ring form of the GEMM (round 3).  What the tile kernel above could not do: keep HBM busy while the matrix pipe works.  Its operand
// tiles pass through registers, and LLVM's s_waitcnt insertion gives up on a register-staged software pipeline (the loop's control-flow
// merges and its own register renaming make it wait for vmcnt(0) at every k-step), so a tile was requested only after the previous one
// had landed: measured, the kernel's HBM time ADDED to its LDS / MFMA time (profiles/r02_train_gemm_decomposition.txt; 0.40-0.50 of the
// HBM peak).  Staging registers filled by inline-asm loads, waited for by hand, do not survive either -- the register allocator
// spills or copies them while the loads are in flight (tried: r03).  Here no operand byte touches a register before it is needed:
//   * tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, inline asm: the compiler neither sees nor waits for them) into a ring
//     of two 32 KiB slots (A | B, raw fp32); the request for step t + 1 is issued at the start of step t and waited for, by count, at the
//     start of step t + 1 -- across tile boundaries: a workgroup is PERSISTENT and walks a sequence of tiles, so the first tiles of the next
//     tile land while the epilogue of this one runs;
//   * the MFMA fragments are read straight from the raw tiles and split into their two half parts in registers (per use: twice the
//     conversions of the tile kernel, which the matrix pipe hides; no image writes, ONE barrier per k-step instead of two);
//   * k-minor tiles [128 rows][32 k] keep their 16-byte quads XOR-swizzled -- row r holds quad q at position q ^ ((r >> 1) & 7) -- so that
//     the fragment reads (b128, 16 lanes = 16 consecutive rows) are bank-conflict free while a DMA instruction still fetches whole
//     128-byte lines; k-major tiles [32 k][128 rows] are read with eight b32 per fragment (lanes = consecutive rows: conflict free);
//   * epilogue, fast form (plain 16-byte-aligned result rows): the accumulators leave through the ring slot just consumed, 64 rows per
//     pass, and bias / per-point addend / activation derivative of the saved output / activation / column sums are applied to the
//     row-major float4 items on their way out (coalesced 16-byte loads and stores); general form: scalar code in accumulator layout.
// Operands it takes: k-minor with aligned quads (the tile kernel's mode 0) or k-major with 16-byte-aligned rows; anything else (a weight
// slice with an odd row stride, one-column outputs' gradients) stays on the tile kernel.
// =====================================================================================================================
#define TR_SLOT_FLOATS ((TG_BM + TG_BN) * TG_BK)  // A tile | B tile, raw fp32: 32 KiB
#define TR_RING_BYTES (2 * TR_SLOT_FLOATS * 4)    // two slots, 64 KiB: two workgroups per CU

typedef float tr_f32x4 __attribute__((ext_vector_type(4)));
#ifndef TR_ASM_DMA
#define TR_ASM_DMA 1
#endif
#ifndef TR_RX
#define TR_RX 0  /* developer decomposition builds (tools/gemm_decompose.sh; results wrong by construction): 1 no result stores, 2 no operand requests inside the loop, 4 no MFMAs, 8 no epilogue, 16 no fragment conversions */
#endif

// at most N vector-memory operations of this wave outstanding (they complete in issue order)
template <int N>
__device__ __forceinline__ void tr_wait_vm() {
#if defined(__AMDGCN__) && TR_ASM_DMA
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit count");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// 16 bytes per lane, global -> LDS without passing through registers: lane l's quad lands at wave_base + 16 l (wave_base wave-uniform)
__device__ __forceinline__ void tr_dma16(const float* gp, float* wave_base, int lane) {
#if defined(__AMDGCN__) && TR_ASM_DMA
  const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)wave_base);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(off) : "m0", "memory");
#pragma clang diagnostic pop
#else
  reinterpret_cast<tr_f32x4*>(wave_base)[lane] = *reinterpret_cast<const tr_f32x4*>(gp);
#endif
}

// request one operand tile into `dst` (4096 floats).  MODE 0: [128 rows][32 k] k-minor, swizzled quads; MODE 2: [32 k][128 rows] k-major.
// Every address is clamped into the operand (rows beyond it only feed result rows / columns that are never stored; the k tail is
// zeroed in the fragments).
template <int MODE>
__device__ __forceinline__ void tr_ring_request(const TrOperand& o, int row0, int k0, int kend, float* dst, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int L = j * 256 + tid;  // the quad's position in the tile
    const float* gp;
    if (MODE == 0) {
      const int row = L >> 3, kq = (L & 7) ^ ((row >> 1) & 7);
      const int r = row0 + row, rc = r < o.nrows ? r : o.nrows - 1;
      const int k = k0 + 4 * kq, kc = k < kend ? k : 0;  // a row holds at least round_up4(kend) floats (checked by the host wrapper)
      gp = o.p + (long)rc * o.rs + kc;
    } else {
      const int kk = L >> 5, fq = L & 31;
      const int k = k0 + kk, kc = k < kend ? k : kend - 1;
      const int f = row0 + 4 * fq, fmax = ((o.nrows + 3) & ~3) - 4;  // a k-row holds at least round_up4(nrows) floats (host wrapper)
      gp = o.p + (long)kc * o.ks + (f < fmax ? f : fmax);
    }
    tr_dma16(gp, dst + 4 * (j * 256 + wave * 64), lane);
  }
}

// the lane's eight consecutive k of tile row `row` for the 16-k group k16 (h = lane >> 5), raw
template <int MODE>
__device__ __forceinline__ void tr_ring_fragment(const float* tile, int row, int k16, int h, float (&v)[8]) {
  if (MODE == 0) {
    const int s = (row >> 1) & 7, kq = 4 * k16 + 2 * h;
    const tr_f32x4 q0 = *reinterpret_cast<const tr_f32x4*>(tile + 4 * (row * 8 + (kq ^ s)));
    const tr_f32x4 q1 = *reinterpret_cast<const tr_f32x4*>(tile + 4 * (row * 8 + ((kq + 1) ^ s)));
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = q0[e]; v[4 + e] = q1[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(16 * k16 + 8 * h + e) * TG_BM + row];
  }
}
// eight raw values -> the two MFMA operand parts (hi | mid halves)
template <bool SCALED>
__device__ __forceinline__ void tr_ring_split(const float (&v)[8], float scale, tr_u32x4& hi, tr_u32x4& mid) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned h, m;
    tr_split2_pair<SCALED>(v[2 * e], v[2 * e + 1], scale, h, m);
    hi[e] = h; mid[e] = m;
  }
}

struct TrCursor {
  int unit, nt;          // position in the workgroup's sequence of tiles
  int m0, n0, kend, z;   // the tile
  int k0;                // next k-step
};
__device__ __forceinline__ void tr_cursor_tile(TrCursor& c, const TrGemmArgs& g) {
  c.z = c.unit / g.mt;
  c.m0 = (c.unit - c.z * g.mt) * TG_BM;
  c.n0 = c.nt * TG_BN;
  c.k0 = c.z * g.k_chunk;
  c.kend = g.K < c.k0 + g.k_chunk ? g.K : c.k0 + g.k_chunk;
}
// -> false when the sequence is exhausted (the cursor then stays on a valid tile: a request issued from it is harmless)
__device__ __forceinline__ bool tr_cursor_advance(TrCursor& c, const TrGemmArgs& g, int units, int stride) {
  c.k0 += TG_BK;
  if (c.k0 < c.kend) return true;
  int nt = c.nt + 1, unit = c.unit;
  if (nt == g.nt) { nt = 0; unit += stride; }
  if (unit >= units) { c.k0 -= TG_BK; return false; }
  c.nt = nt; c.unit = unit;
  tr_cursor_tile(c, g);
  return true;
}

__device__ __forceinline__ float tr_act(float v, int act) {
  if (act == 1) {
    // ELU: exp(v) - 1 (6e-8 absolute error) away from zero, the cubic Taylor polynomial (4e-8 relative) for -0.01 < v <= 0
    const float e = __expf(v) - 1.0f, q = v * (1.0f + v * (0.5f + v * (1.0f / 6.0f)));
    return v > 0.f ? v : (v > -0.01f ? q : e);
  }
  if (act == 2) return fmaxf(v, 0.f);  // ReLU (MotionMLP)
  return v;
}

// A workgroup walks units blockIdx.x, blockIdx.x + gridDim.x, ... (a unit = (reduction chunk z, row tile) with its column tiles back to
// back, so that the second column tile finds the row tile's operand in this CU's caches); grid = the resident workgroups of the device.
// KSCALE: operand b carries a scale on the reduction index (kscale) -- its own instantiation: as a run-time branch in the fragment loop
// it cost every weight gradient 10 %
// EPI: the epilogue -- -1 general form; fast form: 0 bias / activation only, 1 the saved output's activation derivative (act_y), 2 a per-point
// addend, 3 a row scale.  Compile-time so that the fast form's global loads are unconditional straight-line code (below).
template <int A_MODE, int B_MODE, int EPI, bool KSCALE>
__global__ void __launch_bounds__(256, 2) k_train_gemm_ring(TrGemmArgs g) {
  constexpr bool FAST = EPI >= 0;
  constexpr int THREADS = 256, PASSES = 2, ITEMS = 8;  // epilogue: 64 rows per pass, quads per thread and pass
  float* ring = reinterpret_cast<float*>(dyn_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int units = g.mt * g.nz, stride = gridDim.x;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // power-of-two scale of the gradient operand: its largest magnitude goes to [2^14, 2^15) (exact; undone in the epilogue)
  float a_scale = 1.0f, a_unscale = 1.0f;
  if (g.a_absmax != nullptr) {
    const unsigned mx = __float_as_uint(g.a_absmax[0]);
    const int e = (int)((mx >> 23) & 0xff);              // biased exponent of the largest magnitude (0: zero / subnormal tensor)
    if (e > 0 && e < 255) {
      const int sh = (127 + 14) - e;                       // multiply by 2^sh
      const int shc = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
      a_scale = __uint_as_float((unsigned)(127 + shc) << 23);
      a_unscale = __uint_as_float((unsigned)(127 - shc) << 23);
    }
  }
  const bool scaled = g.a_absmax != nullptr;
  TrCursor cur, pf;
  int newer_stores = 0;  // vector stores the epilogue just before certainly issued: they are newer than the request the next step waits for

  // ---- epilogue, fast form; Ct = the ring slot the tile's last step consumed ----
  // Loads and stores of a wave retire through ONE in-order counter on gfx9 (vmcnt), and the compiler waits with vmcnt(0) wherever a load
  // MAY be pending at a control-flow merge.  A load issued after a pass's stores therefore waits for the stores' acknowledgements, and a
  // load inside a run-time branch drains everything (measured: 6-12 us per tile, more than the tile's k-steps).  So: every global load of
  // the epilogue is issued HERE in one unconditional batch (what is loaded is a compile-time choice, EPI), consumed once -- the empty asm
  // statements below are where the compiler's wait lands, behind the first LDS pass -- and nothing but stores follows.
  auto epilogue_fast = [&](const TrCursor& t, float* Ct) __attribute__((always_inline)) {
    const int m0 = t.m0, n0 = t.n0;
    const int n4lim = g.N >> 2, c4 = tid & 31, n4 = (n0 >> 2) + c4, n4c = n4 < n4lim ? n4 : n4lim - 1;
    const bool elu_y = g.act_y_kind == 1;
    constexpr bool BIAS = EPI != 1;  // (a product that goes back through an activation derivative has no bias: checked by the host wrapper)
    const bool has_bias = BIAS && g.bias != nullptr && t.z == 0;
    tr_f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS) bias4 = *reinterpret_cast<const tr_f32x4*>(has_bias ? g.bias + 4 * n4c : g.a.p);  // (a valid 16-byte-aligned stand-in)
    tr_f32x4 sv[EPI == 1 || EPI == 2 ? PASSES * ITEMS : 1];
    float rsc[EPI == 3 ? PASSES * ITEMS : 1];
    if constexpr (EPI != 0) {
#pragma unroll
      for (int j = 0; j < PASSES * ITEMS; ++j) {
        const int m = m0 + (j / ITEMS) * 64 + (tid >> 5) + (THREADS / 32) * (j % ITEMS), mc = m < g.M ? m : g.M - 1;
        if constexpr (EPI == 1) sv[j] = *reinterpret_cast<const tr_f32x4*>(g.act_y + (long)mc * g.ld_y + 4 * n4c);
        if constexpr (EPI == 2) sv[j] = *reinterpret_cast<const tr_f32x4*>(g.addend + (long)(mc / g.add_div) * g.ld_add + 4 * n4c);
        if constexpr (EPI == 3) rsc[j] = g.rowscale[mc];
      }
    }
    // the accumulators -> row-major quads through the slot, 64 rows per pass; a thread ends up with 16 quads of its four columns
    tr_f32x4 res[PASSES * ITEMS];
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* crow = Ct + (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * TG_BN + wn * 64 + (lane & 31);
            crow[0] = acc[i][0][r] * a_unscale;
            crow[32] = acc[i][1][r] * a_unscale;
          }
      }
      tr_barrier_lds();
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) res[pass * ITEMS + j] = *reinterpret_cast<const tr_f32x4*>(Ct + ((tid >> 5) + (THREADS / 32) * j) * TG_BN + 4 * c4);
      tr_barrier_lds();  // the slot takes the second pass, then the column sums or the next request
    }
    // the batch of loads has had both LDS passes to land: consumed here, all of it (see above)
#if defined(__AMDGCN__)
    if constexpr (BIAS) asm volatile("" : "+v"(bias4));
    if constexpr (EPI == 1 || EPI == 2) {
#pragma unroll
      for (int j = 0; j < PASSES * ITEMS; ++j) asm volatile("" : "+v"(sv[j]));
    }
    if constexpr (EPI == 3) {
#pragma unroll
      for (int j = 0; j < PASSES * ITEMS; ++j) asm volatile("" : "+v"(rsc[j]));
    }
#endif
    if constexpr (BIAS) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bias4[c] = has_bias ? bias4[c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < PASSES * ITEMS; ++j) {
      tr_f32x4 v = res[j];
      if constexpr (EPI == 3) v *= rsc[j];
      if constexpr (EPI == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = sv[j][c] > 0.f ? v[c] : (elu_y ? v[c] * (sv[j][c] + 1.0f) : 0.f);
      }
      if constexpr (EPI == 2) v += sv[j];
      if constexpr (BIAS) v += bias4;
      res[j] = v;
    }
    if (g.act == 1) {  // uniform branches around the whole tile, straight-line code inside
#pragma unroll
      for (int j = 0; j < PASSES * ITEMS; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) res[j][c] = tr_elu_flat(res[j][c]);
    } else if (g.act == 2) {
#pragma unroll
      for (int j = 0; j < PASSES * ITEMS; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) res[j][c] = fmaxf(res[j][c], 0.f);
    }
    tr_f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    float cmax = 0.f;
#pragma unroll
    for (int j = 0; j < PASSES * ITEMS; ++j) {
      const int m = m0 + (j / ITEMS) * 64 + (tid >> 5) + (THREADS / 32) * (j % ITEMS);
      if (m < g.M && n4 < n4lim) {
        const tr_f32x4 v = res[j];
        if (!(TR_RX & 1) || v[0] == 1.2345f) *reinterpret_cast<tr_f32x4*>(g.c + (long)m * g.ldc + 4 * n4) = v;
        if (EPI <= 1 && g.colsum_part != nullptr) {  // (the bias gradient of the layer in front; the host wrapper keeps it off EPI 2 / 3)
          csum += v;  // a thread keeps its four columns (c4 = tid & 31) over all its rows
          cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
      }
    }
    if (g.colsum_part != nullptr) {
      // the bias gradient's share of this tile and the largest |dZ| (the next GEMMs' scale): the eight row groups meet in LDS and the workgroup
      // writes ONE partial row / value (no atomics; k_train_colsum_reduce adds the partials)
      *reinterpret_cast<tr_f32x4*>(Ct + (tid >> 5) * 128 + 4 * c4) = csum;
      cmax = wave_max(cmax);
      if (lane == 0) Ct[(THREADS / 32) * 128 + wave] = cmax;
      tr_barrier_lds();
      if (tid < 128 && n0 + tid < g.N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < THREADS / 32; ++k) s += Ct[k * 128 + tid];
        g.colsum_part[(long)(m0 / TG_BM) * g.ld_part + n0 + tid] = s;
      }
      if (tid == 0) {
        float mx = 0.f;
#pragma unroll
        for (int k = 0; k < THREADS / 64; ++k) mx = fmaxf(mx, Ct[(THREADS / 32) * 128 + k]);
        g.amax_part[(long)(m0 / TG_BM) * g.nt + t.nt] = mx;
      }
      tr_barrier_lds();  // the slot is the next request's target
    }
    newer_stores = (m0 + TG_BM <= g.M && n0 + TG_BN <= g.N) ? 16 : 0;
  };

  // ---- epilogue, general form: D layout -- lane (j = lane & 31: column n, h = lane >> 5), register r: row (r & 3) + 8 (r >> 2) + 4 h ----
  auto epilogue_general = [&](const TrCursor& t) __attribute__((always_inline)) {
    const int m0 = t.m0, n0 = t.n0;
    const int nA = n0 + wn * 64 + (lane & 31), nB = nA + 32;
    const int nAc = nA < g.N ? nA : g.N - 1, nBc = nB < g.N ? nB : g.N - 1;
    const int mbase = m0 + wm * 64 + 4 * h;
    const bool okA = nA < g.N, okB = nB < g.N;
    const bool elu_y = g.act_y_kind == 1;
    float bA = 0.f, bB = 0.f;
    if (g.bias != nullptr && t.z == 0) { bA = g.bias[nAc]; bB = g.bias[nBc]; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + i * 32 + (r & 3) + 8 * (r >> 2);
        const int mc = m < g.M ? m : g.M - 1;
        float vA = acc[i][0][r] * a_unscale, vB = acc[i][1][r] * a_unscale;
        if (g.act_y != nullptr) {  // dZ = dX * act'(Y): the activation-derivative pass folded into the data gradient that feeds it
          const float* yrow = g.act_y + (long)mc * g.ld_y;
          const float yA = yrow[nAc], yB = yrow[nBc];
          vA = yA > 0.f ? vA : (elu_y ? vA * (yA + 1.0f) : 0.f);
          vB = yB > 0.f ? vB : (elu_y ? vB * (yB + 1.0f) : 0.f);
        }
        if (g.rowscale != nullptr) {
          const float f = g.rowscale[mc];
          vA *= f;
          vB *= f;
        }
        if (g.addend != nullptr) {
          const float* add = g.addend + (long)(mc / g.add_div) * g.ld_add;
          vA += add[nAc];
          vB += add[nBc];
        }
        vA = tr_act(vA + bA, g.act);
        vB = tr_act(vB + bB, g.act);
        if (m < g.M) {
          float* crow = g.c + (long)m * g.ldc;
          if (g.accumulate == 0) {
            if (okA) crow[nA] = vA;
            if (okB) crow[nB] = vB;
          } else {  // += : fire-and-forget fp32 atomics (no read latency in the epilogue)
            if (okA) atomicAdd(crow + nA, vA);
            if (okB) atomicAdd(crow + nB, vB);
          }
        }
      }
    newer_stores = 0;
  };

  cur.unit = blockIdx.x; cur.nt = 0;
  tr_cursor_tile(cur, g);
  pf = cur;
  auto request = [&](int slot) __attribute__((always_inline)) {
    float* dst = ring + slot * TR_SLOT_FLOATS;
    tr_ring_request<A_MODE>(g.a, pf.m0, pf.k0, pf.kend, dst, tid);
    tr_ring_request<B_MODE>(g.b, pf.n0, pf.k0, pf.kend, dst + TG_BM * TG_BK, tid);
    tr_cursor_advance(pf, g, units, stride);
  };
  request(0);

  // one k-step on ring slot SLOT
  auto step = [&](auto slot_tag) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    float* At = ring + SLOT * TR_SLOT_FLOATS;
    float* Bt = At + TG_BM * TG_BK;
    // this step's tiles (the only request in flight; the stores of an epilogue just before are newer) have landed -- in every wave,
    // after the barrier, which also says that every wave has finished reading the other slot: it takes the next request
    if (newer_stores != 0) tr_wait_vm<16>();
    else tr_wait_vm<0>();
    newer_stores = 0;
    tr_barrier_lds();
    if (!(TR_RX & 2)) request(SLOT ^ 1);
    else tr_cursor_advance(pf, g, units, stride);
    const bool k_edge = cur.k0 + TG_BK > cur.kend;
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
      tr_u32x4 a[2][2], b[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float va[8], vb[8];
        tr_ring_fragment<A_MODE>(At, wm * 64 + i * 32 + (lane & 31), k16, h, va);
        tr_ring_fragment<B_MODE>(Bt, wn * 64 + i * 32 + (lane & 31), k16, h, vb);
        if constexpr (KSCALE) {
          // dW = dZ^T (diag(s) X): the scale of the reduction index on operand b.  Wave-uniform addresses (both halves' values, then a
          // select on h): scalar loads, which the vector-memory counter of the ring does not see
          // (constant address space: what makes the compiler pick s_load for a uniform address; the data was written by earlier kernels)
#if defined(__AMDGCN__)
          typedef const __attribute__((address_space(4))) float* tr_kptr;
          tr_kptr ksc = (tr_kptr)(uintptr_t)g.kscale;
#else
          const float* ksc = g.kscale;
#endif
          const int kb = cur.k0 + 16 * k16;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int klo = kb + e < g.K ? kb + e : g.K - 1, khi = kb + 8 + e < g.K ? kb + 8 + e : g.K - 1;
            const float flo = ksc[klo], fhi = ksc[khi];
            vb[e] *= h ? fhi : flo;
          }
        }
        if (k_edge) {  // the k tail of a tile (its last step only): both operands, a zero times a stray NaN is a NaN
          const int kb = cur.k0 + 16 * k16 + 8 * h;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            va[e] = kb + e < cur.kend ? va[e] : 0.f;
            vb[e] = kb + e < cur.kend ? vb[e] : 0.f;
          }
        }
        if (TR_RX & 16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[i][0][e] = __float_as_uint(va[e]); a[i][1][e] = __float_as_uint(va[4 + e]); b[i][0][e] = __float_as_uint(vb[e]); b[i][1][e] = __float_as_uint(vb[4 + e]); }
        } else {
          if (scaled) tr_ring_split<true>(va, a_scale, a[i][0], a[i][1]);
          else tr_ring_split<false>(va, 1.0f, a[i][0], a[i][1]);
          tr_ring_split<false>(vb, 1.0f, b[i][0], b[i][1]);
        }
      }
      // the three partial products of a tile form a dependent chain through its accumulator: issue them term by term ACROSS the four
      // tiles so that consecutive MFMAs are independent (smallest partial products first: mid.hi, hi.mid, hi.hi)
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (TR_RX & 4) acc[i][j][term] += __uint_as_float(a[i][pa][0] ^ b[j][pb][0]);
            else acc[i][j] = tr_mfma(a[i][pa], b[j][pb], acc[i][j]);
          }
      }
    }
    if ((TR_RX & 8) && cur.k0 + TG_BK >= cur.kend) {
      if (acc[0][0][0] == 1.2345f) g.c[tid] = acc[1][1][3];
    } else if (cur.k0 + TG_BK >= cur.kend) {  // the tile's last step
      if (FAST) {
        tr_barrier_lds();  // every wave has finished reading the slot: it takes the result tile (64 rows x 128 floats per pass)
        epilogue_fast(cur, At);
      } else {
        epilogue_general(cur);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
  };
  for (;;) {
    step(std::integral_constant<int, 0>{});
    if (!tr_cursor_advance(cur, g, units, stride)) break;
    step(std::integral_constant<int, 1>{});
    if (!tr_cursor_advance(cur, g, units, stride)) break;
  }
  tr_wait_vm<0>();  // the request past the end of the sequence must land before the workgroup's LDS is released
}

// resident workgroups of k_train_gemm_ring<A, B, E, S> per device: the persistent kernel's grid (queried once per instantiation and device)
template <int A_MODE, int B_MODE, int EPI, bool KSCALE>
static int tr_ring_slots() {
  static int slots[DYN_MAX_DEVICES] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int d = (dev >= 0 && dev < DYN_MAX_DEVICES) ? dev : 0;
  if (slots[d] == 0) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_train_gemm_ring<A_MODE, B_MODE, EPI, KSCALE>, 256, TR_RING_BYTES) != hipSuccess || per_cu < 1) per_cu = 2;
    const int n_cu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    slots[d] = per_cu * n_cu;
  }
  return slots[d];
}
// which form takes a product: 0 automatic -- the ring form for the backward products (weight gradient: both operands stream from HBM,
// +25-35 % measured; data gradient: +0-10 %), the tile kernel for the forward ones (ring: level to -4 % since its epilogue is straight-line
// code with one batch of loads; -8 % to -30 % before) --, 1 tile kernel only, 2 ring form wherever the operands allow it.  DYNIBAR_TRAIN_GEMM = auto | tile |
// ring sets the initial value; dyn_train_gemm_mode() changes it (A/B timing, tests).
static int g_tr_gemm_mode = -1;
static int tr_gemm_mode() {
  if (g_tr_gemm_mode < 0) {
    const char* e = getenv("DYNIBAR_TRAIN_GEMM");
    g_tr_gemm_mode = e == nullptr ? 0 : (strcmp(e, "tile") == 0 ? 1 : (strcmp(e, "ring") == 0 ? 2 : 0));
  }
  return g_tr_gemm_mode;
}
extern "C" int dyn_train_gemm_mode(int mode) {
  DYN_REQUIRE(mode >= 0 && mode <= 2, "dyn_train_gemm_mode: mode %d (0 automatic, 1 tile, 2 ring)", mode);
  g_tr_gemm_mode = mode;
  return 0;
}

extern "C" int dyn_train_gemm(const DynTrainGemmParams* p, void* stream) {
  DYN_REQUIRE(p != nullptr && p->A && p->B && p->C, "dyn_train_gemm: null pointer");
  DYN_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, "dyn_train_gemm: empty problem (M %d N %d K %d)", p->M, p->N, p->K);
  DYN_REQUIRE((p->a_rs == 1 || p->a_ks == 1) && (p->b_rs == 1 || p->b_ks == 1), "dyn_train_gemm: each operand needs one unit stride");
  DYN_REQUIRE(p->k_split >= 1 && (p->k_split == 1 || (p->accumulate == 2 && p->act == 0 && p->addend == nullptr)),
              "dyn_train_gemm: a split reduction needs accumulate = 2 (atomic) and a linear epilogue");
  DYN_REQUIRE(p->addend == nullptr || p->add_div >= 1, "dyn_train_gemm: add_div must be >= 1");
  DYN_REQUIRE(p->rowscale == nullptr || p->k_split == 1, "dyn_train_gemm: rowscale needs k_split = 1");
  TrGemmArgs g;
  g.a.p = p->A; g.a.rs = p->a_rs; g.a.ks = p->a_ks; g.a.nrows = p->M; g.a.aligned = ((uintptr_t)p->A & 15) == 0;
  g.b.p = p->B; g.b.rs = p->b_rs; g.b.ks = p->b_ks; g.b.nrows = p->N; g.b.aligned = ((uintptr_t)p->B & 15) == 0;
  g.c = p->C; g.ldc = p->ldc; g.M = p->M; g.N = p->N; g.K = p->K;
  int chunk = (p->K + p->k_split - 1) / p->k_split;
  chunk = ((chunk + TG_BK - 1) / TG_BK) * TG_BK;
  g.k_chunk = chunk;
  const int nz = (p->K + chunk - 1) / chunk;
  g.bias = p->bias; g.addend = p->addend; g.ld_add = p->ld_add; g.add_div = p->add_div > 0 ? p->add_div : 1;
  g.act = p->act; g.accumulate = p->accumulate; g.a_absmax = p->a_absmax;
  DYN_REQUIRE(p->act_y == nullptr || (p->accumulate == 0 && p->k_split == 1 && (p->act_y_kind == 1 || p->act_y_kind == 2)),
              "dyn_train_gemm: act_y needs accumulate = 0, k_split = 1 and act_y_kind 1 (ELU) or 2 (ReLU)");
  g.act_y = p->act_y; g.ld_y = p->ld_y; g.act_y_kind = p->act_y_kind;
  g.c_vec = p->accumulate == 0 && (p->N & 3) == 0 && (p->ldc & 3) == 0 && ((uintptr_t)p->C & 15) == 0;
  DYN_REQUIRE(p->colsum_part == nullptr || (g.c_vec && p->amax_part != nullptr && p->ld_part >= p->N),
              "dyn_train_gemm: colsum_part needs plain 16-byte-aligned stores (accumulate 0, N and ldc multiples of 4), amax_part and ld_part >= N");
  g.colsum_part = p->colsum_part; g.ld_part = p->ld_part; g.amax_part = p->amax_part;
  g.rowscale = p->rowscale; g.kscale = p->kscale;
  g.act_y_vec = p->act_y != nullptr && (p->N & 3) == 0 && (p->ld_y & 3) == 0 && ((uintptr_t)p->act_y & 15) == 0;
  g.mt = dyn_cdiv(p->M, TG_BM); g.nt = dyn_cdiv(p->N, TG_BN); g.nz = nz;
  {
    // ring form: each operand k-minor with aligned quads (the tile kernel's mode 0) or k-major with 16-byte-aligned k-rows
    const int k4r = (p->K + 3) & ~3;
    auto ring_mode = [&](const TrOperand& o) -> int {
      if (!o.aligned) return -1;
      if (o.ks == 1) return ((o.rs & 3) == 0 && o.rs >= k4r) ? 0 : -1;
      return ((o.ks & 3) == 0 && o.ks >= ((o.nrows + 3) & ~3)) ? 2 : -1;
    };
    const int ra = ring_mode(g.a), rb = ring_mode(g.b);
    const bool add_vec = p->addend == nullptr || ((p->ld_add & 3) == 0 && ((uintptr_t)p->addend & 15) == 0);
    const int sides = (p->act_y != nullptr) + (p->addend != nullptr) + (p->rowscale != nullptr);
    const bool fast = g.c_vec && (p->bias == nullptr || ((uintptr_t)p->bias & 15) == 0) && (p->act_y == nullptr || g.act_y_vec) && add_vec && sides <= 1;
    const int mode = tr_gemm_mode();
    // (forward products with an addend / row scale on the ring form -- the tile kernel reads those by 4-byte accumulator-layout loads -- measured:
    // GEMM time -0.2 ms per iteration, not worth a second arithmetic order in the forward pass; -DTR_AUTO_SIDE_RING=1 to try)
#ifndef TR_AUTO_SIDE_RING
#define TR_AUTO_SIDE_RING 0
#endif
    const bool side_fwd = TR_AUTO_SIDE_RING && fast && (p->addend != nullptr || p->rowscale != nullptr);
    if (ra >= 0 && rb >= 0 && (mode == 2 || (mode == 0 && (ra == 2 || rb == 2 || side_fwd)) || p->kscale != nullptr)) {  // (a scale on the reduction index exists on the ring form only: also under DYNIBAR_TRAIN_GEMM=tile)
      const long units = (long)g.mt * g.nz;
      // the epilogue (EPI of k_train_gemm_ring): the fast forms that are instantiated -- forward shape (both operands k-minor): plain, addend,
      // row scale; data-gradient shape (b k-major): plain, act_y, addend --, else the general form
      int epi = -1;
      if (fast && ra == 0) {
        const int want = p->act_y != nullptr ? 1 : (p->addend != nullptr ? 2 : (p->rowscale != nullptr ? 3 : 0));
        if (rb == 0 && want != 1) epi = want;
        if (rb == 2 && want <= 2) epi = want;  // (addend on the data-gradient shape: dX += ..., the addend being C itself)
        if (epi == 1 && p->bias != nullptr) epi = -1;  // the act_y form carries no bias
      }
      const bool ring_sums = p->colsum_part == nullptr || epi == 0 || epi == 1;  // else: the tile kernel
#define TG_RING(A, B, E, S)                                                                                                                \
  if (ring_sums && ra == A && rb == B && epi == E && (p->kscale != nullptr) == S) {                                                                  \
    const long slots = tr_ring_slots<A, B, E, S>();                                                                                        \
    const dim3 rgrid((unsigned)(units < slots ? units : slots));                                                                           \
    DYN_LAUNCH(DYN_K_TRAIN_GEMM, "dyn_train_gemm", (k_train_gemm_ring<A, B, E, S>), rgrid, dim3(256), TR_RING_BYTES, (hipStream_t)stream,   \
               g);                                                                                                                         \
    return 0;                                                                                                                              \
  }
      TG_RING(0, 0, 0, false) TG_RING(0, 0, 2, false) TG_RING(0, 0, 3, false) TG_RING(0, 0, -1, false)
      TG_RING(0, 2, 0, false) TG_RING(0, 2, 1, false) TG_RING(0, 2, 2, false) TG_RING(0, 2, -1, false)
      TG_RING(2, 0, -1, false) TG_RING(2, 2, -1, false)
      TG_RING(2, 2, -1, true)  // the weight gradient of a Linear on x * s[row]
#undef TG_RING
    }
  }
  DYN_REQUIRE(p->kscale == nullptr, "dyn_train_gemm: kscale is for the weight-gradient shape on the ring form (both operands k-major with 16-byte-aligned rows, accumulation into C; dyn_train_gemm_mode 0 or 2)");
  const dim3 grid(dyn_cdiv(p->M, TG_BM), dyn_cdiv(p->N, TG_BN), nz);
  // loader mode per operand: 0 = k-minor dwordx4 (aligned base, row stride a multiple of 4 floats and >= round_up4(K)), 1 = k-minor
  // dword, 2 = k-major
  const int k4 = (p->K + 3) & ~3;
  const int am = p->a_ks == 1 ? ((g.a.aligned && (p->a_rs & 3) == 0 && p->a_rs >= k4 && p->k_split == 1) ? 0 : 1) : 2;
  const int bm = p->b_ks == 1 ? ((g.b.aligned && (p->b_rs & 3) == 0 && p->b_rs >= k4 && p->k_split == 1) ? 0 : 1) : 2;
#define TG_CASE(A, B)                                                                                                                      \
  if (am == A && bm == B) {                                                                                                                \
    DYN_LAUNCH(DYN_K_TRAIN_GEMM, "dyn_train_gemm", (k_train_gemm<A, B>), grid, dim3(256), TG_LDS_BYTES, (hipStream_t)stream, g);            \
    return 0;                                                                                                                              \
  }
  TG_CASE(0, 0) TG_CASE(0, 1) TG_CASE(0, 2) TG_CASE(1, 0) TG_CASE(1, 1) TG_CASE(1, 2) TG_CASE(2, 0) TG_CASE(2, 1) TG_CASE(2, 2)
#undef TG_CASE
  return 0;
}

// dbias[n] += sum over the row tiles of colsum_part[tile, n];  *absmax = max(*absmax, amax_part[...]): the second stage of the sums the
// data-gradient GEMM leaves per workgroup.  grid (column blocks of 64, row chunks): a block adds its chunk and issues one atomic per column.
__global__ void __launch_bounds__(256) k_train_colsum_reduce(const float* __restrict__ part, long tiles, int N, long ld, float* __restrict__ dbias,
                                                             const float* __restrict__ amax_part, long n_amax, float* __restrict__ absmax) {
  float* red = reinterpret_cast<float*>(dyn_smem);  // [4][64] + 4
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const long per = (tiles + gridDim.y - 1) / gridDim.y, t0 = (long)blockIdx.y * per, t1 = t0 + per < tiles ? t0 + per : tiles;
  float sum = 0.f;
  if (c < N)
    for (long t = t0 + rl; t < t1; t += 4) sum += part[t * ld + c];
  red[rl * 64 + (threadIdx.x & 63)] = sum;
  float m = 0.f;
  if (blockIdx.x == 0) {
    const long pa = (n_amax + gridDim.y - 1) / gridDim.y, a0 = (long)blockIdx.y * pa, a1 = a0 + pa < n_amax ? a0 + pa : n_amax;
    for (long i = a0 + threadIdx.x; i < a1; i += 256) m = fmaxf(m, amax_part[i]);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[256 + rl] = m;
  __syncthreads();
  if (threadIdx.x < 64 && c < N && dbias != nullptr) atomicAdd(dbias + c, (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]));
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float mm = fmaxf(fmaxf(red[256], red[257]), fmaxf(red[258], red[259]));
    if (mm > 0.f) atomicMax(reinterpret_cast<unsigned*>(absmax), __float_as_uint(mm));
  }
}
extern "C" int dyn_train_colsum_reduce(const float* colsum_part, long tiles, int N, long ld_part, float* dbias, const float* amax_part, long n_amax,
                                       float* absmax, void* stream) {
  DYN_REQUIRE(colsum_part && amax_part && absmax && tiles > 0 && N > 0 && ld_part >= N && n_amax > 0, "dyn_train_colsum_reduce: bad arguments");
  const unsigned chunks = (unsigned)(tiles < 64 ? tiles : 64);
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_colsum_reduce", k_train_colsum_reduce, dim3((unsigned)dyn_cdiv(N, 64), chunks), dim3(256), 260 * sizeof(float),
             (hipStream_t)stream, colsum_part, tiles, N, ld_part, dbias, amax_part, n_amax, absmax);
  return 0;
}

// =====================================================================================================================
// Row kernels.  N = P * V rows (row = point * V + view), P = R * S points.
// =====================================================================================================================
__device__ __forceinline__ float tr_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// dZ = dY * act'(Y) in place (ELU: y > 0 ? 1 : y + 1, from the saved OUTPUT), plus the bias gradient (column sums, atomics) and the
// gradient of a per-point addend (sums over the seg rows of a point).  A thread owns one column of a run of rows.
// Largest magnitude of a block into *absmax (non-negative floats order like their bit patterns): waves meet in LDS so that a block issues
// ONE atomic -- same-address atomics serialise in L2 (thousands of waves on one word cost more than the streaming pass itself).
// `red` = 4 floats of LDS not in use by anything else at this point.
__device__ __forceinline__ void tr_block_absmax(float amax, float* red, float* __restrict__ absmax) {
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(absmax), __float_as_uint(m));
  }
}

// U consecutive rows of one column: all loads first (they are in flight together), then the derivative, the stores and the sums
template <int U>
__device__ __forceinline__ float tr_act_rows(float* __restrict__ dy, const float* __restrict__ y, long r, int c, long ld_dy, long ld_y, int act, float& amax) {
  float d[U], yv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) d[u] = dy[(r + u) * ld_dy + c];
  if (act != 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) yv[u] = y[(r + u) * ld_y + c];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d[u] = yv[u] > 0.f ? d[u] : (act == 1 ? d[u] * (yv[u] + 1.0f) : 0.f);  // ELU' = y + 1 below zero; ReLU' = 0
      dy[(r + u) * ld_dy + c] = d[u];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    s += d[u];
    amax = fmaxf(amax, fabsf(d[u]));
  }
  return s;
}
__device__ __forceinline__ float tr_act_span(float* __restrict__ dy, const float* __restrict__ y, long ra, long rb, int c, long ld_dy, long ld_y, int act,
                                             float& amax) {
  float s = 0.f;
  long r = ra;
  for (; r + 7 < rb; r += 8) s += tr_act_rows<8>(dy, y, r, c, ld_dy, ld_y, act, amax);
  for (; r + 3 < rb; r += 4) s += tr_act_rows<4>(dy, y, r, c, ld_dy, ld_y, act, amax);
  for (; r < rb; ++r) s += tr_act_rows<1>(dy, y, r, c, ld_dy, ld_y, act, amax);
  return s;
}

__global__ void __launch_bounds__(256) k_train_act_bwd(float* __restrict__ dy, const float* __restrict__ y, long rows, int cols, long ld_dy, long ld_y,
                                                       int act, float* __restrict__ dbias, int seg, float* __restrict__ dseg, long ld_seg, int ct,
                                                       int run, float* __restrict__ absmax) {
  const int cx = threadIdx.x % ct, cy = threadIdx.x / ct;
  const long chunk = (long)blockIdx.x * (256 / ct) + cy;
  const long r0 = chunk * run;
  const long rend = r0 + run < rows ? r0 + run : rows;
  float amax = 0.f;
  for (int c = cx; c < cols; c += ct) {
    float colsum = 0.f;
    if (dseg == nullptr) {
      colsum = tr_act_span(dy, y, r0, rend, c, ld_dy, ld_y, act, amax);
    } else {  // r0 and rows are multiples of seg (run is): whole segments, one sum stored per segment
      long sidx = r0 / seg;
      for (long r = r0; r < rend; r += seg) {
        const float segsum = tr_act_span(dy, y, r, r + seg, c, ld_dy, ld_y, act, amax);
        dseg[sidx++ * ld_seg + c] = segsum;
        colsum += segsum;
      }
    }
    if (dbias != nullptr && r0 < rows) atomicAdd(dbias + c, colsum);
  }
  if (absmax != nullptr) tr_block_absmax(amax, reinterpret_cast<float*>(dyn_smem), absmax);  // largest |dZ|: scales the backward GEMMs
}

// The same pass with 16-byte accesses (cols, both leading dimensions and ld_seg multiples of four, 16-byte aligned bases: the 64 / 128 /
// 256-wide hidden layers, i.e. nearly all of the traffic).  L = cols / 4 lanes span a row, the block's G = 256 / L lane groups own
// consecutive spans of `span` rows (whole segments), four rows per trip; the G partial column sums meet in LDS, so a block still issues one
// atomic per column.
__device__ __forceinline__ float4 tr_act4(float4 d, float4 y, int act) {
  if (act == 1) {
    d.x = y.x > 0.f ? d.x : d.x * (y.x + 1.0f); d.y = y.y > 0.f ? d.y : d.y * (y.y + 1.0f);
    d.z = y.z > 0.f ? d.z : d.z * (y.z + 1.0f); d.w = y.w > 0.f ? d.w : d.w * (y.w + 1.0f);
  } else {
    d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f; d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
  }
  return d;
}
template <int U>
__device__ __forceinline__ void tr_act_rows4(float4* __restrict__ dy, const float4* __restrict__ y, long r, long ld_dy4, long ld_y4, int act, float4& sum,
                                             float& amax) {
  float4 d[U], yv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) d[u] = dy[(r + u) * ld_dy4];
  if (act != 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) yv[u] = y[(r + u) * ld_y4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d[u] = tr_act4(d[u], yv[u], act);
      dy[(r + u) * ld_dy4] = d[u];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    sum.x += d[u].x; sum.y += d[u].y; sum.z += d[u].z; sum.w += d[u].w;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d[u].x), fabsf(d[u].y))), fmaxf(fabsf(d[u].z), fabsf(d[u].w)));
  }
}
__device__ __forceinline__ void tr_act_span4(float4* __restrict__ dy, const float4* __restrict__ y, long ra, long rb, long ld_dy4, long ld_y4, int act,
                                             float4& sum, float& amax) {
  long r = ra;
  for (; r + 3 < rb; r += 4) tr_act_rows4<4>(dy, y, r, ld_dy4, ld_y4, act, sum, amax);
  for (; r < rb; ++r) tr_act_rows4<1>(dy, y, r, ld_dy4, ld_y4, act, sum, amax);
}
__global__ void __launch_bounds__(256) k_train_act_bwd4(float4* __restrict__ dy, const float4* __restrict__ y, long rows, int L, long ld_dy4, long ld_y4,
                                                        int act, float* __restrict__ dbias, int seg, float4* __restrict__ dseg, long ld_seg4, int span,
                                                        float* __restrict__ absmax) {
  float4* part = dyn_smem;  // [G][L] partial column sums (4 KiB)
  const int G = 256 / L, g = threadIdx.x / L, q = threadIdx.x - g * L;
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float amax = 0.f;
  if (g < G) {
    const long ra = ((long)blockIdx.x * G + g) * span;
    const long rb = ra + span < rows ? ra + span : rows;
    float4* dq = dy + q;
    const float4* yq = y != nullptr ? y + q : nullptr;
    if (dseg == nullptr) {
      if (ra < rb) tr_act_span4(dq, yq, ra, rb, ld_dy4, ld_y4, act, colsum, amax);
    } else {  // ra, span and rows are multiples of seg
      long sidx = ra / seg;
      for (long r = ra; r < rb; r += seg) {
        float4 ss = make_float4(0.f, 0.f, 0.f, 0.f);
        tr_act_span4(dq, yq, r, r + seg, ld_dy4, ld_y4, act, ss, amax);
        dseg[sidx++ * ld_seg4 + q] = ss;
        colsum.x += ss.x; colsum.y += ss.y; colsum.z += ss.z; colsum.w += ss.w;
      }
    }
    part[g * L + q] = colsum;
  }
  __syncthreads();
  if (dbias != nullptr && threadIdx.x < 4 * L) {  // column threadIdx.x: the G partial sums, then ONE atomic per column and block
    const float* pf = reinterpret_cast<const float*>(part);
    float s = 0.f;
    for (int k = 0; k < G; ++k) s += pf[k * 4 * L + threadIdx.x];
    atomicAdd(dbias + threadIdx.x, s);
  }
  if (absmax != nullptr) tr_block_absmax(amax, reinterpret_cast<float*>(dyn_smem + 256), absmax);
}

extern "C" int dyn_train_act_bwd(float* dY, const float* Y, long rows, int cols, long ld_dy, long ld_y, int act, float* dbias, int seg,
                                 float* dseg, long ld_seg, float* absmax, void* stream) {
  DYN_REQUIRE(dY != nullptr && rows > 0 && cols > 0, "dyn_train_act_bwd: bad arguments");
  DYN_REQUIRE(act == 0 || Y != nullptr, "dyn_train_act_bwd: ELU / ReLU backward needs the saved output");
  DYN_REQUIRE(dseg == nullptr || (seg >= 1 && rows % seg == 0), "dyn_train_act_bwd: rows must be whole segments");
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if ((cols & 3) == 0 && cols >= 16 && cols <= 256 && (ld_dy & 3) == 0 && (act == 0 || (ld_y & 3) == 0) && al16(dY) && (act == 0 || al16(Y)) &&
      (dseg == nullptr || ((ld_seg & 3) == 0 && al16(dseg)))) {
    const int L = cols / 4, G = 256 / L;
    const int unit = dseg != nullptr ? seg : 1;
    const long want = (rows >= (1L << 20) ? 2048 : 256) / G;  // rows per block: as the scalar kernel's runs (same-address atomics per column)
    const int span = (int)((want + unit - 1) / unit) * unit;
    const long blocks = (rows + (long)G * span - 1) / ((long)G * span);
    DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_act_bwd", k_train_act_bwd4, dim3((unsigned)blocks), dim3(256), 257 * sizeof(float4), (hipStream_t)stream,
               reinterpret_cast<float4*>(dY), reinterpret_cast<const float4*>(act != 0 ? Y : nullptr), rows, L, ld_dy / 4, ld_y / 4, act, dbias, seg,
               reinterpret_cast<float4*>(dseg), ld_seg / 4, span, absmax);
    return 0;
  }
  const int ct = cols <= 32 ? 32 : cols <= 64 ? 64 : cols <= 128 ? 128 : 256;
  int run = dseg != nullptr ? seg : 1;
  // rows per thread: the bias gradient costs one fp32 atomic per column and run, and same-address atomics are what this kernel waits for
  // (measured: 64-row runs 2x slower than 256-row runs): 1024 rows for the 3 M-row matrices, fewer where that would leave CUs idle
  const int target = rows >= (1L << 20) ? 1024 : 256;
  while (run < target) run += (dseg != nullptr ? seg : 1);
  const long chunks = (rows + run - 1) / run;
  const int per_block = 256 / ct;
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_act_bwd", k_train_act_bwd, dim3((unsigned)((chunks + per_block - 1) / per_block)), dim3(256), sizeof(float4),
             (hipStream_t)stream, dY, Y, rows, cols, ld_dy, ld_y, act, dbias, seg, dseg, ld_seg, ct, run, absmax);
  return 0;
}

// largest |x| of a [rows, cols] matrix (leading dimension ld): the scale of a gradient tensor that no activation-derivative pass produced
__global__ void __launch_bounds__(256) k_train_absmax(const float* __restrict__ x, long rows, int cols, long ld, int ct, int vec, float* __restrict__ absmax) {
  float amax = 0.f;
  if (vec) {  // contiguous, a multiple of four elements, 16-byte aligned: two independent 16-byte loads per trip
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const long n = rows * cols / 4, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 2 * stride) {
      const long i1 = i + stride < n ? i + stride : i;
      const float4 a = x4[i], b = x4[i1];
      amax = fmaxf(amax, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
    }
  } else if (ld == cols) {  // contiguous: one flat stream, four independent loads per trip
    const long n = rows * cols, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
      const long i1 = i + stride < n ? i + stride : i, i2 = i + 2 * stride < n ? i + 2 * stride : i, i3 = i + 3 * stride < n ? i + 3 * stride : i;
      const float a0 = x[i], a1 = x[i1], a2 = x[i2], a3 = x[i3];
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(a0), fabsf(a1))), fmaxf(fabsf(a2), fabsf(a3)));
    }
  } else {  // a column window of a wider matrix: a thread keeps its column, the block walks rows (no per-element division); re-reading
            // a valid row past the end is harmless for a maximum, so the four loads of a trip are issued without branches
    const int cx = threadIdx.x % ct, cy = threadIdx.x / ct;
    const long stride = (long)gridDim.x * (256 / ct);
    for (long r = (long)blockIdx.x * (256 / ct) + cy; r < rows; r += 4 * stride) {
      const long r1 = r + stride < rows ? r + stride : r, r2 = r + 2 * stride < rows ? r + 2 * stride : r, r3 = r + 3 * stride < rows ? r + 3 * stride : r;
      for (int c = cx; c < cols; c += ct) {
        const float a0 = x[r * ld + c], a1 = x[r1 * ld + c], a2 = x[r2 * ld + c], a3 = x[r3 * ld + c];
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(a0), fabsf(a1))), fmaxf(fabsf(a2), fabsf(a3)));
      }
    }
  }
  tr_block_absmax(amax, reinterpret_cast<float*>(dyn_smem), absmax);
}
extern "C" int dyn_train_absmax(const float* x, long rows, int cols, long ld, float* absmax, void* stream) {
  DYN_REQUIRE(x && absmax && rows > 0 && cols > 0 && ld >= cols, "dyn_train_absmax: bad arguments");
  const int ct = cols <= 32 ? 32 : cols <= 64 ? 64 : cols <= 128 ? 128 : 256;
  const int vec = ld == cols && ((rows * cols) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const long blocks = vec ? (rows * cols / 4 + 511) / 512 : ld == cols ? (rows * cols + 1023) / 1024 : (rows + 4 * (256 / ct) - 1) / (4 * (256 / ct));
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_absmax", k_train_absmax, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), sizeof(float4),
             (hipStream_t)stream, x, rows, cols, ld, ct, vec, absmax);
  return 0;
}

// ---- Fourier features of the static net's per-view input (mlp_network.py:423-441; render_ray.py:372-396) ---------------------------
// a0[row] = [PE(pts) (33) | PE(src Pluecker) (66) | ray_diff (4) | 0]  (ld 104);  ref_pe[ray] = PE(ref Pluecker) (66, ld 68);
// mask_eff = mask * (sum(rgb) > 1e-3) when mask_rgb (mlp_network.py:445-448).  No gradient flows into any of these.
__device__ __forceinline__ void tr_embed(float x, float* out, int stride) {  // out[0] = x, out[(1 + f) stride] = cos(2^f x), out[(6 + f) stride] = sin
  out[0] = x;
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    float s, c;
    sincosf((float)(1 << f) * x, &s, &c);
    out[(1 + f) * stride] = c;
    out[(6 + f) * stride] = s;
  }
}
__device__ __forceinline__ void tr_unit3(float x, float y, float z, float& ox, float& oy, float& oz) {
  const float d = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
  ox = x / d; oy = y / d; oz = z / d;
}

__global__ void __launch_bounds__(256) k_train_static_embed(const float* __restrict__ pts, const float* __restrict__ ray_o,
                                                            const float* __restrict__ ray_d, const float* __restrict__ centers, int center_stride,
                                                            const float* __restrict__ ray_diff, const float* __restrict__ rgb_feat,
                                                            const float* __restrict__ mask, int R, int S, int V, int mask_rgb,
                                                            float* __restrict__ a0, float* __restrict__ ref_pe, float* __restrict__ mask_eff) {
  const long N = (long)R * S * V;
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < R) {  // the first R threads also make the per-ray reference features
    const int r = (int)row;
    float d[3], c[6];
    tr_unit3(ray_d[r * 3], ray_d[r * 3 + 1], ray_d[r * 3 + 2], d[0], d[1], d[2]);
    const float ox = ray_o[r * 3], oy = ray_o[r * 3 + 1], oz = ray_o[r * 3 + 2];
    c[0] = d[0]; c[1] = d[1]; c[2] = d[2];
    c[3] = oy * d[2] - oz * d[1];
    c[4] = oz * d[0] - ox * d[2];
    c[5] = ox * d[1] - oy * d[0];
    if (dyn_ref_cross_axis(R) != DYN_CROSS_XYZ) {  // a batch of exactly 3 rays: the reference's torch.cross runs over the rays (csrc/dyn_device.h)
      float m[3];
      dyn_ref_moment_over_rays(ray_o, ray_d, r, m);
      c[3] = m[0]; c[4] = m[1]; c[5] = m[2];
    }
    float* o = ref_pe + (long)r * 68;
    for (int k = 0; k < 6; ++k) tr_embed(c[k], o + k, 6);
    o[66] = 0.f; o[67] = 0.f;
  }
  // the 104-float row of a thread is assembled in LDS ([64 rows][105]: odd stride) and the wave's 64 rows leave as ONE contiguous run of
  // 26 KiB: per-thread scalar stores into 416-byte rows measured 8.8 GB of HBM traffic per launch for 1.7 GB of algorithmic bytes
  float* tile = reinterpret_cast<float*>(dyn_smem) + (threadIdx.x >> 6) * (64 * 105);
  const int lane = threadIdx.x & 63;
  const long wrow0 = row - lane;                       // first row of this wave
  const bool has = row < N;
  const long rowc = has ? row : N - 1;                 // idle lanes shadow the last row (their LDS row is never copied out)
  const long p = rowc / V;
  const int v = (int)(rowc - p * V);
  float* o = tile + lane * 105;
  const float px = pts[p * 3], py = pts[p * 3 + 1], pz = pts[p * 3 + 2];
  tr_embed(px, o + 0, 3);
  tr_embed(py, o + 1, 3);
  tr_embed(pz, o + 2, 3);
  const float cx = centers[v * center_stride], cy = centers[v * center_stride + 1], cz = centers[v * center_stride + 2];
  float c[6];
  tr_unit3(px - cx, py - cy, pz - cz, c[0], c[1], c[2]);
  c[3] = cy * c[2] - cz * c[1];
  c[4] = cz * c[0] - cx * c[2];
  c[5] = cx * c[1] - cy * c[0];
  const int cross_axis = dyn_src_cross_axis(V, R, S);
  if (cross_axis != DYN_CROSS_XYZ) {  // exactly 3 views / rays / samples: the reference's torch.cross runs over that axis, not over xyz (csrc/dyn_device.h)
    const int ray = (int)(p / S), smp = (int)(p - (long)ray * S);
    float m[3];
    dyn_src_moment_over_axis(cross_axis, v, ray, smp,
        [&](int, int rr, int ss, float (&q)[3]) { const float* s3 = pts + ((long)rr * S + ss) * 3; q[0] = s3[0]; q[1] = s3[1]; q[2] = s3[2]; },
        [&](int vv, float (&c3)[3]) { c3[0] = centers[vv * center_stride]; c3[1] = centers[vv * center_stride + 1]; c3[2] = centers[vv * center_stride + 2]; }, m);
    c[3] = m[0]; c[4] = m[1]; c[5] = m[2];
  }
  for (int k = 0; k < 6; ++k) tr_embed(c[k], o + 33 + k, 6);
  const float4 rd = *reinterpret_cast<const float4*>(ray_diff + rowc * 4);
  o[99] = rd.x; o[100] = rd.y; o[101] = rd.z; o[102] = rd.w; o[103] = 0.f;
  if (has) {
    float m = mask[row];
    if (mask_rgb) {
      const float* f = rgb_feat + row * 35;
      m = m * (((f[0] + f[1]) + f[2]) > 1e-3f ? 1.0f : 0.0f);
    }
    mask_eff[row] = m;
  }
  // (the wave's own LDS rows: no workgroup barrier needed, a wave executes in lock step; the fence orders the LDS writes before the reads)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS writes have landed
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  const long nrows = N - wrow0 < 64 ? N - wrow0 : 64;
  if (nrows > 0) {
    float* dst = a0 + wrow0 * 104;
    const int total = (int)nrows * 104;
    for (int t = lane; t < total; t += 64) {
      const int rr = t / 104;
      dst[t] = tile[rr * 105 + (t - rr * 104)];
    }
  }
}

extern "C" int dyn_train_static_embed(const float* pts, const float* ray_o, const float* ray_d, const float* centers, int center_stride,
                                      const float* ray_diff, const float* rgb_feat, const float* mask, int R, int S, int V, int mask_rgb,
                                      float* a0, float* ref_pe, float* mask_eff, void* stream) {
  DYN_REQUIRE(pts && ray_o && ray_d && centers && ray_diff && rgb_feat && mask && a0 && ref_pe && mask_eff, "dyn_train_static_embed: null pointer");
  DYN_REQUIRE(R > 0 && S > 0 && V > 0, "dyn_train_static_embed: empty shape");
  const long N = (long)R * S * V;
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_static_embed", k_train_static_embed, dim3((unsigned)((N + 255) / 256)), dim3(256), (size_t)4 * 64 * 105 * 4, (hipStream_t)stream,
             pts, ray_o, ray_d, centers, center_stride, ray_diff, rgb_feat, mask, R, S, V, mask_rgb, a0, ref_pe, mask_eff);
  return 0;
}

// ---- f = [rgb_feat | src_feat * ref_feat] (mlp_network.py:450) and its backward ---------------------------------------------------------
// forward: one thread per (row, c < 70), f ld 72.  backward: one workgroup per ray (its S V rows are contiguous):
// d src_feat[row, c] = df[row, 35 + c] ref_feat[ray, c]; d ref_feat[ray, c] = sum_rows df[row, 35 + c] src_feat[row, c].
__global__ void __launch_bounds__(256) k_train_build_f(const float* __restrict__ rgb_feat, const float* __restrict__ src_feat, long ld_src,
                                                       const float* __restrict__ ref_feat, long ld_ref, long N, int rows_per_ray,
                                                       float* __restrict__ f) {
  // a thread writes one 16-byte quad of a 72-float row (18 quads): a quarter of the store instructions of the one-element form, which ran at
  // 2.4 TB/s; the 35-float input rows have no 16-byte alignment and are read by element (the lines are shared by the row's threads)
  long row, ray;
  int q;
  if (N * 18 < (1L << 32)) {  // 32-bit index arithmetic (a 64-bit division costs ~100 instructions)
    const unsigned idx = blockIdx.x * 256u + threadIdx.x, r = idx / 18u;
    row = r; q = (int)(idx - r * 18u); ray = r / (unsigned)rows_per_ray;
  } else {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    row = idx / 18; q = (int)(idx - row * 18); ray = row / rows_per_ray;
  }
  if (row >= N) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    v[e] = 0.f;
    if (c < 35) v[e] = rgb_feat[row * 35 + c];
    else if (c < 70) v[e] = src_feat[row * ld_src + c - 35] * ref_feat[ray * ld_ref + c - 35];
  }
  *reinterpret_cast<float4*>(f + row * 72 + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void __launch_bounds__(256) k_train_build_f_bwd(const float* __restrict__ df, long ld_df, const float* __restrict__ src_feat, long ld_src,
                                                           const float* __restrict__ ref_feat, long ld_ref, int rows_per_ray,
                                                           float* __restrict__ dsrc, long ld_dsrc, float* __restrict__ dref, long ld_dref) {
  float* red = reinterpret_cast<float*>(dyn_smem);  // [256]
  const long ray = blockIdx.x;
  const int c = threadIdx.x % 36, slot = threadIdx.x / 36;  // 7 row slots x 36 columns (35 used) = 252 threads
  float acc = 0.f;
  if (slot < 7 && c < 35) {
    const float rf = ref_feat[ray * ld_ref + c];
    for (int i = slot; i < rows_per_ray; i += 7) {
      const long row = ray * rows_per_ray + i;
      const float d = df[row * ld_df + 35 + c];
      dsrc[row * ld_dsrc + c] = d * rf;
      acc += d * src_feat[row * ld_src + c];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 35) {
    float s = 0.f;
    for (int k = 0; k < 7; ++k) s += red[k * 36 + threadIdx.x];
    dref[ray * ld_dref + threadIdx.x] = s;
  }
}
extern "C" int dyn_train_build_f(const float* rgb_feat, const float* src_feat, long ld_src, const float* ref_feat, long ld_ref, long N,
                                 int rows_per_ray, float* f, void* stream) {
  DYN_REQUIRE(rgb_feat && src_feat && ref_feat && f && N > 0 && rows_per_ray > 0, "dyn_train_build_f: bad arguments");
  DYN_REQUIRE(((uintptr_t)f & 15) == 0, "dyn_train_build_f: f must be 16-byte aligned");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_build_f", k_train_build_f, dim3((unsigned)((N * 18 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
             rgb_feat, src_feat, ld_src, ref_feat, ld_ref, N, rows_per_ray, f);
  return 0;
}
extern "C" int dyn_train_build_f_bwd(const float* df, long ld_df, const float* src_feat, long ld_src, const float* ref_feat, long ld_ref, long R,
                                     int rows_per_ray, float* dsrc, long ld_dsrc, float* dref, long ld_dref, void* stream) {
  DYN_REQUIRE(df && src_feat && ref_feat && dsrc && dref && R > 0 && rows_per_ray > 0, "dyn_train_build_f_bwd: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_build_f_bwd", k_train_build_f_bwd, dim3((unsigned)R), dim3(256), 1024, (hipStream_t)stream, df, ld_df,
             src_feat, ld_src, ref_feat, ld_ref, rows_per_ray, dsrc, ld_dsrc, dref, ld_dref);
  return 0;
}

// ---- pooling weights over the views of a point ---------------------------------------------------------------------------------------
// mode 0 (mlp_network.py:452-459): aa: u = (e - min_v e) mask, e = exp(|s| (dot - 1)); else u = mask.  w = u / (sum_v u + 1e-8).
// mode 1 (:470-471, :476): vis = sigmoid(logit) mask; w = vis / (sum_v vis + 1e-8); also wmean = mean_v w and nvalid = sum_v mask.
// One thread per point.  Backward: dw -> (mode 0) ds (one atomic per workgroup), (mode 1) dlogit, with the direct gradient of vis added.
__global__ void __launch_bounds__(256) k_train_view_weights(int mode, const float* __restrict__ in, long in_stride, const float* __restrict__ mask,
                                                            const float* __restrict__ s_param, long P, int V, float* __restrict__ w,
                                                            float* __restrict__ vis_out, long vis_stride, float* __restrict__ wmean, long wmean_stride,
                                                            float* __restrict__ nvalid) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const long r0 = p * V;
  if (mode == 0) {
    if (s_param != nullptr) {
      const float sa = fabsf(s_param[0]);
      // e_v - min e as expm1(a_v) - min expm1(a): the same difference, without rounding e ~ 1 to 6e-8 first (dyn_nets.hip, aa_expm1)
      float mn = INFINITY;
      for (int v = 0; v < V; ++v) mn = fminf(mn, expm1f(sa * (in[(r0 + v) * in_stride] - 1.0f)));
      float sum = 0.f;
      for (int v = 0; v < V; ++v) sum += (expm1f(sa * (in[(r0 + v) * in_stride] - 1.0f)) - mn) * mask[r0 + v];
      for (int v = 0; v < V; ++v) w[r0 + v] = (expm1f(sa * (in[(r0 + v) * in_stride] - 1.0f)) - mn) * mask[r0 + v] / (sum + 1e-8f);
    } else {
      float sum = 0.f;
      for (int v = 0; v < V; ++v) sum += mask[r0 + v];
      for (int v = 0; v < V; ++v) w[r0 + v] = mask[r0 + v] / (sum + 1e-8f);
    }
  } else {
    float sum = 0.f, nv = 0.f;
    for (int v = 0; v < V; ++v) {
      const float vi = tr_sigmoid(in[(r0 + v) * in_stride]) * mask[r0 + v];
      vis_out[(r0 + v) * vis_stride] = vi;
      sum += vi;
      nv += mask[r0 + v];
    }
    float ws = 0.f;
    for (int v = 0; v < V; ++v) {
      const float wv = vis_out[(r0 + v) * vis_stride] / (sum + 1e-8f);
      w[r0 + v] = wv;
      ws += wv;
    }
    wmean[p * wmean_stride] = ws / (float)V;
    nvalid[p] = nv;
  }
}
__global__ void __launch_bounds__(256) k_train_view_weights_bwd(int mode, const float* __restrict__ in, long in_stride, const float* __restrict__ mask,
                                                                const float* __restrict__ s_param, long P, int V, const float* __restrict__ w,
                                                                const float* __restrict__ dw, const float* __restrict__ dvis_direct,
                                                                long dvis_stride, const float* __restrict__ vis, long vis_stride,
                                                                const float* __restrict__ dwmean, long dwmean_stride,
                                                                float* __restrict__ dlogit, long dlogit_stride, float* __restrict__ ds) {
  float* red = reinterpret_cast<float*>(dyn_smem);
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float ds_local = 0.f;
  if (p < P) {
    const long r0 = p * V;
    if (mode == 0) {
      // w_v = u_v / (U + eps), u_v = (e_v - e_min) m_v:  du_v = (dw_v - sum_k dw_k w_k) / (U + eps)
      const float s0 = s_param[0], sa = fabsf(s0);
      float mn = INFINITY, U = 0.f, dot = 0.f;
      int amin = 0;
      for (int v = 0; v < V; ++v) {
        const float e = expm1f(sa * (in[(r0 + v) * in_stride] - 1.0f));
        if (e < mn) { mn = e; amin = v; }
      }
      for (int v = 0; v < V; ++v) {
        U += (expm1f(sa * (in[(r0 + v) * in_stride] - 1.0f)) - mn) * mask[r0 + v];
        dot += dw[r0 + v] * w[r0 + v];
      }
      // d/d|s| of u_v = (e_v - e_min) m_v is (e_v x_v - e_min x_min) m_v, x = dot - 1: formed per view as one difference of
      // neighbours rather than as two large sums that cancel
      const float xm = in[(r0 + amin) * in_stride] - 1.0f;
      const float em = expf(sa * xm) * xm;
      float dsa = 0.f;
      for (int v = 0; v < V; ++v) {
        const float du = (dw[r0 + v] - dot) / (U + 1e-8f);
        const float x = in[(r0 + v) * in_stride] - 1.0f;
        dsa += du * mask[r0 + v] * (expf(sa * x) * x - em);
      }
      ds_local = dsa * (s0 > 0.f ? 1.0f : (s0 < 0.f ? -1.0f : 0.0f));
    } else {
      float U = 0.f, dot = 0.f;
      const float dwm = dwmean != nullptr ? dwmean[p * dwmean_stride] / (float)V : 0.f;
      for (int v = 0; v < V; ++v) {
        U += vis[(r0 + v) * vis_stride];
        dot += (dw[r0 + v] + dwm) * w[r0 + v];
      }
      for (int v = 0; v < V; ++v) {
        float dv = (dw[r0 + v] + dwm - dot) / (U + 1e-8f);
        if (dvis_direct != nullptr) dv += dvis_direct[(r0 + v) * dvis_stride];
        const float sg = tr_sigmoid(in[(r0 + v) * in_stride]);
        dlogit[(r0 + v) * dlogit_stride] = dv * mask[r0 + v] * sg * (1.0f - sg);
      }
    }
  }
  if (mode == 0) {
    red[threadIdx.x] = ds_local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(ds, red[0]);
  }
}
extern "C" int dyn_train_view_weights(int mode, const float* in, long in_stride, const float* mask, const float* s_param, long P, int V, float* w,
                                      float* vis_out, long vis_stride, float* wmean, long wmean_stride, float* nvalid, void* stream) {
  DYN_REQUIRE(mask && w && P > 0 && V > 0, "dyn_train_view_weights: bad arguments");
  DYN_REQUIRE(mode == 0 ? (s_param == nullptr || in != nullptr) : (in && vis_out && wmean && nvalid), "dyn_train_view_weights: missing arrays for mode %d", mode);
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_view_weights", k_train_view_weights, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
             mode, in, in_stride, mask, s_param, P, V, w, vis_out, vis_stride, wmean, wmean_stride, nvalid);
  return 0;
}
extern "C" int dyn_train_view_weights_bwd(int mode, const float* in, long in_stride, const float* mask, const float* s_param, long P, int V,
                                          const float* w, const float* dw, const float* dvis_direct, long dvis_stride, const float* vis,
                                          long vis_stride, const float* dwmean, long dwmean_stride, float* dlogit, long dlogit_stride, float* ds,
                                          void* stream) {
  DYN_REQUIRE(in && mask && w && dw && P > 0 && V > 0, "dyn_train_view_weights_bwd: bad arguments");
  DYN_REQUIRE(mode == 0 ? (s_param && ds) : (vis && dlogit), "dyn_train_view_weights_bwd: missing arrays for mode %d", mode);
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_view_weights_bwd", k_train_view_weights_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 1024,
             (hipStream_t)stream, mode, in, in_stride, mask, s_param, P, V, w, dw, dvis_direct, dvis_stride, vis, vis_stride, dwmean, dwmean_stride,
             dlogit, dlogit_stride, ds);
  return 0;
}

// ---- weighted mean / variance over the views of a point (mlp_network.py:115-119) ------------------------------------------------------
// one thread per (point, column).  backward: dX (+)= w (dmean_t + 2 (x - mean) dvar), dmean_t = dmean - 2 dvar sum_v w (x - mean);
// dw[row] = sum_c [x dmean_t + (x - mean)^2 dvar] by a wave reduction over the columns (one wavefront per point).
__global__ void __launch_bounds__(256) k_train_meanvar(const float* __restrict__ x, long ldx, const float* __restrict__ w, long P, int V, int C,
                                                       float* __restrict__ mean, float* __restrict__ var, long ld_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long p = idx / C;
  const int c = (int)(idx - p * C);
  if (p >= P) return;
  const long r0 = p * V;
  float m = 0.f;
  for (int v = 0; v < V; ++v) m += x[(r0 + v) * ldx + c] * w[r0 + v];
  float s = 0.f;
  for (int v = 0; v < V; ++v) {
    const float d = x[(r0 + v) * ldx + c] - m;
    s += w[r0 + v] * (d * d);
  }
  mean[p * ld_out + c] = m;
  var[p * ld_out + c] = s;
}
__global__ void __launch_bounds__(256) k_train_meanvar_bwd(const float* __restrict__ x, long ldx, const float* __restrict__ w, long P, int V, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ dmean,
                                                           const float* __restrict__ dvar, long ld_stat, float* __restrict__ dx, long ld_dx,
                                                           int accumulate, float* __restrict__ dw, int dw_accumulate) {
  // one wavefront per point, lane = column (+ 64, + 128, + 192: C <= 256); the per-view sums over the columns are wave reductions
  const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= P) return;
  const long r0 = p * V;
  float m[4], dmt[4], dv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = lane + 64 * q;
    m[q] = 0.f; dmt[q] = 0.f; dv[q] = 0.f;
    if (c < C) {
      m[q] = mean[p * ld_stat + c];
      dv[q] = dvar[p * ld_stat + c];
      float sw = 0.f;
      for (int v = 0; v < V; ++v) sw += w[r0 + v] * (x[(r0 + v) * ldx + c] - m[q]);
      dmt[q] = dmean[p * ld_stat + c] - 2.0f * dv[q] * sw;
    }
  }
  for (int v = 0; v < V; ++v) {
    const float wv = w[r0 + v];
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = lane + 64 * q;
      if (c < C) {
        const float xv = x[(r0 + v) * ldx + c], d = xv - m[q];
        const float g = wv * (dmt[q] + 2.0f * d * dv[q]);
        float* o = dx + (r0 + v) * ld_dx + c;
        if (accumulate) *o += g; else *o = g;
        part += xv * dmt[q] + d * d * dv[q];
      }
    }
    part = wave_sum(part);
    if (lane == 0) {
      if (dw_accumulate) dw[r0 + v] += part; else dw[r0 + v] = part;
    }
  }
}
// The same two kernels for V <= 16 views (every configuration the reference ships) with a point's rows held in registers: one pass
// over x, and all of a thread's loads issued before the first use (clamped rows with weight 0 stand in for the views beyond V).
template <int VMAX>
__global__ void __launch_bounds__(256) k_train_meanvar_reg(const float* __restrict__ x, long ldx, const float* __restrict__ w, long P, int V, int C,
                                                           float* __restrict__ mean, float* __restrict__ var, long ld_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long p = idx / C;
  const int c = (int)(idx - p * C);
  if (p >= P) return;
  const long r0 = p * V;
  float xv[VMAX], wv[VMAX];
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    const long row = r0 + (v < V ? v : 0);
    xv[v] = x[row * ldx + c];
    wv[v] = w[row];
  }
  float m = 0.f;
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    wv[v] = v < V ? wv[v] : 0.f;
    m += xv[v] * wv[v];
  }
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    const float d = xv[v] - m;
    sq += wv[v] * (d * d);
  }
  mean[p * ld_out + c] = m;
  var[p * ld_out + c] = sq;
}
template <int VMAX, int Q>
__global__ void __launch_bounds__(256) k_train_meanvar_bwd_reg(const float* __restrict__ x, long ldx, const float* __restrict__ w, long P, int V, int C,
                                                               const float* __restrict__ mean, const float* __restrict__ dmean,
                                                               const float* __restrict__ dvar, long ld_stat, float* __restrict__ dx, long ld_dx,
                                                               int accumulate, float* __restrict__ dw, int dw_accumulate) {
  const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= P) return;
  const long r0 = p * V;
  float xv[Q][VMAX], old[Q][VMAX], wv[VMAX], m[Q], dmn[Q], dv[Q];
  int cc[Q];
  bool ok[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    ok[q] = lane + 64 * q < C;
    cc[q] = ok[q] ? lane + 64 * q : 0;  // idle lanes shadow column 0 (their results are dropped)
    m[q] = mean[p * ld_stat + cc[q]];
    dv[q] = dvar[p * ld_stat + cc[q]];
    dmn[q] = dmean[p * ld_stat + cc[q]];
#pragma unroll
    for (int v = 0; v < VMAX; ++v) {
      const long row = r0 + (v < V ? v : 0);
      xv[q][v] = x[row * ldx + cc[q]];
      old[q][v] = accumulate ? dx[row * ld_dx + cc[q]] : 0.f;
    }
  }
#pragma unroll
  for (int v = 0; v < VMAX; ++v) wv[v] = v < V ? w[r0 + v] : 0.f;
  float dmt[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    float sw = 0.f;
#pragma unroll
    for (int v = 0; v < VMAX; ++v) sw += wv[v] * (xv[q][v] - m[q]);
    dmt[q] = dmn[q] - 2.0f * dv[q] * sw;
  }
#pragma unroll
  for (int v = 0; v < VMAX; ++v) {
    if (v < V) {
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float d = xv[q][v] - m[q];
        const float g = wv[v] * (dmt[q] + 2.0f * d * dv[q]);
        if (ok[q]) {
          dx[(r0 + v) * ld_dx + cc[q]] = old[q][v] + g;
          part += xv[q][v] * dmt[q] + d * d * dv[q];
        }
      }
      part = wave_sum(part);
      if (lane == 0) {
        if (dw_accumulate) dw[r0 + v] += part; else dw[r0 + v] = part;
      }
    }
  }
}
extern "C" int dyn_train_meanvar(const float* x, long ldx, const float* w, long P, int V, int C, float* mean, float* var, long ld_out, void* stream) {
  DYN_REQUIRE(x && w && mean && var && P > 0 && V > 0 && C > 0, "dyn_train_meanvar: bad arguments");
  if (V <= 16) {
    DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_meanvar", k_train_meanvar_reg<16>, dim3((unsigned)((P * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
               x, ldx, w, P, V, C, mean, var, ld_out);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_meanvar", k_train_meanvar, dim3((unsigned)((P * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, w,
             P, V, C, mean, var, ld_out);
  return 0;
}
extern "C" int dyn_train_meanvar_bwd(const float* x, long ldx, const float* w, long P, int V, int C, const float* mean, const float* dmean,
                                     const float* dvar, long ld_stat, float* dx, long ld_dx, int accumulate, float* dw, int dw_accumulate,
                                     void* stream) {
  DYN_REQUIRE(x && w && mean && dmean && dvar && dx && dw && P > 0 && V > 0 && C > 0 && C <= 256, "dyn_train_meanvar_bwd: bad arguments (C <= 256)");
  if (V <= 16 && C <= 128) {
    DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_meanvar_bwd", (k_train_meanvar_bwd_reg<16, 2>), dim3((unsigned)((P + 3) / 4)), dim3(256), 0,
               (hipStream_t)stream, x, ldx, w, P, V, C, mean, dmean, dvar, ld_stat, dx, ld_dx, accumulate, dw, dw_accumulate);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_meanvar_bwd", k_train_meanvar_bwd, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx,
             w, P, V, C, mean, dmean, dvar, ld_stat, dx, ld_dx, accumulate, dw, dw_accumulate);
  return 0;
}

// ---- y[row, :] = x[row, :] * s[row]; backward dx (+)= dy s, ds[row] (+)= sum_c dy x: one wavefront per row ---------------------------
__global__ void __launch_bounds__(256) k_train_rowscale(const float* __restrict__ x, long ldx, const float* __restrict__ s, long s_stride, long N, int C,
                                                        float* __restrict__ y, long ldy) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx / C;
  const int c = (int)(idx - row * C);
  if (row >= N) return;
  y[row * ldy + c] = x[row * ldx + c] * s[row * s_stride];
}
// the same for rows of 4 * 2^k floats at 16-byte alignment: one 16-byte access per thread, no 64-bit division
__global__ void __launch_bounds__(256) k_train_rowscale4(const float4* __restrict__ x, int ldx4, const float* __restrict__ s, long s_stride, long N,
                                                         int c4_shift, float4* __restrict__ y, int ldy4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx >> c4_shift;
  const int c = (int)(idx & ((1 << c4_shift) - 1));
  if (row >= N) return;
  const float4 v = x[row * ldx4 + c];
  const float f = s[row * s_stride];
  y[row * ldy4 + c] = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
}
__global__ void __launch_bounds__(256) k_train_rowscale_bwd(const float* __restrict__ dy, long ld_dy, const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ s, long s_stride, long N, int C, float* __restrict__ dx,
                                                            long ld_dx, int accumulate, float* __restrict__ ds, long ds_stride, int ds_accumulate) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float sv = s[row * s_stride];
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = dy[row * ld_dy + c];
    acc += d * x[row * ldx + c];
    float* o = dx + row * ld_dx + c;
    if (accumulate) *o += d * sv; else *o = d * sv;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if (ds_accumulate) ds[row * ds_stride] += acc; else ds[row * ds_stride] = acc;
  }
}
extern "C" int dyn_train_rowscale(const float* x, long ldx, const float* s, long s_stride, long N, int C, float* y, long ldy, void* stream) {
  DYN_REQUIRE(x && s && y && N > 0 && C > 0, "dyn_train_rowscale: bad arguments");
  const int c4 = C / 4;
  if ((C & 3) == 0 && (c4 & (c4 - 1)) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
    int sh = 0;
    while ((1 << sh) < c4) ++sh;
    DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_rowscale", k_train_rowscale4, dim3((unsigned)((N * c4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const float4*>(x), (int)(ldx / 4), s, s_stride, N, sh, reinterpret_cast<float4*>(y), (int)(ldy / 4));
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_rowscale", k_train_rowscale, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, s,
             s_stride, N, C, y, ldy);
  return 0;
}
// 16-byte accesses: L = C / 4 lanes (a power of two) span a row, so a wave covers 64 / L rows per instruction and the row's dot product is
// a shuffle reduction inside its lane group
__global__ void __launch_bounds__(256) k_train_rowscale_bwd4(const float4* __restrict__ dy, long ld_dy4, const float4* __restrict__ x, long ldx4,
                                                             const float* __restrict__ s, long s_stride, long N, int sh, float4* __restrict__ dx,
                                                             long ld_dx4, int accumulate, float* __restrict__ ds, long ds_stride, int ds_accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row_raw = idx >> sh;
  const bool live = row_raw < N;
  const long row = live ? row_raw : N - 1;  // every lane takes part in the shuffles
  const int L = 1 << sh, q = (int)(idx & (L - 1));
  const float4 d = dy[row * ld_dy4 + q], xv = x[row * ldx4 + q];
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (accumulate) o = dx[row * ld_dx4 + q];
  const float sv = s[row * s_stride];
  float dsv = 0.f;
  if (ds_accumulate) dsv = ds[row * ds_stride];
  o.x += d.x * sv; o.y += d.y * sv; o.z += d.z * sv; o.w += d.w * sv;
  if (live) dx[row * ld_dx4 + q] = o;
  float acc = (d.x * xv.x + d.y * xv.y) + (d.z * xv.z + d.w * xv.w);
  for (int off = L >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (live && q == 0) ds[row * ds_stride] = dsv + acc;
}
extern "C" int dyn_train_rowscale_bwd(const float* dy, long ld_dy, const float* x, long ldx, const float* s, long s_stride, long N, int C, float* dx,
                                      long ld_dx, int accumulate, float* ds, long ds_stride, int ds_accumulate, void* stream) {
  DYN_REQUIRE(dy && x && s && dx && ds && N > 0 && C > 0, "dyn_train_rowscale_bwd: bad arguments");
  const int c4 = C / 4;
  if ((C & 3) == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0 && ((ld_dy | ldx | ld_dx) & 3) == 0 &&
      (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) == 0) {
    int sh = 0;
    while ((1 << sh) < c4) ++sh;
    DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_rowscale_bwd", k_train_rowscale_bwd4, dim3((unsigned)((N * c4 + 255) / 256)), dim3(256), 0,
               (hipStream_t)stream, reinterpret_cast<const float4*>(dy), ld_dy / 4, reinterpret_cast<const float4*>(x), ldx / 4, s, s_stride, N, sh,
               reinterpret_cast<float4*>(dx), ld_dx / 4, accumulate, ds, ds_stride, ds_accumulate);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_rowscale_bwd", k_train_rowscale_bwd, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy,
             x, ldx, s, s_stride, N, C, dx, ld_dx, accumulate, ds, ds_stride, ds_accumulate);
  return 0;
}

// ---- vis_fc output split (mlp_network.py:466-469): x2 = x1 + xv[:, :128]; vis0 = sigmoid(xv[:, 128]) mask -----------------------------
// backward: dxv[:, :128] = dx2; dxv[:, 128] = dvis0 mask sig (1 - sig)   (dx1 += dx2 is the caller's accumulate of the same array)
__global__ void __launch_bounds__(256) k_train_vis_split(const float* __restrict__ x1, long ld1, const float* __restrict__ xv, long ldv,
                                                         const float* __restrict__ mask, const float* __restrict__ ray_diff, long N,
                                                         float* __restrict__ x2, long ld2, float* __restrict__ vis0) {
  // one thread per four columns (all leading dimensions are multiples of four floats: checked by the host wrapper)
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx >> 5;
  const int c = (int)(idx & 31) * 4;
  if (row >= N) return;
  const float4 a = *reinterpret_cast<const float4*>(x1 + row * ld1 + c), b = *reinterpret_cast<const float4*>(xv + row * ldv + c);
  *reinterpret_cast<float4*>(x2 + row * ld2 + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  if (c == 0) vis0[row] = tr_sigmoid(xv[row * ldv + 128]) * mask[row];
  if (ray_diff != nullptr && c == 4) {  // columns 129..132 = ray_diff, 133..135 = 0 (column 128, vis, is written later)
    const float4 rd = *reinterpret_cast<const float4*>(ray_diff + row * 4);
    float* o = x2 + row * ld2 + 129;
    o[0] = rd.x; o[1] = rd.y; o[2] = rd.z; o[3] = rd.w; o[4] = 0.f; o[5] = 0.f; o[6] = 0.f;
  }
}
__global__ void __launch_bounds__(256) k_train_vis_split_bwd(const float* __restrict__ dx2, long ld_dx2, const float* __restrict__ dvis0,
                                                             const float* __restrict__ xv, long ldv, const float* __restrict__ mask, long N,
                                                             float* __restrict__ dxv, long ld_dxv) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx >> 5;
  const int c = (int)(idx & 31) * 4;
  if (row >= N) return;
  *reinterpret_cast<float4*>(dxv + row * ld_dxv + c) = *reinterpret_cast<const float4*>(dx2 + row * ld_dx2 + c);
  if (c == 0) {
    const float sg = tr_sigmoid(xv[row * ldv + 128]);
    dxv[row * ld_dxv + 128] = dvis0[row] * mask[row] * sg * (1.0f - sg);
  }
}
// ---- two passes in one: the row-scale / visibility-split backward followed by the activation derivative of the layer whose output the
// gradient has now been collected for (round 3; before: `..._bwd` wrote dX and dyn_train_act_bwd read and rewrote it -- three passes over
// the row matrix saved per call, and for the 129-column vis_fc.2 the scalar form of that kernel).  128 columns as 32 float4 lanes per
// row; a block's eight lane groups own consecutive spans of rows, four rows per trip (every load of a trip in flight before the first use);
// bias gradient: the groups' partial column sums meet in LDS, one atomic per column and block; largest |result| -> *absmax. ----
#ifndef TR_FUSE_SPAN
#define TR_FUSE_SPAN 256  // rows per lane group: 2048 rows per block (same-address atomics per column: see k_train_act_bwd)
#endif
__device__ __forceinline__ float tr_dact(float y, int act) { return y > 0.f ? 1.0f : (act == 1 ? y + 1.0f : 0.f); }

// dx = (dx + dy * s[row]) * act'(x)   (x: the saved OUTPUT of the activation, which is also what s multiplied in the forward pass),
// ds[row] (=|+=) <dy[row], x[row]>
__global__ void __launch_bounds__(256) k_train_rowscale_act_bwd4(const float4* __restrict__ dy, long ld_dy4, const float4* __restrict__ x, long ldx4,
                                                                 const float* __restrict__ s, long s_stride, long N, float4* __restrict__ dx,
                                                                 long ld_dx4, float* __restrict__ ds, long ds_stride, int ds_accumulate, int act,
                                                                 float* __restrict__ dbias, float* __restrict__ absmax) {
  float4* part = dyn_smem;  // [8][32] partial column sums
  const int g = threadIdx.x >> 5, q = threadIdx.x & 31;
  const long ra = ((long)blockIdx.x * 8 + g) * TR_FUSE_SPAN;
  const long rb = ra + TR_FUSE_SPAN < N ? ra + TR_FUSE_SPAN : N;
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float amax = 0.f;
  // the trip count is the same for the two lane groups of a wave (its shuffles need every lane): rows beyond a group's span are clamped
  // (loaded again, never stored)
  const long ra_even = ((long)blockIdx.x * 8 + (g & ~1)) * TR_FUSE_SPAN;
  for (int t = 0; t < TR_FUSE_SPAN && ra_even + t < N; t += 4) {
    const long r = ra + t;
    float4 d[4], xv[4], o[4];
    float sv[4], dsv[4];
    long rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rr[u] = r + u < N ? r + u : N - 1;
      d[u] = dy[rr[u] * ld_dy4 + q];
      xv[u] = x[rr[u] * ldx4 + q];
      o[u] = dx[rr[u] * ld_dx4 + q];
      sv[u] = s[rr[u] * s_stride];
      dsv[u] = ds_accumulate ? ds[rr[u] * ds_stride] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dot = (d[u].x * xv[u].x + d[u].y * xv[u].y) + (d[u].z * xv[u].z + d[u].w * xv[u].w);
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
      float4 v = make_float4(o[u].x + d[u].x * sv[u], o[u].y + d[u].y * sv[u], o[u].z + d[u].z * sv[u], o[u].w + d[u].w * sv[u]);
      if (act != 0) {
        v.x *= tr_dact(xv[u].x, act); v.y *= tr_dact(xv[u].y, act); v.z *= tr_dact(xv[u].z, act); v.w *= tr_dact(xv[u].w, act);
      }
      if (r + u < rb) {
        dx[rr[u] * ld_dx4 + q] = v;
        if (q == 0) ds[rr[u] * ds_stride] = dsv[u] + dot;
        colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
  }
  part[g * 32 + q] = colsum;
  __syncthreads();
  if (dbias != nullptr && threadIdx.x < 128) {
    const float* pf = reinterpret_cast<const float*>(part);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += pf[k * 128 + threadIdx.x];
    atomicAdd(dbias + threadIdx.x, t);
  }
  if (absmax != nullptr) tr_block_absmax(amax, reinterpret_cast<float*>(dyn_smem + 256), absmax);
}
extern "C" int dyn_train_rowscale_act_bwd(const float* dy, long ld_dy, const float* x, long ldx, const float* s, long s_stride, long N, float* dx,
                                          long ld_dx, float* ds, long ds_stride, int ds_accumulate, int act, float* dbias, float* absmax,
                                          void* stream) {
  DYN_REQUIRE(dy && x && s && dx && ds && N > 0, "dyn_train_rowscale_act_bwd: bad arguments");
  DYN_REQUIRE(act >= 0 && act <= 2, "dyn_train_rowscale_act_bwd: act %d (0 none, 1 ELU, 2 ReLU)", act);
  DYN_REQUIRE(((ld_dy | ldx | ld_dx) & 3) == 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) == 0,
              "dyn_train_rowscale_act_bwd: 128 columns in 16-byte-aligned rows (leading dimensions multiples of 4 floats)");
  const long blocks = (N + 8L * TR_FUSE_SPAN - 1) / (8L * TR_FUSE_SPAN);
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_rowscale_act_bwd", k_train_rowscale_act_bwd4, dim3((unsigned)blocks), dim3(256), 257 * sizeof(float4),
             (hipStream_t)stream, reinterpret_cast<const float4*>(dy), ld_dy / 4, reinterpret_cast<const float4*>(x), ldx / 4, s, s_stride, N,
             reinterpret_cast<float4*>(dx), ld_dx / 4, ds, ds_stride, ds_accumulate, act, dbias, absmax);
  return 0;
}

// dxv[:, 0:128] = dx2 * ELU'(xv[:, 0:128]),  dxv[:, 128] = dvis0 * mask * sigmoid'(xv[:, 128]) * ELU'(xv[:, 128]): the split's backward
// and the ELU of vis_fc.2 (129 outputs) in one pass; dbias[129] += column sums.  With dxs != NULL the row-scale backward in front of it
// (x * vis, mlp_network.py:474) rides along as well: dx2 += dxs * vis0[row] first (written back: dx2 goes on collecting the gradient of x),
// and dvis0[row] = <dxs[row], x2[row]> is formed in the pass instead of read.
__global__ void __launch_bounds__(256) k_train_vis_split_act_bwd4(float4* __restrict__ dx2, long ld_dx24, const float* __restrict__ dvis0,
                                                                  const float* __restrict__ xv, long ldv, const float* __restrict__ mask, long N,
                                                                  float* __restrict__ dxv, long ld_dxv, float* __restrict__ dbias,
                                                                  float* __restrict__ absmax, const float4* __restrict__ dxs, long ld_dxs4,
                                                                  const float4* __restrict__ x2, long ldx24, const float* __restrict__ vis0) {
  float4* part = dyn_smem;                                        // [8][32] partial column sums
  float* part_v = reinterpret_cast<float*>(dyn_smem + 256) + 8;   // [8] partial sums of column 128 (behind tr_block_absmax's four floats)
  const int g = threadIdx.x >> 5, q = threadIdx.x & 31;
  const long ra = ((long)blockIdx.x * 8 + g) * TR_FUSE_SPAN;
  const long rb = ra + TR_FUSE_SPAN < N ? ra + TR_FUSE_SPAN : N;
  const long ra_even = ((long)blockIdx.x * 8 + (g & ~1)) * TR_FUSE_SPAN;  // (the same trip count for the two lane groups of a wave: shuffles)
  const bool fused = dxs != nullptr;
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float vsum = 0.f, amax = 0.f;
  for (int t = 0; t < TR_FUSE_SPAN && ra_even + t < N; t += 4) {
    const long r = ra + t;
    float4 d[4], y[4], ds[4], xx[4];
    float yv[4], dv[4], mk[4], vs[4];
    long rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rr[u] = r + u < N ? r + u : N - 1;
      d[u] = dx2[rr[u] * ld_dx24 + q];
      y[u] = *reinterpret_cast<const float4*>(xv + rr[u] * ldv + 4 * q);
      if (fused) {
        ds[u] = dxs[rr[u] * ld_dxs4 + q];
        xx[u] = x2[rr[u] * ldx24 + q];
        vs[u] = vis0[rr[u]];
      }
      if (q == 0) { yv[u] = xv[rr[u] * ldv + 128]; mk[u] = mask[rr[u]]; dv[u] = fused ? 0.f : dvis0[rr[u]]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = r + u < rb;
      float dot = 0.f;
      if (fused) {
        dot = (ds[u].x * xx[u].x + ds[u].y * xx[u].y) + (ds[u].z * xx[u].z + ds[u].w * xx[u].w);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        d[u].x += ds[u].x * vs[u]; d[u].y += ds[u].y * vs[u]; d[u].z += ds[u].z * vs[u]; d[u].w += ds[u].w * vs[u];
        if (live) dx2[rr[u] * ld_dx24 + q] = d[u];
      }
      if (!live) continue;
      const float4 v = make_float4(d[u].x * tr_dact(y[u].x, 1), d[u].y * tr_dact(y[u].y, 1), d[u].z * tr_dact(y[u].z, 1), d[u].w * tr_dact(y[u].w, 1));
      *reinterpret_cast<float4*>(dxv + (r + u) * ld_dxv + 4 * q) = v;
      colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      if (q == 0) {
        const float sg = tr_sigmoid(yv[u]);
        const float w = (fused ? dot : dv[u]) * mk[u] * sg * (1.0f - sg) * tr_dact(yv[u], 1);
        dxv[(r + u) * ld_dxv + 128] = w;
        vsum += w;
        amax = fmaxf(amax, fabsf(w));
      }
    }
  }
  part[g * 32 + q] = colsum;
  if (q == 0) part_v[g] = vsum;
  __syncthreads();
  if (dbias != nullptr && threadIdx.x < 128) {
    const float* pf = reinterpret_cast<const float*>(part);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += pf[k * 128 + threadIdx.x];
    atomicAdd(dbias + threadIdx.x, t);
  }
  if (dbias != nullptr && threadIdx.x == 128) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part_v[k];
    atomicAdd(dbias + 128, t);
  }
  if (absmax != nullptr) tr_block_absmax(amax, reinterpret_cast<float*>(dyn_smem + 256), absmax);
}
extern "C" int dyn_train_vis_split_act_bwd(float* dx2, long ld_dx2, const float* dvis0, const float* xv, long ldv, const float* mask, long N,
                                           float* dxv, long ld_dxv, float* dbias, float* absmax, const float* dxs, long ld_dxs, const float* x2,
                                           long ldx2, const float* vis0, void* stream) {
  DYN_REQUIRE(dx2 && xv && mask && dxv && N > 0 && (dvis0 != nullptr || dxs != nullptr), "dyn_train_vis_split_act_bwd: bad arguments");
  DYN_REQUIRE(((ld_dx2 | ldv | ld_dxv) & 3) == 0 && ldv >= 129 && ld_dxv >= 129 && (((uintptr_t)dx2 | (uintptr_t)xv | (uintptr_t)dxv) & 15) == 0,
              "dyn_train_vis_split_act_bwd: rows must be 16-byte aligned (leading dimensions multiples of 4 floats, >= 129 for xv / dxv)");
  DYN_REQUIRE(dxs == nullptr || (x2 && vis0 && ((ld_dxs | ldx2) & 3) == 0 && (((uintptr_t)dxs | (uintptr_t)x2) & 15) == 0),
              "dyn_train_vis_split_act_bwd: the row-scale part needs dxs, x2 (16-byte-aligned rows) and vis0");
  const long blocks = (N + 8L * TR_FUSE_SPAN - 1) / (8L * TR_FUSE_SPAN);
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_vis_split_act_bwd", k_train_vis_split_act_bwd4, dim3((unsigned)blocks), dim3(256), 260 * sizeof(float4),
             (hipStream_t)stream, reinterpret_cast<float4*>(dx2), ld_dx2 / 4, dvis0, xv, ldv, mask, N, dxv, ld_dxv, dbias, absmax,
             reinterpret_cast<const float4*>(dxs), ld_dxs / 4, reinterpret_cast<const float4*>(x2), ldx2 / 4, vis0);
  return 0;
}

// y[row] = <x[row, 0:C], w> + b: the forward of a Linear with ONE output (vis_fc2.2, rgb_fc.4 of the static net, out_geometry_fc.2) as a
// row kernel -- as a GEMM it ran a 128-column tile for one useful column.  C = 4 L columns (L a power of two lanes per row), four rows
// per lane group and trip, the row's dot product by a shuffle reduction inside its lane group.
__global__ void __launch_bounds__(256) k_train_rowdot4(const float4* __restrict__ x, long ldx4, const float4* __restrict__ w, const float* __restrict__ bias,
                                                       long N, int sh, float* __restrict__ y, long y_stride) {
  const int L = 1 << sh, G = 256 >> sh, g = threadIdx.x >> sh, q = threadIdx.x & (L - 1);
  const float4 wq = w[q];
  const float b = bias != nullptr ? bias[0] : 0.f;
  const long r0 = ((long)blockIdx.x * G + g) * 4;
  float4 xv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long rr = r0 + u < N ? r0 + u : N - 1;  // every lane takes part in the shuffles
    xv[u] = x[rr * ldx4 + q];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float dot = (xv[u].x * wq.x + xv[u].y * wq.y) + (xv[u].z * wq.z + xv[u].w * wq.w);
    for (int off = L >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if (q == 0 && r0 + u < N) y[(r0 + u) * y_stride] = dot + b;
  }
}
extern "C" int dyn_train_rowdot(const float* X, long ldx, const float* w, const float* bias, long N, int C, float* y, long y_stride, void* stream) {
  DYN_REQUIRE(X && w && y && N > 0 && C > 0, "dyn_train_rowdot: bad arguments");
  const int c4 = C / 4;
  DYN_REQUIRE((C & 3) == 0 && c4 >= 1 && c4 <= 64 && (c4 & (c4 - 1)) == 0, "dyn_train_rowdot: %d columns (4, 8, 16, ... 256)", C);
  DYN_REQUIRE((ldx & 3) == 0 && (((uintptr_t)X | (uintptr_t)w) & 15) == 0, "dyn_train_rowdot: rows must be 16-byte aligned");
  int sh = 0;
  while ((1 << sh) < c4) ++sh;
  const long per_block = (long)(256 >> sh) * 4;
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_rowdot", k_train_rowdot4, dim3((unsigned)((N + per_block - 1) / per_block)), dim3(256), 0, (hipStream_t)stream,
             reinterpret_cast<const float4*>(X), ldx / 4, reinterpret_cast<const float4*>(w), bias, N, sh, y, y_stride);
  return 0;
}

// dX[row, :] = dz[row] * w[:] * act'(Y[row, :]): the data gradient of a Linear with ONE output (vis_fc2.2, rgb_fc.4 of the static net,
// out_geometry_fc.2: a rank-one product -- as a GEMM it ran a 128-column tile for one useful column, 4.7 ms per iteration) through the
// activation of the layer in front of it, with that layer's bias gradient (column sums) and the scale of dX.  C = 4 L columns (L a power
// of two, 4 .. 64 lanes per row); a block's 256 / L lane groups own consecutive spans of rows, four rows per trip.
__global__ void __launch_bounds__(256) k_train_outer_act_bwd4(const float* __restrict__ dz, long dz_stride, const float4* __restrict__ w,
                                                              const float4* __restrict__ y, long ld_y4, long N, int sh, int act,
                                                              float4* __restrict__ dx, long ld_dx4, float* __restrict__ dbias,
                                                              float* __restrict__ absmax, float* __restrict__ dw) {
  float4* part = dyn_smem;  // [G][L] partial column sums
  const int L = 1 << sh, G = 256 >> sh, g = threadIdx.x >> sh, q = threadIdx.x & (L - 1);
  const long ra = ((long)blockIdx.x * G + g) * TR_FUSE_SPAN;
  const long rb = ra + TR_FUSE_SPAN < N ? ra + TR_FUSE_SPAN : N;
  const float4 wq = w[q];
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f), wsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float amax = 0.f;
  for (long r = ra; r < rb; r += 4) {
    float4 yv[4];
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = r + u < rb ? r + u : rb - 1;
      d[u] = dz[rr * dz_stride];
      if (act != 0) yv[u] = y[rr * ld_y4 + q];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u >= rb) continue;
      float4 v = make_float4(d[u] * wq.x, d[u] * wq.y, d[u] * wq.z, d[u] * wq.w);
      if (act != 0) {
        // the weight gradient of the one-output layer rides along: its input IS the saved output Y of the layer in front
        wsum.x = fmaf(d[u], yv[u].x, wsum.x); wsum.y = fmaf(d[u], yv[u].y, wsum.y); wsum.z = fmaf(d[u], yv[u].z, wsum.z); wsum.w = fmaf(d[u], yv[u].w, wsum.w);
        v.x *= tr_dact(yv[u].x, act); v.y *= tr_dact(yv[u].y, act); v.z *= tr_dact(yv[u].z, act); v.w *= tr_dact(yv[u].w, act);
      }
      dx[(r + u) * ld_dx4 + q] = v;
      colsum.x += v.x; colsum.y += v.y; colsum.z += v.z; colsum.w += v.w;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  }
  part[g * L + q] = colsum;
  __syncthreads();
  if (dbias != nullptr && threadIdx.x < 4 * L) {
    const float* pf = reinterpret_cast<const float*>(part);
    float t = 0.f;
    for (int k = 0; k < G; ++k) t += pf[k * 4 * L + threadIdx.x];
    atomicAdd(dbias + threadIdx.x, t);
  }
  if (absmax != nullptr) tr_block_absmax(amax, reinterpret_cast<float*>(dyn_smem + 256), absmax);
  if (dw != nullptr) {  // uniform over the block
    __syncthreads();
    part[g * L + q] = wsum;
    __syncthreads();
    if (threadIdx.x < 4 * L) {
      const float* pf = reinterpret_cast<const float*>(part);
      float t = 0.f;
      for (int k = 0; k < G; ++k) t += pf[k * 4 * L + threadIdx.x];
      atomicAdd(dw + threadIdx.x, t);
    }
  }
}
extern "C" int dyn_train_outer_act_bwd(const float* dz, long dz_stride, const float* w, const float* Y, long ld_y, long N, int C, int act,
                                       float* dX, long ld_dx, float* dbias, float* absmax, float* dW, void* stream) {
  DYN_REQUIRE(dz && w && dX && N > 0 && C > 0, "dyn_train_outer_act_bwd: bad arguments");
  DYN_REQUIRE(act == 0 || Y != nullptr, "dyn_train_outer_act_bwd: ELU / ReLU backward needs the saved output");
  DYN_REQUIRE(dW == nullptr || act != 0, "dyn_train_outer_act_bwd: the weight gradient is taken against Y (the layer's input), which needs act != 0");
  const int c4 = C / 4;
  DYN_REQUIRE((C & 3) == 0 && c4 >= 1 && c4 <= 64 && (c4 & (c4 - 1)) == 0, "dyn_train_outer_act_bwd: %d columns (4, 8, 16, ... 256)", C);
  DYN_REQUIRE((ld_dx & 3) == 0 && (act == 0 || (ld_y & 3) == 0) && (((uintptr_t)w | (uintptr_t)dX | (act ? (uintptr_t)Y : 0)) & 15) == 0,
              "dyn_train_outer_act_bwd: rows must be 16-byte aligned (leading dimensions multiples of 4 floats)");
  int sh = 0;
  while ((1 << sh) < c4) ++sh;
  const long per_block = (long)(256 >> sh) * TR_FUSE_SPAN;
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_outer_act_bwd", k_train_outer_act_bwd4, dim3((unsigned)((N + per_block - 1) / per_block)), dim3(256),
             257 * sizeof(float4), (hipStream_t)stream, dz, dz_stride, reinterpret_cast<const float4*>(w),
             reinterpret_cast<const float4*>(act != 0 ? Y : nullptr), ld_y / 4, N, sh, act, reinterpret_cast<float4*>(dX), ld_dx / 4, dbias, absmax, dW);
  return 0;
}

extern "C" int dyn_train_vis_split(const float* x1, long ld1, const float* xv, long ldv, const float* mask, const float* ray_diff, long N, float* x2,
                                   long ld2, float* vis0, void* stream) {
  DYN_REQUIRE(x1 && xv && mask && x2 && vis0 && N > 0, "dyn_train_vis_split: bad arguments");
  DYN_REQUIRE(((ld1 | ldv | ld2) & 3) == 0 && (((uintptr_t)x1 | (uintptr_t)xv | (uintptr_t)x2) & 15) == 0,
              "dyn_train_vis_split: rows must be 16-byte aligned (leading dimensions multiples of 4 floats)");
  DYN_REQUIRE(ray_diff == nullptr || ld2 >= 136, "dyn_train_vis_split: the [x2 | vis | ray_diff | 0 0 0] layout needs ld2 >= 136");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_vis_split", k_train_vis_split, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x1,
             ld1, xv, ldv, mask, ray_diff, N, x2, ld2, vis0);
  return 0;
}
extern "C" int dyn_train_vis_split_bwd(const float* dx2, long ld_dx2, const float* dvis0, const float* xv, long ldv, const float* mask, long N,
                                       float* dxv, long ld_dxv, void* stream) {
  DYN_REQUIRE(dx2 && dvis0 && xv && mask && dxv && N > 0, "dyn_train_vis_split_bwd: bad arguments");
  DYN_REQUIRE(((ld_dx2 | ldv | ld_dxv) & 3) == 0 && (((uintptr_t)dx2 | (uintptr_t)dxv) & 15) == 0,
              "dyn_train_vis_split_bwd: rows must be 16-byte aligned (leading dimensions multiples of 4 floats)");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_vis_split_bwd", k_train_vis_split_bwd, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0,
             (hipStream_t)stream, dx2, ld_dx2, dvis0, xv, ldv, mask, N, dxv, ld_dxv);
  return 0;
}

// ---- ray attention core (mlp_network.py:13-31, :83-97): per (ray, head) softmax(q k^T / sqrt(d_k), query rows masked) v ----------------
// qkv [P, 3 * 128] (q | k | v, head h in columns 32 h .. 32 h + 31 of each), one workgroup of S threads per (ray, head): thread = query
// row, K and V of the head in LDS.  The probabilities are saved for the backward pass.  A query row whose point is seen by fewer than two views
// (mask = num_valid_obs > 1, mlp_network.py:486-488) has every score replaced by
// -1e9 (uniform probabilities, no gradient into its scores).
__global__ void k_train_attn(const float* __restrict__ qkv, const float* __restrict__ nvalid, int S, float* __restrict__ out,
                             float* __restrict__ prob) {
  float* ks = reinterpret_cast<float*>(dyn_smem);  // [S][33]
  float* vs = ks + S * 33;
  const int ray = blockIdx.x >> 2, head = blockIdx.x & 3, i = threadIdx.x;
  const long p = (long)ray * S + i;
  float q[32];
  for (int d = 0; d < 32; ++d) {
    ks[i * 33 + d] = qkv[p * 384 + 128 + head * 32 + d];
    vs[i * 33 + d] = qkv[p * 384 + 256 + head * 32 + d];
    q[d] = qkv[p * 384 + head * 32 + d] / 5.656854249492381f;
  }
  __syncthreads();
  float* pr = prob + ((long)blockIdx.x * S + i) * S;
  const bool live = nvalid[p] > 1.0f;
  float mx = -INFINITY;
  for (int j = 0; j < S; ++j) {
    float s = 0.f;
    for (int d = 0; d < 32; ++d) s += q[d] * ks[j * 33 + d];
    if (!live) s = -1e9f;
    pr[j] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int j = 0; j < S; ++j) {
    const float e = expf(pr[j] - mx);
    pr[j] = e;
    den += e;
  }
  float o[32];
  for (int d = 0; d < 32; ++d) o[d] = 0.f;
  for (int j = 0; j < S; ++j) {
    const float a = pr[j] / den;
    pr[j] = a;
    for (int d = 0; d < 32; ++d) o[d] += a * vs[j * 33 + d];
  }
  for (int d = 0; d < 32; ++d) out[p * 128 + head * 32 + d] = o[d];
}
// backward: thread i first as query row (dP_ij = dO_i . V_j, dS = P (dP - sum_j P dP), dQ_i = sum_j dS_ij K_j / sqrt(d)), dS kept in the
// prob buffer; then as key row (dK_i = sum_q dS_qi Q_q / sqrt(d), dV_i = sum_q P_qi dO_q) -- P is needed again, so dS goes to LDS-free
// scratch: the second half of the prob buffer row is not available, hence dS overwrites a separate array.
__global__ void k_train_attn_bwd(const float* __restrict__ qkv, const float* __restrict__ nvalid, int S, const float* __restrict__ prob,
                                 const float* __restrict__ dout, float* __restrict__ dscore, float* __restrict__ dqkv) {
  float* ks = reinterpret_cast<float*>(dyn_smem);  // [S][33]: K, later Q
  float* vs = ks + S * 33;                         // [S][33]: V, later dO
  const int ray = blockIdx.x >> 2, head = blockIdx.x & 3, i = threadIdx.x;
  const long p = (long)ray * S + i;
  float dO[32];
  for (int d = 0; d < 32; ++d) {
    ks[i * 33 + d] = qkv[p * 384 + 128 + head * 32 + d];
    vs[i * 33 + d] = qkv[p * 384 + 256 + head * 32 + d];
    dO[d] = dout[p * 128 + head * 32 + d];
  }
  __syncthreads();
  const float* pr = prob + ((long)blockIdx.x * S + i) * S;
  float* dsr = dscore + ((long)blockIdx.x * S + i) * S;
  const bool live = nvalid[p] > 1.0f;
  float dot = 0.f;
  for (int j = 0; j < S; ++j) {
    float dp = 0.f;
    for (int d = 0; d < 32; ++d) dp += dO[d] * vs[j * 33 + d];
    dsr[j] = dp;
    dot += dp * pr[j];
  }
  float dq[32];
  for (int d = 0; d < 32; ++d) dq[d] = 0.f;
  for (int j = 0; j < S; ++j) {
    const float ds = live ? pr[j] * (dsr[j] - dot) : 0.f;
    dsr[j] = ds;
    for (int d = 0; d < 32; ++d) dq[d] += ds * ks[j * 33 + d];
  }
#pragma unroll
  for (int d4 = 0; d4 < 8; ++d4)
    reinterpret_cast<float4*>(dqkv + p * 384 + head * 32)[d4] = make_float4(dq[4 * d4] / 5.656854249492381f, dq[4 * d4 + 1] / 5.656854249492381f,
                                                                            dq[4 * d4 + 2] / 5.656854249492381f, dq[4 * d4 + 3] / 5.656854249492381f);
  __syncthreads();  // every thread is done with K and V
  for (int d = 0; d < 32; ++d) {
    ks[i * 33 + d] = qkv[p * 384 + head * 32 + d] / 5.656854249492381f;  // Q / sqrt(d)
    vs[i * 33 + d] = dO[d];
  }
  __threadfence_block();
  __syncthreads();
  float dk[32], dv[32];
  for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  const float* pcol = prob + (long)blockIdx.x * S * S + i;
  const float* dcol = dscore + (long)blockIdx.x * S * S + i;
  for (int qi = 0; qi < S; ++qi) {
    const float a = pcol[(long)qi * S], ds = dcol[(long)qi * S];
    for (int d = 0; d < 32; ++d) {
      dk[d] += ds * ks[qi * 33 + d];
      dv[d] += a * vs[qi * 33 + d];
    }
  }
  for (int d = 0; d < 32; ++d) {
    dqkv[p * 384 + 128 + head * 32 + d] = dk[d];
    dqkv[p * 384 + 256 + head * 32 + d] = dv[d];
  }
}
#define TR_ATTN_LDS_MAX_S 112
// The same two kernels with the S x S score / probability tiles of a (ray, head) kept in LDS (rays of up to 112 samples): the kernels
// above use the global `prob` / `dscore` arrays as per-thread scratch rows -- 4-byte accesses at a stride of S floats, measured at 6-8 GB of
// HBM traffic per launch for 0.6 GB of algorithmic bytes.  Here the probabilities leave once, as coalesced rows, and come back once.
// FOUR lanes per row (a block = 4 S threads): with one thread per row a (ray, head) was ONE wavefront holding 36-52 KiB of LDS -- three
// waves per CU, latency bound (0.4 / 0.9 ms per launch).  Lane c of row i takes the keys (later: the queries) j = c, c + 4, ...; the four
// partial sums meet by two butterfly steps inside the wave.  K, V rows at a 36-float stride (16-byte aligned: an inner product reads four
// values per LDS instruction, and the four rows a wave touches at once sit in disjoint banks); score rows at S + 4.
// (DPP quad_perm lane selects on the VALU: [1,0,3,2] and [2,3,0,1]; no LDS traffic)
__device__ __forceinline__ float tr_quad_get1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float tr_quad_get2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float tr_quad_sum(float v) {
  v += tr_quad_get1(v);
  v += tr_quad_get2(v);
  return v;
}
__global__ void __launch_bounds__(4 * TR_ATTN_LDS_MAX_S) k_train_attn_lds(const float* __restrict__ qkv, const float* __restrict__ nvalid, int S, float* __restrict__ out,
                                 float* __restrict__ prob) {
  float* ks = reinterpret_cast<float*>(dyn_smem);  // [S][36]
  float* vs = ks + S * 36;                         // [S][36]
  float* ps = vs + S * 36;                         // [S][S + 4]
  const int ray = blockIdx.x >> 2, head = blockIdx.x & 3, i = threadIdx.x >> 2, c = threadIdx.x & 3, SP = S + 4;
  const long p = (long)ray * S + i;
  float q[32];
  {  // the row's Q (every lane of the row: the same 128 bytes, one L1 line) and this lane's quarter of its K and V
    const float4* row = reinterpret_cast<const float4*>(qkv + p * 384 + head * 32);
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 qv = row[d4];
      q[4 * d4] = qv.x / 5.656854249492381f; q[4 * d4 + 1] = qv.y / 5.656854249492381f;
      q[4 * d4 + 2] = qv.z / 5.656854249492381f; q[4 * d4 + 3] = qv.w / 5.656854249492381f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<float4*>(ks + i * 36 + 4 * (2 * c + u)) = row[32 + 2 * c + u];
      *reinterpret_cast<float4*>(vs + i * 36 + 4 * (2 * c + u)) = row[64 + 2 * c + u];
    }
  }
  __syncthreads();
  const bool live = nvalid[p] > 1.0f;
  float* pr = ps + i * SP;
  float mx = -INFINITY;
  for (int j = c; j < S; j += 4) {
    float sc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 k4 = *reinterpret_cast<const float4*>(ks + j * 36 + 4 * d4);
      sc += (q[4 * d4] * k4.x + q[4 * d4 + 1] * k4.y) + (q[4 * d4 + 2] * k4.z + q[4 * d4 + 3] * k4.w);
    }
    if (!live) sc = -1e9f;
    pr[j] = sc;
    mx = fmaxf(mx, sc);
  }
  mx = fmaxf(mx, tr_quad_get1(mx));
  mx = fmaxf(mx, tr_quad_get2(mx));
  float den = 0.f;
  for (int j = c; j < S; j += 4) {
    const float e = expf(pr[j] - mx);
    pr[j] = e;
    den += e;
  }
  den = tr_quad_sum(den);
  float o[32];
  for (int d = 0; d < 32; ++d) o[d] = 0.f;
  for (int j = c; j < S; j += 4) {
    const float a = pr[j] / den;
    pr[j] = a;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 v4 = *reinterpret_cast<const float4*>(vs + j * 36 + 4 * d4);
      o[4 * d4] += a * v4.x; o[4 * d4 + 1] += a * v4.y; o[4 * d4 + 2] += a * v4.z; o[4 * d4 + 3] += a * v4.w;
    }
  }
#pragma unroll
  for (int d = 0; d < 32; ++d) o[d] = tr_quad_sum(o[d]);
#pragma unroll
  for (int d4 = 0; d4 < 8; ++d4)
    if ((d4 >> 1) == c) reinterpret_cast<float4*>(out + p * 128 + head * 32)[d4] = make_float4(o[4 * d4], o[4 * d4 + 1], o[4 * d4 + 2], o[4 * d4 + 3]);
  __syncthreads();
  float* g = prob + (long)blockIdx.x * S * S;
  for (int idx = threadIdx.x; idx < S * S; idx += 4 * S) g[idx] = ps[(idx / S) * SP + (idx % S)];  // coalesced rows
}
__global__ void __launch_bounds__(4 * TR_ATTN_LDS_MAX_S) k_train_attn_bwd_lds(const float* __restrict__ qkv, const float* __restrict__ nvalid, int S, const float* __restrict__ prob,
                                     const float* __restrict__ dout, float* __restrict__ dqkv) {
  float* ks = reinterpret_cast<float*>(dyn_smem);  // [S][36]: K, later Q / sqrt(d)
  float* vs = ks + S * 36;                         // [S][36]: V, later dO
  float* ps = vs + S * 36;                         // [S][S + 4]: probabilities
  float* db = ps + S * (S + 4);                    // [S][S + 4]: score gradients
  const int ray = blockIdx.x >> 2, head = blockIdx.x & 3, i = threadIdx.x >> 2, c = threadIdx.x & 3, SP = S + 4;
  const long p = (long)ray * S + i;
  float dO[32], q[32];
  {
    const float4* row = reinterpret_cast<const float4*>(qkv + p * 384 + head * 32);
    const float4* drow = reinterpret_cast<const float4*>(dout + p * 128 + head * 32);
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 qv = row[d4], dv4 = drow[d4];
      q[4 * d4] = qv.x / 5.656854249492381f; q[4 * d4 + 1] = qv.y / 5.656854249492381f;
      q[4 * d4 + 2] = qv.z / 5.656854249492381f; q[4 * d4 + 3] = qv.w / 5.656854249492381f;
      dO[4 * d4] = dv4.x; dO[4 * d4 + 1] = dv4.y; dO[4 * d4 + 2] = dv4.z; dO[4 * d4 + 3] = dv4.w;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<float4*>(ks + i * 36 + 4 * (2 * c + u)) = row[32 + 2 * c + u];
      *reinterpret_cast<float4*>(vs + i * 36 + 4 * (2 * c + u)) = row[64 + 2 * c + u];
    }
  }
  const float* g = prob + (long)blockIdx.x * S * S;
  for (int idx = threadIdx.x; idx < S * S; idx += 4 * S) ps[(idx / S) * SP + (idx % S)] = g[idx];
  __syncthreads();
  // as query row i: dP_ij = dO_i . V_j, dS = P (dP - sum_j P dP), dQ_i = sum_j dS_ij K_j / sqrt(d)
  const bool live = nvalid[p] > 1.0f;
  float dot = 0.f;
  for (int j = c; j < S; j += 4) {
    float dp = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 v4 = *reinterpret_cast<const float4*>(vs + j * 36 + 4 * d4);
      dp += (dO[4 * d4] * v4.x + dO[4 * d4 + 1] * v4.y) + (dO[4 * d4 + 2] * v4.z + dO[4 * d4 + 3] * v4.w);
    }
    db[i * SP + j] = dp;
    dot += dp * ps[i * SP + j];
  }
  dot = tr_quad_sum(dot);
  float acc[32];
  for (int d = 0; d < 32; ++d) acc[d] = 0.f;
  for (int j = c; j < S; j += 4) {
    const float ds = live ? ps[i * SP + j] * (db[i * SP + j] - dot) : 0.f;
    db[i * SP + j] = ds;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 k4 = *reinterpret_cast<const float4*>(ks + j * 36 + 4 * d4);
      acc[4 * d4] += ds * k4.x; acc[4 * d4 + 1] += ds * k4.y; acc[4 * d4 + 2] += ds * k4.z; acc[4 * d4 + 3] += ds * k4.w;
    }
  }
#pragma unroll
  for (int d = 0; d < 32; ++d) acc[d] = tr_quad_sum(acc[d]) / 5.656854249492381f;
#pragma unroll
  for (int d4 = 0; d4 < 8; ++d4)
    if ((d4 >> 1) == c) reinterpret_cast<float4*>(dqkv + p * 384 + head * 32)[d4] = make_float4(acc[4 * d4], acc[4 * d4 + 1], acc[4 * d4 + 2], acc[4 * d4 + 3]);
  __syncthreads();  // every thread is done with K and V
#pragma unroll
  for (int d4 = 0; d4 < 8; ++d4)
    if ((d4 >> 1) == c) {
      *reinterpret_cast<float4*>(ks + i * 36 + 4 * d4) = make_float4(q[4 * d4], q[4 * d4 + 1], q[4 * d4 + 2], q[4 * d4 + 3]);
      *reinterpret_cast<float4*>(vs + i * 36 + 4 * d4) = make_float4(dO[4 * d4], dO[4 * d4 + 1], dO[4 * d4 + 2], dO[4 * d4 + 3]);
    }
  __syncthreads();
  // as key row i: dK_i = sum_q dS_qi Q_q / sqrt(d), dV_i = sum_q P_qi dO_q
  float dk[32], dv[32];
  for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int qi = c; qi < S; qi += 4) {
    const float a = ps[qi * SP + i], ds = db[qi * SP + i];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const float4 q4 = *reinterpret_cast<const float4*>(ks + qi * 36 + 4 * d4), o4 = *reinterpret_cast<const float4*>(vs + qi * 36 + 4 * d4);
      dk[4 * d4] += ds * q4.x; dk[4 * d4 + 1] += ds * q4.y; dk[4 * d4 + 2] += ds * q4.z; dk[4 * d4 + 3] += ds * q4.w;
      dv[4 * d4] += a * o4.x; dv[4 * d4 + 1] += a * o4.y; dv[4 * d4 + 2] += a * o4.z; dv[4 * d4 + 3] += a * o4.w;
    }
  }
#pragma unroll
  for (int d = 0; d < 32; ++d) { dk[d] = tr_quad_sum(dk[d]); dv[d] = tr_quad_sum(dv[d]); }
#pragma unroll
  for (int d4 = 0; d4 < 8; ++d4)
    if ((d4 >> 1) == c) {
      reinterpret_cast<float4*>(dqkv + p * 384 + 128 + head * 32)[d4] = make_float4(dk[4 * d4], dk[4 * d4 + 1], dk[4 * d4 + 2], dk[4 * d4 + 3]);
      reinterpret_cast<float4*>(dqkv + p * 384 + 256 + head * 32)[d4] = make_float4(dv[4 * d4], dv[4 * d4 + 1], dv[4 * d4 + 2], dv[4 * d4 + 3]);
    }
}

extern "C" int dyn_train_attn(const float* qkv, const float* nvalid, int R, int S, float* out, float* prob, void* stream) {
  DYN_REQUIRE(qkv && nvalid && out && prob && R > 0 && S > 0 && S <= 256, "dyn_train_attn: bad arguments (S <= 256)");
  if (S <= TR_ATTN_LDS_MAX_S && (((uintptr_t)qkv | (uintptr_t)out) & 15) == 0) {  // (16-byte row accesses)
    DYN_LAUNCH(DYN_K_TRAIN_ATTN, "dyn_train_attn", k_train_attn_lds, dim3((unsigned)R * 4), dim3(4 * S), (size_t)(S * 72 + S * (S + 4)) * 4, (hipStream_t)stream,
               qkv, nvalid, S, out, prob);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_ATTN, "dyn_train_attn", k_train_attn, dim3((unsigned)R * 4), dim3(S), (size_t)S * 33 * 8, (hipStream_t)stream, qkv, nvalid, S,
             out, prob);
  return 0;
}
extern "C" int dyn_train_attn_bwd(const float* qkv, const float* nvalid, int R, int S, const float* prob, const float* dout, float* dscore,
                                  float* dqkv, void* stream) {
  DYN_REQUIRE(qkv && nvalid && prob && dout && dscore && dqkv && R > 0 && S > 0 && S <= 256, "dyn_train_attn_bwd: bad arguments (S <= 256)");
  if (S <= TR_ATTN_LDS_MAX_S && (((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0) {
    DYN_LAUNCH(DYN_K_TRAIN_ATTN, "dyn_train_attn_bwd", k_train_attn_bwd_lds, dim3((unsigned)R * 4), dim3(4 * S), (size_t)(S * 72 + 2 * S * (S + 4)) * 4,
               (hipStream_t)stream, qkv, nvalid, S, prob, dout, dqkv);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_ATTN, "dyn_train_attn_bwd", k_train_attn_bwd, dim3((unsigned)R * 4), dim3(S), (size_t)S * 33 * 8, (hipStream_t)stream, qkv,
             nvalid, S, prob, dout, dscore, dqkv);
  return 0;
}

// ---- residual + LayerNorm(eps 1e-6) over 128 columns (mlp_network.py:99-102): one wavefront per row ------------------------------------
// forward saves xhat and rstd; backward: dy_in = rstd (g - mean(g) - xhat mean(g xhat)), g = dout gamma; dgamma, dbeta by atomics.
__global__ void __launch_bounds__(256) k_train_layernorm(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, long P, float* __restrict__ out, float* __restrict__ xhat,
                                                         float* __restrict__ rstd) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= P) return;
  const float y0 = a[row * 128 + lane] + b[row * 128 + lane], y1 = a[row * 128 + 64 + lane] + b[row * 128 + 64 + lane];
  const float mean = wave_sum(y0 + y1) * (1.0f / 128.0f);
  const float d0 = y0 - mean, d1 = y1 - mean;
  const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / 128.0f);
  const float rs = 1.0f / sqrtf(var + 1e-6f);
  xhat[row * 128 + lane] = d0 * rs;
  xhat[row * 128 + 64 + lane] = d1 * rs;
  out[row * 128 + lane] = d0 * rs * gamma[lane] + beta[lane];
  out[row * 128 + 64 + lane] = d1 * rs * gamma[64 + lane] + beta[64 + lane];
  if (lane == 0) rstd[row] = rs;
}
__global__ void __launch_bounds__(256) k_train_layernorm_bwd(const float* __restrict__ dout, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, long P, int rows_per_wave, float* __restrict__ din,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float dg0 = 0.f, dg1 = 0.f, db0 = 0.f, db1 = 0.f;
  for (long row = w * rows_per_wave; row < (w + 1) * rows_per_wave && row < P; ++row) {
    const float o0 = dout[row * 128 + lane], o1 = dout[row * 128 + 64 + lane];
    const float x0 = xhat[row * 128 + lane], x1 = xhat[row * 128 + 64 + lane];
    const float g0 = o0 * gamma[lane], g1 = o1 * gamma[64 + lane];
    const float mg = wave_sum(g0 + g1) * (1.0f / 128.0f);
    const float mgx = wave_sum(g0 * x0 + g1 * x1) * (1.0f / 128.0f);
    const float rs = rstd[row];
    din[row * 128 + lane] = rs * (g0 - mg - x0 * mgx);
    din[row * 128 + 64 + lane] = rs * (g1 - mg - x1 * mgx);
    dg0 += o0 * x0; dg1 += o1 * x1; db0 += o0; db1 += o1;
  }
  // the block's four waves meet in LDS: one atomic per column and block (same-address atomics serialise in L2)
  float* red = reinterpret_cast<float*>(dyn_smem);  // [4 waves][256]
  float* mine = red + (threadIdx.x >> 6) * 256;
  mine[lane] = dg0; mine[64 + lane] = dg1; mine[128 + lane] = db0; mine[192 + lane] = db1;
  __syncthreads();
  const int t = threadIdx.x;
  const float sum = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
  atomicAdd((t < 128 ? dgamma : dbeta - 128) + t, sum);
}
extern "C" int dyn_train_layernorm(const float* a, const float* b, const float* gamma, const float* beta, long P, float* out, float* xhat, float* rstd,
                                   void* stream) {
  DYN_REQUIRE(a && b && gamma && beta && out && xhat && rstd && P > 0, "dyn_train_layernorm: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_layernorm", k_train_layernorm, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, gamma, beta,
             P, out, xhat, rstd);
  return 0;
}
extern "C" int dyn_train_layernorm_bwd(const float* dout, const float* xhat, const float* rstd, const float* gamma, long P, float* din, float* dgamma,
                                       float* dbeta, void* stream) {
  DYN_REQUIRE(dout && xhat && rstd && gamma && din && dgamma && dbeta && P > 0, "dyn_train_layernorm_bwd: bad arguments");
  const int rpw = 16;
  const long waves = (P + rpw - 1) / rpw;
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_layernorm_bwd", k_train_layernorm_bwd, dim3((unsigned)((waves + 3) / 4)), dim3(256), 1024 * sizeof(float), (hipStream_t)stream, dout,
             xhat, rstd, gamma, P, rpw, din, dgamma, dbeta);
  return 0;
}

// ---- colour blending over views + density fill (mlp_network.py:503-527): one thread per point ------------------------------------------
// logit masked to -1e9 where mask == 0, softmax over views, rgb = sum_v rgb_in blend; raw = [rgb, nvalid < 1 ? -1e9 : sigma].
__global__ void __launch_bounds__(256) k_train_blend(const float* __restrict__ logit, long logit_stride, const float* __restrict__ mask,
                                                     const float* __restrict__ rgb_feat, const float* __restrict__ sigma, long sigma_stride,
                                                     const float* __restrict__ nvalid, long P, int V, float* __restrict__ blend,
                                                     float* __restrict__ raw) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const long r0 = p * V;
  float mx = -INFINITY;
  for (int v = 0; v < V; ++v) {
    const float l = mask[r0 + v] == 0.f ? -1e9f : logit[(r0 + v) * logit_stride];
    mx = fmaxf(mx, l);
  }
  float den = 0.f;
  for (int v = 0; v < V; ++v) {
    const float l = mask[r0 + v] == 0.f ? -1e9f : logit[(r0 + v) * logit_stride];
    den += expf(l - mx);
  }
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int v = 0; v < V; ++v) {
    const float l = mask[r0 + v] == 0.f ? -1e9f : logit[(r0 + v) * logit_stride];
    const float bw = expf(l - mx) / den;
    blend[r0 + v] = bw;
    const float* f = rgb_feat + (r0 + v) * 35;
    c0 += f[0] * bw; c1 += f[1] * bw; c2 += f[2] * bw;
  }
  raw[p * 4] = c0; raw[p * 4 + 1] = c1; raw[p * 4 + 2] = c2;
  raw[p * 4 + 3] = nvalid[p] < 1.0f ? -1e9f : sigma[p * sigma_stride];
}
__global__ void __launch_bounds__(256) k_train_blend_bwd(const float* __restrict__ draw, const float* __restrict__ blend, const float* __restrict__ mask,
                                                         const float* __restrict__ rgb_feat, const float* __restrict__ nvalid, long P, int V,
                                                         float* __restrict__ dlogit, long dlogit_stride, float* __restrict__ dsigma,
                                                         long dsigma_stride) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const long r0 = p * V;
  const float d0 = draw[p * 4], d1 = draw[p * 4 + 1], d2 = draw[p * 4 + 2];
  float dot = 0.f;
  for (int v = 0; v < V; ++v) {
    const float* f = rgb_feat + (r0 + v) * 35;
    dot += (d0 * f[0] + d1 * f[1] + d2 * f[2]) * blend[r0 + v];
  }
  for (int v = 0; v < V; ++v) {
    const float* f = rgb_feat + (r0 + v) * 35;
    const float db = d0 * f[0] + d1 * f[1] + d2 * f[2];
    dlogit[(r0 + v) * dlogit_stride] = mask[r0 + v] == 0.f ? 0.f : blend[r0 + v] * (db - dot);
  }
  dsigma[p * dsigma_stride] = nvalid[p] < 1.0f ? 0.f : draw[p * 4 + 3];
}
extern "C" int dyn_train_blend(const float* logit, long logit_stride, const float* mask, const float* rgb_feat, const float* sigma, long sigma_stride,
                               const float* nvalid, long P, int V, float* blend, float* raw, void* stream) {
  DYN_REQUIRE(logit && mask && rgb_feat && sigma && nvalid && blend && raw && P > 0 && V > 0, "dyn_train_blend: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_blend", k_train_blend, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logit, logit_stride,
             mask, rgb_feat, sigma, sigma_stride, nvalid, P, V, blend, raw);
  return 0;
}
extern "C" int dyn_train_blend_bwd(const float* draw, const float* blend, const float* mask, const float* rgb_feat, const float* nvalid, long P, int V,
                                   float* dlogit, long dlogit_stride, float* dsigma, long dsigma_stride, void* stream) {
  DYN_REQUIRE(draw && blend && mask && rgb_feat && nvalid && dlogit && dsigma && P > 0 && V > 0, "dyn_train_blend_bwd: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_blend_bwd", k_train_blend_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, draw, blend,
             mask, rgb_feat, nvalid, P, V, dlogit, dlogit_stride, dsigma, dsigma_stride);
  return 0;
}

// ---- backward of raw2outputs_vanilla (render_ray.py:134-201): one thread per ray ---------------------------------------------------------
// weights_i = alpha_i T_i, T_i = prod_{j<i} (1 - alpha_j + 1e-10); rgb = sum w_i c_i; depth = sum w_i z_i.
// dw_i = drgb . c_i + ddepth z_i + dweights_i;  dalpha_i = dw_i T_i - (sum_{j>i} dw_j w_j) / (1 - alpha_i + 1e-10);
// alpha = 1 - exp(-softplus(sigma) dist), dist = 1 (last sample 1e10): dsigma = dalpha exp(-sp dist) dist sigmoid(sigma).
// One WAVEFRONT per ray, a lane per sample (chunks of 64 samples with carried transmittance / prefix): the one-thread-per-ray form ran 3072
// rays on 48 wavefronts and took 0.07-0.17 ms per launch for 16 MB.  Prefix products / sums by log-step shuffles; 16-byte stores.
__device__ __forceinline__ float tr_scan_sum(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive SUFFIX sum over the wavefront: lane l gets sum_{l' >= l} v(l')
__device__ __forceinline__ float tr_rscan_sum(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(v, o);
    if (lane + o < 64) v += t;
  }
  return v;
}
__device__ __forceinline__ float tr_scan_prod(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o);
    if (lane >= o) v *= t;
  }
  return v;
}
__global__ void __launch_bounds__(256) k_train_composite_bwd(const float* __restrict__ raw, const float* __restrict__ z_vals,
                                                             const float* __restrict__ alpha, const float* __restrict__ weights,
                                                             const float* __restrict__ drgb, const float* __restrict__ ddepth,
                                                             const float* __restrict__ dweights, int R, int S, float* __restrict__ draw) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;  // (whole wavefronts)
  const float g0 = drgb ? drgb[r * 3] : 0.f, g1 = drgb ? drgb[r * 3 + 1] : 0.f, g2 = drgb ? drgb[r * 3 + 2] : 0.f;
  const float gd = ddepth ? ddepth[r] : 0.f;
  // The suffix sums  sum_{j>i} dw_j w_j  are formed DIRECTLY, walking the 64-sample blocks from the far end of the ray (round 5).  Rounds 2-4 took them
  // as total - prefix_i: far down a ray the suffix is a small difference of two large sums, its error (~1e-7 of the TOTAL, the same accumulated
  // rounding for neighbouring samples) is coherent along the ray, and the weight gradients of everything behind sigma -- sums over the samples --
  // came out 3e-4 of their largest entry off at 200 samples per ray while the oracle's own fp32 autograd is at 3e-6 (tools/grad_rootcause.py).
  // pass 1 (forward): the transmittance entering each block, parked in lane b of `Tin` for block b
  const int nb = (S + 63) / 64;
  float Tin = 1.0f, T_run = 1.0f;
  for (int b = 0; b < nb; ++b) {
    const int i = b * 64 + lane;
    const bool ok = i < S;
    const float a = ok ? alpha[(long)r * S + i] : 0.f;
    const float one_m = ok ? 1.0f - a + 1e-10f : 1.0f;
    const float incl = tr_scan_prod(one_m, lane);
    if (lane == (b & 63)) Tin = T_run;
    T_run *= __shfl(incl, 63);
  }
  // pass 2 (backward over the blocks): transmittance as a prefix product inside the block, suffix sums by a reverse scan + the carry of the later blocks
  float suffix_in = 0.f;
  for (int b = nb - 1; b >= 0; --b) {
    const int i = b * 64 + lane;
    const bool ok = i < S;
    const long o = (long)r * S + (ok ? i : S - 1);
    const float4 c = *reinterpret_cast<const float4*>(raw + o * 4);
    const float w = ok ? weights[o] : 0.f, a = ok ? alpha[o] : 0.f;
    const float dw = g0 * c.x + g1 * c.y + g2 * c.z + gd * z_vals[o] + (dweights ? dweights[o] : 0.f);
    const float one_m = ok ? 1.0f - a + 1e-10f : 1.0f;
    const float incl = tr_scan_prod(one_m, lane);
    float T = __shfl_up(incl, 1);
    T = (lane == 0 ? 1.0f : T) * __shfl(Tin, b & 63);
    const float sfx_incl = tr_rscan_sum(ok ? dw * w : 0.f, lane);
    float sfx = __shfl_down(sfx_incl, 1);
    sfx = (lane == 63 ? 0.f : sfx) + suffix_in;  // sum over the samples behind this one
    if (ok) {
      const float dalpha = dw * T - sfx / one_m;
      const float sg = c.w;
      const float sp = sg > 20.0f ? sg : log1pf(expf(sg));
      const float dist = (i == S - 1) ? 1e10f : 1.0f;
      const float dsp = dalpha * expf(-sp * dist) * dist;
      *reinterpret_cast<float4*>(draw + o * 4) = make_float4(g0 * w, g1 * w, g2 * w, sg > 20.0f ? dsp : dsp * tr_sigmoid(sg));
    }
    suffix_in += __shfl(sfx_incl, 0);
  }
}
extern "C" int dyn_train_composite_bwd(const float* raw, const float* z_vals, const float* alpha, const float* weights, const float* drgb,
                                       const float* ddepth, const float* dweights, int R, int S, float* draw, void* stream) {
  DYN_REQUIRE(raw && z_vals && alpha && weights && draw && R > 0 && S > 0, "dyn_train_composite_bwd: bad arguments");
  DYN_REQUIRE(S <= 4096, "dyn_train_composite_bwd: at most 4096 samples per ray (64 blocks of 64: the per-block transmittance is parked in the lanes)");
  DYN_REQUIRE(((size_t)raw | (size_t)draw) % 16 == 0, "dyn_train_composite_bwd: raw / draw must be 16-byte aligned (float4 accesses)");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_composite_bwd", k_train_composite_bwd, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw,
             z_vals, alpha, weights, drgb, ddepth, dweights, R, S, draw);
  return 0;
}

// =====================================================================================================================
// Second slice: the dynamic branch's network (DynibarDynamic.forward, mlp_network.py:236-316) and the two-branch compositing
// (raw2outputs, render_ray.py:214-330).  The dynamic net reuses the GEMM and the pooling / visibility / attention / LayerNorm kernels
// above; what it adds: a broadcast add (time feature onto every row, positional table onto every ray), its Fourier features, its
// colour head (sigmoid, masked to 0 where no view sees the point) and the backward of the two-branch compositing.
// =====================================================================================================================

// y[row, c] = x[row, c] + tab[(row % period), c]   (period 1: one vector for every row; period S: the positional table of a ray)
__global__ void __launch_bounds__(256) k_train_add_table(const float* __restrict__ x, long ldx, const float* __restrict__ tab, long ld_tab, int period,
                                                         long rows, int C, float* __restrict__ y, long ldy) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx / C;
  const int c = (int)(idx - row * C);
  if (row >= rows) return;
  y[row * ldy + c] = x[row * ldx + c] + tab[(row % period) * ld_tab + c];
}
extern "C" int dyn_train_add_table(const float* x, long ldx, const float* tab, long ld_tab, int period, long rows, int C, float* y, long ldy,
                                   void* stream) {
  DYN_REQUIRE(x && tab && y && rows > 0 && C > 0 && period >= 1, "dyn_train_add_table: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_add_table", k_train_add_table, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
             ldx, tab, ld_tab, period, rows, C, y, ldy);
  return 0;
}

// Fourier features of the dynamic net (mlp_network.py:147-160, :290, :303): pts_pe [P,36] = [PE_5(pts) 33 | 0 0 0], dir_pe [R,28] =
// [PE_4(glb_ray_dir) 27 | 0]
__global__ void __launch_bounds__(256) k_train_dynamic_embed(const float* __restrict__ pts, const float* __restrict__ ray_d, long P, int R,
                                                             float* __restrict__ pts_pe, float* __restrict__ dir_pe) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P) {
    float* o = pts_pe + i * 36;
    tr_embed(pts[i * 3], o + 0, 3);
    tr_embed(pts[i * 3 + 1], o + 1, 3);
    tr_embed(pts[i * 3 + 2], o + 2, 3);
    o[33] = 0.f; o[34] = 0.f; o[35] = 0.f;
  }
  if (i < R) {
    float* o = dir_pe + i * 28;
    float dn[3];
    tr_unit3(ray_d[i * 3], ray_d[i * 3 + 1], ray_d[i * 3 + 2], dn[0], dn[1], dn[2]);  // input_ray_dir = F.normalize(ray_d) (render_ray.py:915)
    for (int k = 0; k < 3; ++k) {
      const float x = dn[k];
      o[k] = x;
      for (int f = 0; f < 4; ++f) {
        float s, c;
        sincosf((float)(1 << f) * x, &s, &c);
        o[3 + f * 3 + k] = c;
        o[15 + f * 3 + k] = s;
      }
    }
    o[27] = 0.f;
  }
}
extern "C" int dyn_train_dynamic_embed(const float* pts, const float* ray_d, long P, int R, float* pts_pe, float* dir_pe, void* stream) {
  DYN_REQUIRE(pts && ray_d && pts_pe && dir_pe && P > 0 && R > 0 && R <= P, "dyn_train_dynamic_embed: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_dynamic_embed", k_train_dynamic_embed, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts,
             ray_d, P, R, pts_pe, dir_pe);
  return 0;
}

// colour / density head (mlp_network.py:295-315): raw = [sigmoid(logit) masked to 0 where no view sees the point | sigma - shift, -1e9
// where no view sees the point]; backward into dlogit [P,3(ld)] and dsigma [P]
__global__ void __launch_bounds__(256) k_train_dynamic_head(const float* __restrict__ logit, long ld_logit, const float* __restrict__ sigma,
                                                            const float* __restrict__ nvalid, float shift, long P, float* __restrict__ raw) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const bool seen = nvalid[p] != 0.f;
  for (int c = 0; c < 3; ++c) raw[p * 4 + c] = seen ? tr_sigmoid(logit[p * ld_logit + c]) : 0.f;
  raw[p * 4 + 3] = nvalid[p] < 1.0f ? -1e9f : sigma[p] - shift;
}
__global__ void __launch_bounds__(256) k_train_dynamic_head_bwd(const float* __restrict__ draw, const float* __restrict__ raw,
                                                                const float* __restrict__ nvalid, long P, float* __restrict__ dlogit, long ld_dlogit,
                                                                float* __restrict__ dsigma) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const bool seen = nvalid[p] != 0.f;
  for (int c = 0; c < 3; ++c) {
    const float s = raw[p * 4 + c];
    dlogit[p * ld_dlogit + c] = seen ? draw[p * 4 + c] * s * (1.0f - s) : 0.f;
  }
  dsigma[p] = nvalid[p] < 1.0f ? 0.f : draw[p * 4 + 3];
}
extern "C" int dyn_train_dynamic_head(const float* logit, long ld_logit, const float* sigma, const float* nvalid, float shift, long P, float* raw,
                                      void* stream) {
  DYN_REQUIRE(logit && sigma && nvalid && raw && P > 0, "dyn_train_dynamic_head: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_dynamic_head", k_train_dynamic_head, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logit,
             ld_logit, sigma, nvalid, shift, P, raw);
  return 0;
}
extern "C" int dyn_train_dynamic_head_bwd(const float* draw, const float* raw, const float* nvalid, long P, float* dlogit, long ld_dlogit, float* dsigma,
                                          void* stream) {
  DYN_REQUIRE(draw && raw && nvalid && dlogit && dsigma && P > 0, "dyn_train_dynamic_head_bwd: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_dynamic_head_bwd", k_train_dynamic_head_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
             draw, raw, nvalid, P, dlogit, ld_dlogit, dsigma);
  return 0;
}

// backward of the two-branch compositing (render_ray.py:246-330): one thread per ray.
//   a_d, a_s = 1 - exp(-softplus(sigma) dist);  A = 1 - (1 - a_s)(1 - a_d);  T_i = prod_{j<i} (1 - A_j + 1e-10)
//   w_d = a_d T, w_s = a_s T, w = A T;  rgb_dy = sum w_d c_d, rgb_static = sum w_s c_s, rgb = rgb_dy + rgb_static, depth = sum w z
// upstream: g_rgb, g_rgb_static, g_rgb_dy [R,3], g_depth [R], g_wd, g_ws, g_w [R,S] (each may be NULL)
//   Wd_i = (g_rgb + g_rgb_dy) . c_d,i + g_wd,i;  Ws_i likewise;  Ww_i = g_depth z_i + g_w,i;  Q_i = Wd_i a_d,i + Ws_i a_s,i + Ww_i A_i
//   dA_j = Ww_j T_j - (sum_{i>j} Q_i T_i) / (1 - A_j + 1e-10);  da_d,j = Wd_j T_j + dA_j (1 - a_s,j);  da_s,j = Ws_j T_j + dA_j (1 - a_d,j)
__device__ __forceinline__ float tr_alpha(float sg, bool last, float& dalpha_dsigma) {
  const float sp = sg > 20.0f ? sg : log1pf(expf(sg));
  const float dist = last ? 1e10f : 1.0f;
  const float ex = expf(-sp * dist);
  dalpha_dsigma = ex * dist * (sg > 20.0f ? 1.0f : tr_sigmoid(sg));
  return 1.0f - ex;
}
__global__ void __launch_bounds__(256) k_train_composite2_bwd(const float* __restrict__ raw_dy, const float* __restrict__ raw_st,
                                                              const float* __restrict__ z_vals, const float* __restrict__ g_rgb,
                                                              const float* __restrict__ g_rgb_st, const float* __restrict__ g_rgb_dy,
                                                              const float* __restrict__ g_depth, const float* __restrict__ g_wd,
                                                              const float* __restrict__ g_ws, const float* __restrict__ g_w, int R, int S,
                                                              float* __restrict__ draw_dy, float* __restrict__ draw_st) {
  // one wavefront per ray, a lane per sample (see k_train_composite_bwd)
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float gd[3], gs[3];
  for (int c = 0; c < 3; ++c) {
    const float g = g_rgb ? g_rgb[r * 3 + c] : 0.f;
    gd[c] = g + (g_rgb_dy ? g_rgb_dy[r * 3 + c] : 0.f);
    gs[c] = g + (g_rgb_st ? g_rgb_st[r * 3 + c] : 0.f);
  }
  const float gz = g_depth ? g_depth[r] : 0.f;
  // (suffix sums formed directly from the far end of the ray, not as total - prefix: see k_train_composite_bwd)
  const int nb = (S + 63) / 64;
  float Tin = 1.0f, T_run = 1.0f;
  for (int b = 0; b < nb; ++b) {  // pass 0: the transmittance entering each block, parked in lane b
    const int i = b * 64 + lane;
    const bool ok = i < S;
    const long o = (long)r * S + (ok ? i : S - 1);
    float dd, ds;
    const float ad = tr_alpha(raw_dy[o * 4 + 3], i == S - 1, dd), as = tr_alpha(raw_st[o * 4 + 3], i == S - 1, ds);
    const float A = 1.0f - (1.0f - as) * (1.0f - ad);
    const float one_m = ok ? 1.0f - A + 1e-10f : 1.0f;
    const float incl = tr_scan_prod(one_m, lane);
    if (lane == (b & 63)) Tin = T_run;
    T_run *= __shfl(incl, 63);
  }
  float suffix_in = 0.f;
  for (int b = nb - 1; b >= 0; --b) {  // pass 1: the gradients, blocks from the far end
    const int i = b * 64 + lane;
    const bool ok = i < S;
    const long o = (long)r * S + (ok ? i : S - 1);
    const float4 cd = *reinterpret_cast<const float4*>(raw_dy + o * 4), cs = *reinterpret_cast<const float4*>(raw_st + o * 4);
    float dd, ds;
    const float ad = tr_alpha(cd.w, i == S - 1, dd), as = tr_alpha(cs.w, i == S - 1, ds);
    const float A = 1.0f - (1.0f - as) * (1.0f - ad);
    const float Wd = gd[0] * cd.x + gd[1] * cd.y + gd[2] * cd.z + (g_wd ? g_wd[o] : 0.f);
    const float Ws = gs[0] * cs.x + gs[1] * cs.y + gs[2] * cs.z + (g_ws ? g_ws[o] : 0.f);
    const float Ww = gz * z_vals[o] + (g_w ? g_w[o] : 0.f);
    const float one_m = ok ? 1.0f - A + 1e-10f : 1.0f;
    const float incl = tr_scan_prod(one_m, lane);
    float T = __shfl_up(incl, 1);
    T = (lane == 0 ? 1.0f : T) * __shfl(Tin, b & 63);
    const float term = ok ? (Wd * ad + Ws * as + Ww * A) * T : 0.f;
    const float sfx_incl = tr_rscan_sum(term, lane);
    float sfx = __shfl_down(sfx_incl, 1);
    sfx = (lane == 63 ? 0.f : sfx) + suffix_in;
    if (ok) {
      const float dA = Ww * T - sfx / one_m;
      const float dad = Wd * T + dA * (1.0f - as), das = Ws * T + dA * (1.0f - ad);
      const float wd = ad * T, ws = as * T;
      *reinterpret_cast<float4*>(draw_dy + o * 4) = make_float4(gd[0] * wd, gd[1] * wd, gd[2] * wd, dad * dd);
      *reinterpret_cast<float4*>(draw_st + o * 4) = make_float4(gs[0] * ws, gs[1] * ws, gs[2] * ws, das * ds);
    }
    suffix_in += __shfl(sfx_incl, 0);
  }
}
extern "C" int dyn_train_composite2_bwd(const float* raw_dy, const float* raw_st, const float* z_vals, const float* g_rgb, const float* g_rgb_st,
                                        const float* g_rgb_dy, const float* g_depth, const float* g_wd, const float* g_ws, const float* g_w, int R,
                                        int S, float* draw_dy, float* draw_st, void* stream) {
  DYN_REQUIRE(raw_dy && raw_st && z_vals && draw_dy && draw_st && R > 0 && S > 0, "dyn_train_composite2_bwd: bad arguments");
  DYN_REQUIRE(S <= 4096, "dyn_train_composite2_bwd: at most 4096 samples per ray (64 blocks of 64: the per-block transmittance is parked in the lanes)");
  DYN_REQUIRE(((size_t)raw_dy | (size_t)raw_st | (size_t)draw_dy | (size_t)draw_st) % 16 == 0,
              "dyn_train_composite2_bwd: raw / draw tensors must be 16-byte aligned (float4 accesses)");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_composite2_bwd", k_train_composite2_bwd, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw_dy,
             raw_st, z_vals, g_rgb, g_rgb_st, g_rgb_dy, g_depth, g_wd, g_ws, g_w, R, S, draw_dy, draw_st);
  return 0;
}


// =====================================================================================================================
// Third slice: the motion path.  Generic Fourier features with their backward w.r.t. the input (PeriodicEmbed, mlp_network.py:530-555:
// out = [x | cos(f_0 x) .. cos(f_{NF-1} x) | sin(f_0 x) .. ], x of D columns), and the zeroing of the last samples of every ray
// (render_ray.py:961, :1129).  MotionMLP itself is the GEMM with ReLU (act 2) over these.
// =====================================================================================================================
struct TrFreqs {
  int n;
  float f[16];
};
__global__ void __launch_bounds__(256) k_train_embed(const float* __restrict__ x, long ldx, long rows, int D, TrFreqs fr, float* __restrict__ out, long ld) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx / D;
  const int d = (int)(idx - row * D);
  if (row >= rows) return;
  const float v = x[row * ldx + d];
  float* o = out + row * ld;
  o[d] = v;
  for (int k = 0; k < fr.n; ++k) {
    float sn, cs;
    sincosf(fr.f[k] * v, &sn, &cs);
    o[D + k * D + d] = cs;
    o[D + fr.n * D + k * D + d] = sn;
  }
}
__global__ void __launch_bounds__(256) k_train_embed_bwd(const float* __restrict__ x, long ldx, long rows, int D, TrFreqs fr, const float* __restrict__ dout,
                                                         long ld, float* __restrict__ dx, long ld_dx, int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx / D;
  const int d = (int)(idx - row * D);
  if (row >= rows) return;
  const float v = x[row * ldx + d];
  const float* g = dout + row * ld;
  float acc = g[d];
  for (int k = 0; k < fr.n; ++k) {
    float sn, cs;
    sincosf(fr.f[k] * v, &sn, &cs);
    acc += fr.f[k] * (cs * g[D + fr.n * D + k * D + d] - sn * g[D + k * D + d]);
  }
  if (accumulate) dx[row * ld_dx + d] += acc; else dx[row * ld_dx + d] = acc;
}
static int tr_freqs(const float* freqs, int n, TrFreqs& fr) {
  if (freqs == nullptr || n < 0 || n > 16) return 1;
  fr.n = n;
  for (int i = 0; i < 16; ++i) fr.f[i] = i < n ? freqs[i] : 0.f;
  return 0;
}
extern "C" int dyn_train_embed(const float* x, long ldx, long rows, int D, const float* freqs, int n_freqs, float* out, long ld, void* stream) {
  TrFreqs fr;
  DYN_REQUIRE(x && out && rows > 0 && D > 0 && tr_freqs(freqs, n_freqs, fr) == 0 && ld >= (long)D * (1 + 2 * n_freqs), "dyn_train_embed: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_embed", k_train_embed, dim3((unsigned)((rows * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, D,
             fr, out, ld);
  return 0;
}
extern "C" int dyn_train_embed_bwd(const float* x, long ldx, long rows, int D, const float* freqs, int n_freqs, const float* dout, long ld, float* dx,
                                   long ld_dx, int accumulate, void* stream) {
  TrFreqs fr;
  DYN_REQUIRE(x && dout && dx && rows > 0 && D > 0 && tr_freqs(freqs, n_freqs, fr) == 0, "dyn_train_embed_bwd: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_embed_bwd", k_train_embed_bwd, dim3((unsigned)((rows * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
             rows, D, fr, dout, ld, dx, ld_dx, accumulate);
  return 0;
}

// x[r, s, :] = 0 for the last n_last samples of every ray (forward of raw_coeff[:, -n:, :] *= 0 and its backward), then the rest scaled
__global__ void __launch_bounds__(256) k_train_zero_tail(float* __restrict__ x, long R, int S, int C, int n_last, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * S * C) return;
  const int s = (int)((idx / C) % S);
  x[idx] = s >= S - n_last ? 0.f : x[idx] * scale;
}
extern "C" int dyn_train_zero_tail(float* x, long R, int S, int C, int n_last, float scale, void* stream) {
  DYN_REQUIRE(x && R > 0 && S > 0 && C > 0 && n_last >= 0 && n_last <= S, "dyn_train_zero_tail: bad arguments");
  DYN_LAUNCH(DYN_K_TRAIN_ROWS, "dyn_train_zero_tail", k_train_zero_tail, dim3((unsigned)((R * S * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, R, S,
             C, n_last, scale);
  return 0;
}
