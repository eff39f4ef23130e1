// Register-resident MLP chain engine for gfx950.
//
// Every Linear layer of the per-point networks (reference ibrnet/mlp_network.py) is evaluated TRANSPOSED:
//     out^T [features x rows] = W [features x K] . act^T [K x rows]
// so the weights are the MFMA A operand and the activations the B operand.  One wavefront owns a tile of 32 rows
// (point-views or points): lane l = (j = l & 31 : the row, h = l >> 5 : the half).  The MFMA result layout
// ("D layout": register r of lane (j,h) holds feature (r&3) + 8*(r>>2) + 4*h of the 32-feature output tile, for row j) is
// exactly what the next layer's B operand wants when its k-slots are enumerated in that same order, so activations never
// leave the register file between layers: no LDS round trip, no transposes.  The summation order over K is a
// pack-time permutation of the reference's (results agree to fp32 round-off, not bitwise; tolerance 1e-4 per north_star).
//
// The engine (second half of this file, "B6" in the code): fp32 operands split into 16-bit parts, products on the 16-bit matrix pipe with fp32
//    accumulation.  Default build (DYN_SPLIT_F16 = 1): IEEE half parts (22 mantissa bits), three partial products hi.hi + hi.mid + mid.hi on
//    v_mfma_f32_32x32x16_f16 -- fp32-class products.  The x6 build keeps bf16 parts (v_mfma_f32_32x32x16_bf16, DYN_SPLIT_TERMS = 6: all 24 bits).
//    Weights are split and packed on the host (dyn_nets.hip: pack_layer_b6) into a stream of 48 KiB chunks in consumption order and DMA'd global->LDS
//    (global_load_lds_dwordx4) into a ring shared by the 4 or 8 waves of a workgroup.  Biases are accumulator initial values (LDS tables) or one extra
//    k-slot fed with 1.  (The round-1 engine on the native fp32 MFMA, v_mfma_f32_32x32x2_f32, was removed in round 5; the long-ray attention of
//    k_net_points still multiplies with that instruction directly: mfma32 below.)
// Three layer loops on the shipped engine (round 4): mlp_layer_b6 (two-slot ring, pair-granular pipeline: the kernels that run two or three waves per
// SIMD -- k_static_views, k_dynamic_views, k_selftest), mlp_layer_b6_duo (three-slot ring with a mid-chunk barrier, two output tiles interleaved, A
// fragments four pairs ahead: the kernels that run ONE wave per SIMD -- k_motion_mlp, k_net_points) and mlp_layer_b6_lds (weights resident in LDS,
// no ring: k_static_blend, whose 104 KiB of weights fit).
#pragma once
#include <utility>

#include "dyn_device.h"

// Round 6: NO FOREIGN WAVES BESIDE A ONE-WAVE-PER-SIMD KERNEL.  With two chunk streams (render_image.CHUNK_STREAMS) a rendered frame was not reproducible: 50-100 of
// its 147 456 rays differed by up to 1e-3 from run to run, in rounds 5 and 6 alike (tools/ragged_frame_ab.py).  tools/concurrency_probe*.py traced it to ONE kernel,
// k_static_ref_feat (40 registers, no LDS): whenever its waves shared a SIMD with a wave of k_motion_mlp or k_net_points -- the kernels that run one wave per SIMD on
// 190-256 architectural + 208-256 accumulation registers and leave 56-64 registers of the lane free -- aligned groups of 16 lanes of it came back with slightly wrong
// sums (one term of the 66 off), on fixed inputs; every other small kernel of the path, co-resident in the same way, stayed bit-exact, and so did everything beside the
// two-wave kernels, which leave no room for a foreign wave.  Neither the missing M0 wait state of the LDS-DMA statements (fixed all the same) nor the function call in
// k_motion_mlp is the cause; what is, inside the part, was not found.  What removes it, measured (profiles/r06_stream_determinism.txt): the one-wave-per-SIMD kernels and
// k_static_ref_feat claim all 512 registers of a lane (a clobber of v255 / a255: the kernel descriptor then asks for the whole file), so no wave of another kernel is ever
// placed beside them: 0 differing bytes in 24 + 12 probe rounds and in twelve two-stream frames against the one-stream frame.  Cost: the small kernels of the other stream
// no longer fill those SIMDs (frame +0.6 %), k_static_ref_feat 20 -> 42 us.
#ifndef DYN_EXCLUSIVE_CU
#define DYN_EXCLUSIVE_CU 1
#endif
#if DYN_EXCLUSIVE_CU && defined(__AMDGCN__)
#define DYN_CLAIM_REGISTER_FILE() asm volatile("; the whole register file" ::: "v255", "a255")
#else
#define DYN_CLAIM_REGISTER_FILE()
#endif
#define DYN_NET_THREADS 256   // workgroup of the point-level kernels: 4 waves share one weight ring
#define DYN_VIEW_THREADS 512  // workgroup of the view-level kernels: 8 waves (2 per SIMD) share one weight ring

// feature index (within a 32-feature tile) held in register r by a lane of half h
__host__ __device__ constexpr int dyn_fi(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// accumulators start at the layer's bias: table [2][16 * NT] in D-layout order (pack_rowtab), normally in LDS
template <int NT>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[NT], const float* tab) {
  const float4* t4 = reinterpret_cast<const float4*>(tab + ((threadIdx.x & 63) >> 5) * 16 * NT);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = t4[t * 4 + q];
      acc[t][q * 4] = b.x; acc[t][q * 4 + 1] = b.y; acc[t][q * 4 + 2] = b.z; acc[t][q * 4 + 3] = b.w;
    }
}

// ELU as a median: for v > 0, v <= e^v - 1 and 0 < v; for v <= 0, v <= e^v - 1 <= 0 -- so elu(v) = med3(v, e^v - 1, 0) in one instruction
// instead of a compare + select (which also hides worse under the matrix pipe, tools/ubench/mfma_valu_kind.hip)
__device__ __forceinline__ float elu1(float v) { return __builtin_amdgcn_fmed3f(v, __expf(v) - 1.0f, 0.0f); }
__device__ __forceinline__ float sigmoid1(float v) { return 1.0f / (1.0f + __expf(-v)); }
// ELU in the exponent's own domain: a layer whose output only feeds ELUs is packed with weights and bias times log2(e), so its
// accumulators hold u = v log2(e); elu_s(u) = log2(e) ELU(v) = med3(u, 2^u log2(e) - log2(e), 0) is three instructions (v_exp_f32 takes u as it
// is: the multiply by log2(e) of __expf is gone; the subtraction rides in the fma), and the consumer -- always a Linear or a dot-product
// table -- is packed times ln(2).  float(log2 e) float(ln 2) = 1 + 4e-9: invisible next to the split products' 2^-20.
#define DYN_LOG2E 1.44269504088896340736
#define DYN_LN2 0.69314718055994530942
#define DYN_ELU_PRE DYN_LOG2E  /* pack-time factor of a layer (weights and bias) whose output goes through elu_s */
#define DYN_ELU_POST DYN_LN2   /* pack-time factor of the weights that consume elu_s outputs */
__device__ __forceinline__ float elu_s(float u) {
#if defined(__AMDGCN__)
  return __builtin_amdgcn_fmed3f(u, fmaf(__builtin_amdgcn_exp2f(u), (float)DYN_LOG2E, -(float)DYN_LOG2E), 0.0f);
#else
  return __builtin_amdgcn_fmed3f(u, fmaf(exp2f(u), (float)DYN_LOG2E, -(float)DYN_LOG2E), 0.0f);
#endif
}

template <int NT>
__device__ __forceinline__ void acc_elu(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = elu1(acc[t][r]);
}
template <int NT>
__device__ __forceinline__ void acc_elu_s(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = elu_s(acc[t][r]);
}

// ---- reductions over the VSEG consecutive lanes (views, padded to a power of two) of one point ------------------------------
// Butterflies on the DPP lane-select of the VALU (the add / min / max itself carries the cross-lane read: one instruction per step,
// no LDS traffic and no lgkmcnt wait): quad_perm for xor 1 and xor 2, row_half_mirror / row_mirror to join the two halves of 8 / 16
// lanes (after the quad steps every lane of a half holds the half's value, so joining with the mirrored lane is an all-reduce);
// only the 32-lane case needs one ds_bpermute for its last step.  Every lane of the segment ends with the bit-identical result.
// (V, seg_base are unused; kept so call sites read like the maths.)
#define DYN_DPP_XOR1 0xB1         /* quad_perm [1,0,3,2] */
#define DYN_DPP_XOR2 0x4E         /* quad_perm [2,3,0,1] */
#define DYN_DPP_HALF_MIRROR 0x141 /* lane i <-> 7 - i within 8 */
#define DYN_DPP_ROW_MIRROR 0x140  /* lane i <-> 15 - i within 16 */
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int VSEG, class Op>
__device__ __forceinline__ float seg_reduce(float v, Op op) {
  static_assert(VSEG == 4 || VSEG == 8 || VSEG == 16 || VSEG == 32, "segments are 4, 8, 16 or 32 lanes");
  v = op(v, dpp_get<DYN_DPP_XOR1>(v));
  v = op(v, dpp_get<DYN_DPP_XOR2>(v));
  if (VSEG >= 8) v = op(v, dpp_get<DYN_DPP_HALF_MIRROR>(v));
  if (VSEG >= 16) v = op(v, dpp_get<DYN_DPP_ROW_MIRROR>(v));
  if (VSEG >= 32) v = op(v, __shfl_xor(v, 16));
  return v;
}
// v + (lane i <-> 7 - i)(v): the compiler folds the quad_perm lane moves into v_add_f32_dpp but leaves row_half_mirror as a separate
// v_mov_b32_dpp; written out, with the two wait states a DPP read of a just-written VGPR needs
__device__ __forceinline__ float add_half_mirror(float v) {
#if defined(__AMDGCN__)
  float r;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v));
  return r;
#else
  return v + dpp_get<DYN_DPP_HALF_MIRROR>(v);
#endif
}
template <int VSEG>
__device__ __forceinline__ float seg_sum(float v, int, int) {
  if (VSEG == 8) {  // xor 1, xor 2 (folded by the compiler), then the hand-folded half mirror
    v = seg_reduce<4>(v, [](float a, float b) {
#pragma clang fp contract(off)
      return a + b;
    });
    return add_half_mirror(v);
  }
  // contraction off: "x * w + dpp(x * w)" fused into an fma cannot take the DPP operand, a plain add folds the lane move into v_add_f32_dpp
  return seg_reduce<VSEG>(v, [](float a, float b) {
#pragma clang fp contract(off)
    return a + b;
  });
}
// four independent sums at once: the quad steps interleave (no DPP wait states between them) and the four half-mirror adds share ONE
// s_nop instead of carrying one each (the statistics of the view chain are 400 such sums per row tile)
template <int VSEG>
__device__ __forceinline__ void seg_sum4(float& a, float& b, float& c, float& d) {
#if defined(__AMDGCN__)
  if (VSEG == 8) {
    auto quad = [](float v) {
      return seg_reduce<4>(v, [](float x, float y) {
#pragma clang fp contract(off)
        return x + y;
      });
    };
    a = quad(a); b = quad(b); c = quad(c); d = quad(d);
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return;
  }
#endif
  a = seg_sum<VSEG>(a, 0, 0); b = seg_sum<VSEG>(b, 0, 0); c = seg_sum<VSEG>(c, 0, 0); d = seg_sum<VSEG>(d, 0, 0);
}
template <int VSEG>
__device__ __forceinline__ float seg_min(float v, int, int) {
  return seg_reduce<VSEG>(v, [](float a, float b) { return fminf(a, b); });
}
template <int VSEG>
__device__ __forceinline__ float seg_max(float v, int, int) {
  return seg_reduce<VSEG>(v, [](float a, float b) { return fmaxf(a, b); });
}

// dot product of the lane's 16*NTI activation registers with a [2][16*NTI] table in LDS (row h), summed over both halves
template <int NTI>
__device__ __forceinline__ float row_dot(const f32x16 (&act)[NTI], const float* tab) {
  const int h = (threadIdx.x & 63) >> 5;
  const float4* t4 = reinterpret_cast<const float4*>(tab + h * 16 * NTI);
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NTI; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = t4[t * 4 + q];
      s = fmaf(act[t][q * 4 + 0], w.x, s);
      s = fmaf(act[t][q * 4 + 1], w.y, s);
      s = fmaf(act[t][q * 4 + 2], w.z, s);
      s = fmaf(act[t][q * 4 + 3], w.w, s);
    }
  return s + __shfl_xor(s, 32);
}

// ====================================================================================================================
// Split-bf16 engine ("B6" in the code): the same register-resident chains on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x
// the fp32 MFMA rate), fp32 in / fp32 accumulate.  An fp32 operand is split into bf16 parts by repeated round-to-nearest,
// x = hi + mid (+ lo); bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32.  DYN_SPLIT_TERMS selects how many
// partial products of x.w are kept:
//   3 (default): two parts per operand (16 mantissa bits), products hi.hi, hi.mid, mid.hi; the dropped terms are <= 2^-17 |x w|.
//      Measured through the whole coarse+fine render_rays_mv against the real reference's outputs: worst error 3 % of the stated
//      tolerances (sigma 2.5e-5, colours < 1e-6) -- and 64x finer operands than the TF32 tensor cores the reference's own A100
//      runs used (PyTorch 1.10 default).  3 MFMAs x 32 cycles cover K = 16 against 8 x 64 cycles of v_mfma_f32_32x32x2_f32.
//   6: three parts per operand (all 24 mantissa bits), products down to 2^-18 (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi):
//      fp32-class round-off (measured slightly better than an fp32 fma chain), 6 MFMAs per K = 16.
// Layouts: A (weights) lane (n = l & 31, h = l >> 5) holds W[n][k = 8h + i], B (activations) lane (j, h) holds act[k = 8h + i][j],
// i = 0..7; the D layout is unchanged, so MFMA group m of input tile T consumes the lane's registers r = 8m + i, i.e. feature
// 32T + fi(8m + i, h): the chain still never leaves the register file.  Weights are split on the host; a (k-group, output tile)
// pair is DYN_SPLIT_PARTS lane-linear 1 KiB images [hi | mid (| lo)], a chunk is 48 KiB of pairs (>= 2300 matrix-pipe cycles per
// wave, longer than the ~1.1 us an LDS-DMA chunk needs from issue to landing).
// ====================================================================================================================
#ifndef DYN_SPLIT_TERMS
#define DYN_SPLIT_TERMS 3
#endif
// DYN_SPLIT_F16 = 1: the two parts of the 3-term engine are IEEE half floats instead of bf16 (v_mfma_f32_32x32x16_f16: same shape, same
// rate).  Two halves carry 22 mantissa bits where two bf16 carry 16, so the kept products hi.hi + hi.mid + mid.hi are good to ~2^-20
// of |x w| (activation parts by truncation, weight parts by round-to-nearest on the host; dropped mid.mid <= 2^-22): fp32-class
// products at the cost of the 3-term engine.  Range: a half holds |x| < 65504 (beyond, the truncating convert saturates and the
// second part absorbs up to another 65504: graceful, not silent inf) and residuals below 6e-5 are subnormal halves, an ABSOLUTE
// error floor of 2^-24 per operand -- fp32's own epsilon at O(1), which is the scale of every activation and weight of these nets
// (weights outside the half range are refused at pack time).
#ifndef DYN_SPLIT_F16
#define DYN_SPLIT_F16 1
#endif
#if DYN_SPLIT_F16 && DYN_SPLIT_TERMS != 3
#error "the half-float split engine has two parts per operand (DYN_SPLIT_TERMS == 3)"
#endif
#if DYN_SPLIT_TERMS == 3
#define DYN_SPLIT_PARTS 2
#define B6_CHUNK_PAIRS 24
#elif DYN_SPLIT_TERMS == 6
#define DYN_SPLIT_PARTS 3
#define B6_CHUNK_PAIRS 16
#else
#error "DYN_SPLIT_TERMS must be 3 or 6"
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#define B6_PAIR_FLOATS (DYN_SPLIT_PARTS * 256)        // parts x 64 lanes x 4 dwords
#define B6_CHUNK (B6_CHUNK_PAIRS * B6_PAIR_FLOATS)  // floats per chunk (48 KiB)

// developer instrumentation (tools/phasebench.py): cycle stamps of wave 0 of two workgroups of each network kernel
// (PHASE_KID 0: view chain, 1: point chain, 2: blend) at layer boundaries, and the cycles waited at every ring acquire
#ifdef DYN_PHASE_TIMING
__device__ unsigned long long g_phase[3][2][160];  // [0..31] layer boundaries, [32..95] cycles waited in the ring acquire of chunk c, [96..159] its time
#define DYN_PHASE_ON (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))
#define DYN_PHASE(i)                                                                                  \
  do {                                                                                                \
    if (DYN_PHASE_ON) g_phase[PHASE_KID][blockIdx.x != 0][i] = __builtin_readcyclecounter();          \
  } while (0)
extern "C" int dyn_debug_phases(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase)) == hipSuccess ? 0 : 1;
}
#define DYN_PHASE_T0 const unsigned long long phase_t0 = __builtin_readcyclecounter();
#ifdef DYN_PHASE_SKEW
__device__ unsigned long long g_skew[8][4];
extern "C" int dyn_debug_skew(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_skew), sizeof(g_skew)) == hipSuccess ? 0 : 1; }
extern "C" int dyn_debug_skew_reset(void) { unsigned long long z[32] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_skew), z, sizeof(z)) == hipSuccess ? 0 : 1; }
#endif
#define DYN_PHASE_WAIT(R, c)                                                                          \
  do {                                                                                                \
    if (DYN_PHASE_ON && (c) < 64) {                                                                   \
      g_phase[(R).kid][blockIdx.x != 0][32 + (c)] = __builtin_readcyclecounter() - phase_t0;          \
      g_phase[(R).kid][blockIdx.x != 0][96 + (c)] = phase_t0;                                         \
    }                                                                                                 \
  } while (0)
#define DYN_PHASE_RING_KID(R, k) (R).kid = (k)
#else
#define DYN_PHASE(i)
#define DYN_PHASE_T0
#define DYN_PHASE_WAIT(R, c)
#define DYN_PHASE_RING_KID(R, k)
#endif

struct WeightRing6 {
  const float* gbase;  // the packed stream (uniform)
  float* buf;
  int next, total;
  int skip_at, skip_n;  // chunks [skip_at, skip_at + skip_n) of the stream are not ring traffic (their user reads them straight from global)
  int round;            // floats moved by the whole workgroup per DMA instruction (threads * 4)
  int half;             // wave-uniform (SGPR): 0 for waves 0-3 of the workgroup, 1 for waves 4-7 (the SIMD partners)
  // spread DMA (round 5): the pieces of chunk `fill` go out one at a time between the MFMAs of the chunk before it
  unsigned lds_wave;    // LDS byte address of this wave's slice of buffer 0 (+ 512 floats), wave-uniform (an SGPR)
  unsigned lane_off;    // byte offset, from the start of a chunk, of this lane's 16 bytes of the wave's slice (+ 512 floats)
  int fill, issued;     // chunk being requested (-1: none) and how many of this wave's pieces of it have gone out
  int nbuf;             // 2: double buffer (chunk c + 1 streams in under chunk c); 1: one 48 KiB buffer (several small workgroups per CU
                        // hide each other's exposed DMA instead)
#ifdef DYN_PHASE_TIMING
  int kid;
#endif
};


// Round 5: WHERE the two-slot ring's LDS-DMA goes out.  Rounds 1-4 requested the whole next chunk right behind the chunk barrier: all eight
// waves of the workgroup issue their six 1 KiB pieces at the same moment, the L2 -> LDS path takes one piece per ~37 cycles (48 KiB per ~1800
// cycles, tools/motionbench.py `dmaonly`), the request queue backs up and every wave -- both waves of every SIMD, they have just left the same
// barrier -- stands at its `global_load_lds` until its pieces are accepted.  Timing-only builds of k_static_views<8> (tools/experiments/r05_gpu_calls,
// call 2 and 3): no ring at all 1451 us, the barrier alone 1518, the DMA without the barrier 1868-1877, everything 1798-1873: the barrier costs
// nothing, the burst costs 19 %.  Since round 5 ring6_acquire only opens the window (fill = c + 1); the layer loop hands out one piece at a
// time between the MFMAs of the first two thirds of chunk c (ring6_feed), SGPR base + 32-bit lane offset as in the three-slot ring.
#ifndef B6_DMA_POLICY  /* cache policy of the weight pieces: "" (plain), " nt", " sc1", " sc0 sc1" */
#define B6_DMA_POLICY ""
#endif
#ifndef B6_DMA_WINDOW_NUM  /* the pieces go out over the first NUM / DEN of the chunk's pairs */
#define B6_DMA_WINDOW_NUM 2
#define B6_DMA_WINDOW_DEN 3
#endif
template <int OFF>
__device__ __forceinline__ void b6_dma_piece(const float* g_uniform, unsigned lane_off, unsigned l, float* l_emu) {
#if defined(__AMDGCN__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  // (`s_nop 0`: the wait state the ISA asks for between an SALU write of M0 and the LDS-DMA that reads it; hipcc pads nothing inside an asm string)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" B6_DMA_POLICY ::"v"(lane_off), "s"(g_uniform), "s"(l), "n"(OFF) : "m0", "memory");
#pragma clang diagnostic pop
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)g_uniform + lane_off),
                                   (__attribute__((address_space(3))) void*)l_emu, 16, OFF, 0);
#endif
}
__device__ __forceinline__ int ring6_pieces(const WeightRing6& R) { return B6_CHUNK / R.round; }  // 1 KiB pieces per wave and chunk (6 at 8 waves)
// piece k of this wave's slice of chunk R.fill
__device__ __forceinline__ void ring6_piece(const WeightRing6& R, int k) {
  const int waves = R.round / 256, per_wave = B6_CHUNK / waves;
  const int grp = k / 6, i = k % 6;
  const int chunk = R.fill;
  const float* gu = R.gbase + (long)(chunk + (chunk >= R.skip_at ? R.skip_n : 0)) * B6_CHUNK + grp * 1536;
  const int slot = R.nbuf == 2 ? (chunk & 1) : 0;
  const unsigned l = R.lds_wave + (unsigned)((slot * B6_CHUNK + grp * 1536) * sizeof(float));
  float* le = R.buf + slot * B6_CHUNK + (threadIdx.x >> 6) * per_wave + 512 + grp * 1536;  // (emulator build)
  if (i == 0) b6_dma_piece<-2048>(gu, R.lane_off, l, le);
  if (i == 1) b6_dma_piece<-1024>(gu, R.lane_off, l, le);
  if (i == 2) b6_dma_piece<0>(gu, R.lane_off, l, le);
  if (i == 3) b6_dma_piece<1024>(gu, R.lane_off, l, le);
  if (i == 4) b6_dma_piece<2048>(gu, R.lane_off, l, le);
  if (i == 5) b6_dma_piece<3072>(gu, R.lane_off, l, le);
}
// pair `pr` of the `npc` pairs of the chunk being consumed: pieces of the next chunk due by now (all of them by two thirds of the chunk, so that
// the last one has a third of a chunk to land before the barrier that publishes it)
__device__ __forceinline__ void ring6_feed(WeightRing6& R, int pr, int npc) {
  if (R.fill < 0) return;
  const int n = ring6_pieces(R);
  const int den = (B6_DMA_WINDOW_NUM * npc + B6_DMA_WINDOW_DEN - 1) / B6_DMA_WINDOW_DEN > 0 ? (B6_DMA_WINDOW_NUM * npc + B6_DMA_WINDOW_DEN - 1) / B6_DMA_WINDOW_DEN : 1;
  int want = (n * (pr + 1) + den - 1) / den;
  if (want > n) want = n;
  for (; R.issued < want; ++R.issued) ring6_piece(R, R.issued);
}

__device__ __forceinline__ void ring6_issue(const WeightRing6& R, int chunk) {
  // Every wave moves one contiguous slice of the chunk (6 KiB at 8 waves, 12 KiB at 4) as 1 KiB pieces.  The instruction's immediate
  // offset applies to the global and to the LDS address alike and the LDS image is the stream's own layout, so six pieces share one
  // address pair (vector address + M0) set in the middle of their 6 KiB: offsets -2048 ... +3072 fit the 13-bit field.
  const int waves = R.round / 256, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per_wave = B6_CHUNK / waves;  // floats
  const float* g = R.gbase + (long)(chunk + (chunk >= R.skip_at ? R.skip_n : 0)) * B6_CHUNK + wave * per_wave + 512 + lane * 4;
  float* l = R.buf + (R.nbuf == 2 ? (chunk & 1) : 0) * B6_CHUNK + wave * per_wave + 512;
#pragma unroll
  for (int grp = 0; grp < 2; ++grp)
    if (grp * 1536 < per_wave) {
      const auto* gg = (const __attribute__((address_space(1))) void*)(g + grp * 1536);
      auto* ll = (__attribute__((address_space(3))) void*)(l + grp * 1536);
      __builtin_amdgcn_global_load_lds(gg, ll, 16, -2048, 0);
      __builtin_amdgcn_global_load_lds(gg, ll, 16, -1024, 0);
      __builtin_amdgcn_global_load_lds(gg, ll, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gg, ll, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds(gg, ll, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds(gg, ll, 16, 3072, 0);
    }
}
// threads: the workgroup size.  Kernels pass their compile-time constant: the piece loop of ring6_issue then unrolls without branches
// and the implicit blockDim load (a memory round trip, waited for with vmcnt(0)) disappears -- worth 10 % of the view kernel.
__device__ __forceinline__ void ring6_init(WeightRing6& R, const float* stream, int total, float* lds, int skip_at = 1 << 30, int skip_n = 0,
                                           int threads = 0, int nbuf = 2) {
  R.gbase = stream;
  R.buf = lds;
  R.next = 0;
  R.total = total - skip_n;
  R.skip_at = skip_at;
  R.skip_n = skip_n;
  R.round = (threads > 0 ? threads : (int)blockDim.x) * 4;
  R.nbuf = nbuf;
#if defined(__AMDGCN__)
  R.half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
#else
  R.half = (int)threadIdx.x >> 8;
#endif
  {
    const int waves = R.round / 256, per_wave = B6_CHUNK / waves;
#if defined(__AMDGCN__)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    R.lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + wave * per_wave + 512);
#else
    const int wave = (int)threadIdx.x >> 6;
    R.lds_wave = 0;
#endif
    R.lane_off = (unsigned)((wave * per_wave + 512 + (threadIdx.x & 63) * 4) * sizeof(float));
  }
  R.fill = -1;
  R.issued = 0;
  DYN_PHASE_RING_KID(R, 0);
  ring6_issue(R, 0);
}
__device__ __forceinline__ const float* ring6_acquire(WeightRing6& R) {
  DYN_PHASE_T0
  if (R.nbuf == 1) {
    // single buffer: chunk c can only be fetched once every wave has left chunk c - 1 (chunk 0 was issued by ring6_init)
    const int c = R.next++;
    if (c > 0) {
      __syncthreads();
      ring6_issue(R, c);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    DYN_PHASE_WAIT(R, c);
    return R.buf;
  }
  if (R.fill >= 0) {  // whatever the consumer of the previous chunk did not hand out itself (a layer loop that feeds leaves nothing here)
    const int n = ring6_pieces(R);
    for (; R.issued < n; ++R.issued) ring6_piece(R, R.issued);
    R.fill = -1;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#ifdef DYN_PHASE_SKEW
  const unsigned long long skew_t1 = __builtin_readcyclecounter();
#endif
  __syncthreads();
  const int c = R.next++;
#ifdef DYN_PHASE_SKEW
  // developer diagnostic (tools/phasebench.py built with PHASE_FLAGS=-DDYN_PHASE_SKEW): per wave of one workgroup, cycles spent waiting for the
  // wave's own DMA pieces and cycles spent in the chunk barrier, summed over the chunks
  if ((threadIdx.x & 63) == 0 && blockIdx.x == gridDim.x / 2 && R.kid == 0) {
    g_skew[threadIdx.x >> 6][0] += skew_t1 - phase_t0;
    g_skew[threadIdx.x >> 6][1] += __builtin_readcyclecounter() - skew_t1;
    g_skew[threadIdx.x >> 6][2] += 1;
  }
#endif
  DYN_PHASE_WAIT(R, c);
  R.fill = c + 1 < R.total ? c + 1 : -1;  // the window of chunk c + 1 opens: its slot (chunk c - 1's) is free behind this barrier
  R.issued = 0;
  return R.buf + (c & 1) * B6_CHUNK;
}

// Issue priority in the two-wave layer loop (k_static_views, k_dynamic_views).  What ships (B6_PRIO_MODE 5) is, instruction for instruction in the ISA, a PRIORITY PULSE:
// behind every second MFMA triple the wave executes `s_setprio 1`, four VALU instructions of its next operand slice, `s_setprio 0`.  The wave that has just issued its
// products thereby starts its VALU slice ahead of its SIMD partner, which keeps the two waves of a SIMD out of step (in step they want the matrix pipe at the same
// moments and the VALU at the same moments: forcing them into step measured +6 %, round 3).  The source form is the one rounds 3-5 shipped -- a per-lane condition
// `(threadIdx.x >> 8) ^ phase`, which hipcc turns into an exec-masked region WITHOUT a branch in which BOTH `s_setprio` execute (scalar instructions ignore the exec
// mask), separated by the VALU instructions hipcc schedules between the two halves -- because it is the only form that produces exactly this pulse: round 3 thought it
// was alternating the priority of the wave pair, round 5 found out what it really executes, round 6 read the pulse off the disassembly and tried to write it down
// directly (`s_setprio 1` at the site, `s_setprio 0` in front of the pair's products: the pulse then spans the whole VALU slice): slower.  Measured, same box,
// alternating processes, k_static_views<8> us / bench step ms / frame ms: mode 5 1823-1831 / 2.771-2.783 / 656-665; explicit pulse 1863 / 2.823 / 664; position toggle
// (mode 1) 1835 / 2.799 / 725 (its waves end a layer at priority 1 and starve the other stream's kernels); no priority (mode 0) 1845 / 2.806 / 676
// (gpurun_out/r6c1_ab.txt, r6c3_ab.txt; round 5: profiles/r05_ab_variants.txt).  B6_PRIO_FLIP = pairs between two pulses.
#ifndef B6_PRIO_FLIP
#define B6_PRIO_FLIP 2
#endif
#ifndef B6_PRIO_MODE
#define B6_PRIO_MODE 5  /* 5: the pulse (shipped); 1: position toggle; 0: none (A/B builds) */
#endif
__device__ __forceinline__ void ring6_prio_flip(const WeightRing6& R, int phase) {
#if defined(__AMDGCN__)
  if (R.round != 8 * 256) return;
#if B6_PRIO_MODE == 1
  if (phase & 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#elif B6_PRIO_MODE == 5
  // a per-lane condition on purpose (see above): the exec-masked region in which both s_setprio execute, a few VALU instructions apart
  if ((((int)threadIdx.x >> 8) ^ phase) & 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
#endif
}
// exact three-way bf16 split of two fp32 values, each part packed as (first in the low half, second in the high half)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
#if DYN_SPLIT_F16
  // truncating pack-convert (one instruction per pair), exact residuals (<= 13 significant bits), truncating pack-convert again
  const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
  hi = __builtin_bit_cast(unsigned, h);
#if defined(__AMDGCN__)
  // three instructions per pair: the mixed-precision fma reads the half part, subtracts it from the fp32 value exactly and writes the
  // residual as a half (round to nearest) straight into its slot of the packed register
  // (two statements: hipcc puts an `s_nop 0` between them -- 310 in the view chain, free -- but as one statement the two parts no longer
  //  spread between the MFMAs and k_static_blend_ws measured 6 % slower, round 5)
  unsigned m;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(m) : "v"(hi), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(hi), "v"(b));
  mid = m;
#else
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  mid = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
#endif
  lo = 0u;
#else
  f32x2v v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
  f32x2v hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  f32x2v r1 = v - hf;
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2v));
#if DYN_SPLIT_PARTS == 3
  f32x2v mf = {__uint_as_float(mid << 16), __uint_as_float(mid & 0xffff0000u)};
  f32x2v r2 = r1 - mf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2v));
#else
  lo = 0u;
#endif
#endif
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4v a, u32x4v b, f32x16 c) {  // the split engine's MFMA (bf16 or half parts)
#if DYN_SPLIT_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

__host__ __device__ constexpr int b6_layer_chunks(int NT, int NSLOTS, int CP = B6_CHUNK_PAIRS) {
  return (((NSLOTS + 7) / 8) + (CP / NT) - 1) / (CP / NT);
}

// One Linear layer on the B6 engine: NT output tiles, NSLOTS input register slots (one fp32 activation per lane per slot; slot s of
// half h is the layer's input feature fixed at pack time).  feed(s) as in mlp_layer.
struct B6A {
  u32x4v hi, mid, lo;
};
__device__ __forceinline__ B6A b6_load_a(const float* pair, int lane) {
  const u32x4v* w = reinterpret_cast<const u32x4v*>(pair) + lane;
  B6A a;
  a.hi = w[0]; a.mid = w[64];
#if DYN_SPLIT_PARTS == 3
  a.lo = w[128];
#endif
  return a;
}

// split pairs [P0, P1) of k-group g of the lane's B operand (slots 8 g + 2 p, 8 g + 2 p + 1) into their bf16 parts
template <int NSLOTS, int P0, int P1, class Feed>
__device__ __forceinline__ void b6_split_pairs(Feed& feed, int g, u32x4v& bh, u32x4v& bm, u32x4v& bl) {
#pragma unroll
  for (int p2 = P0; p2 < P1; ++p2) {
    const float v0 = (g * 8 + 2 * p2 < NSLOTS) ? feed(g * 8 + 2 * p2) : 0.f;
    const float v1 = (g * 8 + 2 * p2 + 1 < NSLOTS) ? feed(g * 8 + 2 * p2 + 1) : 0.f;
    unsigned h_, m_, l_;
    split3_pair(v0, v1, h_, m_, l_);
    bh[p2] = h_; bm[p2] = m_; bl[p2] = l_;
  }
}

// Scheduling: the layer is a software pipeline over (k-group, output tile) pairs.  While the MFMAs of pair p issue, the A parts of
// pairs p + 1 .. p + B6_AHEAD are in flight from LDS and a slice of the B operand of the next k-group is produced (feed -> ELU etc.
// -> bf16 split).  A scheduling barrier after every pair keeps that interleave: without it the compiler hoists a whole chunk's VALU work (feeds,
// splits) above the chunk's MFMAs, and since the chunk barrier has just aligned the waves, both waves of a SIMD then sit in the
// VALU phase together and in the MFMA phase together -- matrix pipe and VALU never overlap (measured: layer time = sum of both).
// feed(s) is called exactly once per slot, in slot order.
#ifndef B6_AHEAD
#define B6_AHEAD 1  // pairs of A operands in flight from LDS ahead of the pair whose MFMAs issue
#endif
template <int NT, int NSLOTS, int AHEAD = B6_AHEAD, class Feed>
__device__ __forceinline__ void mlp_layer_b6(WeightRing6& R, f32x16 (&acc)[NT], Feed&& feed) {
  constexpr int NG = (NSLOTS + 7) / 8;
  constexpr int GPC = B6_CHUNK_PAIRS / NT;
  constexpr int NCH = (NG + GPC - 1) / GPC;
  static_assert(B6_CHUNK_PAIRS % NT == 0, "tiles per layer must divide the pairs of a chunk");
  static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "output tiles per layer");
  const int lane = threadIdx.x & 63;
  u32x4v bh, bm, bl, nh, nm, nl;
  b6_split_pairs<NSLOTS, 0, 4>(feed, 0, bh, bm, bl);
  nh = bh; nm = bm; nl = bl;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float* buf = ring6_acquire(R);
    constexpr int NPC_MAX = B6_CHUNK_PAIRS;
    const int npc = (NG - c * GPC < GPC ? NG - c * GPC : GPC) * NT;  // pairs of this chunk (compile-time after unrolling)
    B6A q[AHEAD + 1];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
      if (i < npc) q[i] = b6_load_a(buf + i * B6_PAIR_FLOATS, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pr = 0; pr < NPC_MAX; ++pr) {
      if (pr < npc) {
        const int gi = pr / NT, t = pr % NT;
        const int g = c * GPC + gi;
        ring6_feed(R, pr, npc);
#if B6_PRIO_FLIP && (B6_PRIO_MODE == 1 || B6_PRIO_MODE == 5)
        // issue priority every B6_PRIO_FLIP pairs (see ring6_prio_flip)
        if (pr % B6_PRIO_FLIP == 0) ring6_prio_flip(R, (pr / B6_PRIO_FLIP) & 1);
#endif
        if (pr + AHEAD < npc) q[(pr + AHEAD) % (AHEAD + 1)] = b6_load_a(buf + (pr + AHEAD) * B6_PAIR_FLOATS, lane);
        if (g + 1 < NG) {
          // this tile's share of the next k-group's operand: pairs [4 t / NT, 4 (t + 1) / NT)
          if (NT == 1) b6_split_pairs<NSLOTS, 0, 4>(feed, g + 1, nh, nm, nl);
          if (NT == 2 && t == 0) b6_split_pairs<NSLOTS, 0, 2>(feed, g + 1, nh, nm, nl);
          if (NT == 2 && t == 1) b6_split_pairs<NSLOTS, 2, 4>(feed, g + 1, nh, nm, nl);
          if (NT == 4 && t == 0) b6_split_pairs<NSLOTS, 0, 1>(feed, g + 1, nh, nm, nl);
          if (NT == 4 && t == 1) b6_split_pairs<NSLOTS, 1, 2>(feed, g + 1, nh, nm, nl);
          if (NT == 4 && t == 2) b6_split_pairs<NSLOTS, 2, 3>(feed, g + 1, nh, nm, nl);
          if (NT == 4 && t == 3) b6_split_pairs<NSLOTS, 3, 4>(feed, g + 1, nh, nm, nl);
          if (NT == 8 && t == 0) b6_split_pairs<NSLOTS, 0, 1>(feed, g + 1, nh, nm, nl);
          if (NT == 8 && t == 2) b6_split_pairs<NSLOTS, 1, 2>(feed, g + 1, nh, nm, nl);
          if (NT == 8 && t == 4) b6_split_pairs<NSLOTS, 2, 3>(feed, g + 1, nh, nm, nl);
          if (NT == 8 && t == 6) b6_split_pairs<NSLOTS, 3, 4>(feed, g + 1, nh, nm, nl);
        }
        const B6A& cur = q[pr % (AHEAD + 1)];
        // smallest partial products first
#if DYN_SPLIT_TERMS == 6
        acc[t] = mfma_bf16(cur.lo, bh, acc[t]);
        acc[t] = mfma_bf16(cur.hi, bl, acc[t]);
        acc[t] = mfma_bf16(cur.mid, bm, acc[t]);
#endif
        acc[t] = mfma_bf16(cur.mid, bh, acc[t]);
        acc[t] = mfma_bf16(cur.hi, bm, acc[t]);
        acc[t] = mfma_bf16(cur.hi, bh, acc[t]);
        if (t == NT - 1) { bh = nh; bm = nm; bl = nl; }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// =====================================================================================================================
// Round 4: the layer loop of the kernels that run ONE wave per SIMD (k_motion_mlp, k_net_points; > 256 registers per lane).
// With a second wave on the SIMD every bubble of mlp_layer_b6 is filled by the partner; alone, the wave shows what the loop costs
// (k_motion_mlp, a plain 256-wide ReLU chain with ~2.5 VALU per MFMA triple, had the matrix pipe 0.51 busy):
//  * the three partial products of a pair went to ONE accumulator back to back, and hipcc put the next pair's two ds_read_b128 between the
//    first and the second of them: an instruction between two MFMAs on the same accumulator costs ~43 cycles (the result is forwarded
//    only to an MFMA that issues right behind), and a dependent MFMA cannot issue before its predecessor has finished anyway;
//  * the A fragments were requested ONE pair ahead (~64 cycles before their `s_waitcnt lgkmcnt(0)`), less than the LDS latency under
//    four waves' traffic;
//  * the twelve 1 KiB LDS-DMA pieces of the next chunk were issued as one burst right behind the chunk barrier (~60 issue cycles each,
//    nothing to hide them under), and the barrier itself sat in front of the chunk's first LDS reads.
// mlp_layer_b6_duo issues the products of TWO output tiles interleaved (M1 t0, M1 t1, M2 t0, M2 t1, M3 t0, M3 t1: consecutive MFMAs
// never share an accumulator), keeps B6D_AHEAD pairs of A fragments in flight, also across chunk boundaries, gives every gap between
// two MFMAs a few instructions (two LDS reads, one third of a pair's split, one DMA piece), pinned with scheduling barriers, and
// runs on a THREE-slot ring whose only barrier sits in the MIDDLE of a chunk:
//     middle of chunk c:  s_waitcnt vmcnt(0)  (the wave's pieces of chunk c + 1, requested during the second half of chunk c - 1)
//                         s_barrier           (=> chunk c + 1 has landed for everybody; everybody has left chunk c - 1)
//     from there on    :  the pieces of chunk c + 2, a few per unit, into the slot chunk c - 1 occupied -- until shortly before the next barrier
// so no wave waits for memory or for LDS behind a barrier; the barrier only costs the skew of four waves doing identical work.
// =====================================================================================================================
#ifndef B6D_AHEAD
#define B6D_AHEAD 4  // pairs of A fragments in flight from LDS (even)
#endif
#define B6D_SLOTS 3
struct WeightRing3 {
  const float* gbase;  // the packed stream (uniform)
  const float* glane;  // this lane's source of the wave's slice of chunk 0 (+ 512 floats: the middle of a six-piece group, see ring6_issue)
  float* buf;          // LDS: B6D_SLOTS chunks
  unsigned lds_wave;   // LDS byte address of the wave's slice of slot 0 (+ 512 floats), wave-uniform (an SGPR)
  unsigned lane_off;   // byte offset of glane from gbase
  int next, total;     // chunk being consumed / chunks in the stream
  int waves;           // waves of the workgroup (compile-time at every call site)
  int cf;              // floats per chunk: (pairs per chunk) x B6_PAIR_FLOATS -- 48 KiB (24 pairs) by default, 32 KiB for the point kernels' stream
  int fill, issued;    // chunk being requested (-1: none) and how many of this wave's pieces of it have been issued
  // Persistent workgroups walk the stream once per row tile ("pass"): chunk k of a pass sits in slot[k % 3], the slots rotate by `total` between passes,
  // and the tail of a pass requests the first two chunks of the next one (wrap = 1, more = another pass follows: the only run-time value here).
  float* slot[B6D_SLOTS];
  unsigned lds_slot[B6D_SLOTS];  // lds_wave of each slot (SGPRs)
  int wrap;
  bool more;
#ifdef DYN_PHASE_TIMING
  int kid;
#endif
};
// One 1 KiB piece global -> LDS (16 bytes per lane; the LDS side is M0 + offset + 16 lane).  Inline asm: hipcc counts the builtin as an
// access to both memories ("flat"), and while one is pending every LDS wait becomes lgkmcnt(0) -- which would drain the A-fragment
// look-ahead at every unit.  The ring waits for its pieces itself (ring3_barrier).
template <int OFF>
__device__ __forceinline__ void ring3_dma(const float* g, float* l_emu, unsigned l, const float* g_uniform, unsigned lane_off) {
#if defined(__AMDGCN__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  // the piece's global address as SGPR base + 32-bit lane offset (half the address registers per instruction: -1.5 % of k_motion_mlp)
  // (`s_nop 0`: the wait state between the SALU write of M0 and the LDS-DMA that reads it)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(lane_off), "s"(g_uniform), "s"(l), "n"(OFF) : "m0", "memory");
#pragma clang diagnostic pop
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l_emu, 16, OFF, 0);
#endif
}
__device__ __forceinline__ int ring3_pieces(const WeightRing3& R) { return R.cf / (R.waves * 256); }  // 1 KiB pieces per wave and chunk
// piece k of this wave's slice of `chunk` (the slice is contiguous; six pieces share one address pair through the immediate offset, as in ring6_issue)
__device__ __forceinline__ void ring3_piece(const WeightRing3& R, int chunk, int k) {
  const int per_wave = R.cf / R.waves;
  const int grp = k / 6, i = k % 6;
  const int sc = chunk >= R.total ? chunk - R.total : chunk;  // chunks `total`, `total + 1` of a pass are chunks 0, 1 of the next one
  const float* g = R.glane + (long)sc * R.cf + grp * 1536;
  const float* gu = R.gbase + (long)sc * R.cf + grp * 1536;  // (uniform base; the lane's offset within the wave's slice is R.lane_off)
  const unsigned l = R.lds_slot[chunk % B6D_SLOTS] + (unsigned)(grp * 1536 * sizeof(float));
  float* le = R.slot[chunk % B6D_SLOTS] + (threadIdx.x >> 6) * per_wave + 512 + grp * 1536;  // (emulator build)
  if (i == 0) ring3_dma<-2048>(g, le, l, gu, R.lane_off);
  if (i == 1) ring3_dma<-1024>(g, le, l, gu, R.lane_off);
  if (i == 2) ring3_dma<0>(g, le, l, gu, R.lane_off);
  if (i == 3) ring3_dma<1024>(g, le, l, gu, R.lane_off);
  if (i == 4) ring3_dma<2048>(g, le, l, gu, R.lane_off);
  if (i == 5) ring3_dma<3072>(g, le, l, gu, R.lane_off);
}
// ALLOW: DMA pieces of this wave that may still be in flight behind the barrier (the youngest ones: loads complete in order)
template <int ALLOW = 0>
__device__ __forceinline__ void ring3_barrier() {
#if defined(__AMDGCN__)
  static_assert(ALLOW >= 0 && ALLOW < 16, "vmcnt immediate");
  __builtin_amdgcn_s_waitcnt(0x0F70 | ALLOW);  // vmcnt(ALLOW): this wave's older DMA pieces have landed (LDS reads in flight stay in flight)
  __builtin_amdgcn_s_barrier();
#else
  __syncthreads();
#endif
}
__device__ __forceinline__ void ring3_init(WeightRing3& R, const float* stream, int total, float* lds, int threads, int chunk_pairs = B6_CHUNK_PAIRS) {
  R.gbase = stream;
  R.buf = lds;
  R.next = 0;
  R.total = total;
  R.waves = threads / 64;
  R.cf = chunk_pairs * B6_PAIR_FLOATS;
  {
    const int per_wave = R.cf / R.waves;
#if defined(__AMDGCN__)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    R.lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + wave * per_wave + 512);
#else
    const int wave = threadIdx.x >> 6;
    R.lds_wave = 0;
#endif
    R.glane = stream + wave * per_wave + 512 + (threadIdx.x & 63) * 4;
    R.lane_off = (unsigned)((wave * per_wave + 512 + (threadIdx.x & 63) * 4) * sizeof(float));
  }
  R.fill = -1;
  R.issued = 0;
  R.wrap = 0;
  R.more = false;
  for (int j = 0; j < B6D_SLOTS; ++j) {
    R.slot[j] = lds + j * R.cf;
    R.lds_slot[j] = R.lds_wave + (unsigned)(j * R.cf * sizeof(float));
  }
  DYN_PHASE_RING_KID(R, 0);
  for (int c = 0; c < 2 && c < total; ++c)
    for (int k = 0; k < ring3_pieces(R); ++k) ring3_piece(R, c, k);
}
// first read of chunk R.next: chunk 0 is waited for here, every later chunk was published by the barrier in the middle of its predecessor
__device__ __forceinline__ void ring3_enter(WeightRing3& R) {
  if (R.next != 0) return;
  // chunk 1 (requested right behind chunk 0, at ring3_init or in the tail of the previous pass) may still be on its way: the barrier in the middle of
  // chunk 0 waits for it
  const int n = R.total > 1 ? ring3_pieces(R) : 0;
  if (n == 12) ring3_barrier<12>();
  else if (n == 8) ring3_barrier<8>();
  else if (n == 6) ring3_barrier<6>();
  else if (n == 4) ring3_barrier<4>();
  else ring3_barrier<0>();
}
__device__ __forceinline__ const float* ring3_slot(const WeightRing3& R, int chunk) { return R.slot[chunk % B6D_SLOTS]; }
// Request pieces of the chunk being filled.  Its window runs from the barrier in the middle of chunk c to B6D_DMA_MARGIN pairs before the barrier
// in the middle of chunk c + 1 (which waits for them): `half` 0 = the second half of chunk c carries the first half of the pieces, `half` 1 =
// the first half of chunk c + 1 the rest; num / den = how far through that half the wave is.  A CU's four waves then issue about one piece per
// 45 cycles -- the rate the L2 -> LDS path takes them at (tools/motionbench.py, `dmaonly`: 48 KiB per 1790 cycles) -- instead of twice that in
// one half of the time, which backed the queue up and stalled the issuing wave.
#ifndef B6D_DMA_MARGIN
#define B6D_DMA_MARGIN 4
#endif
__device__ __forceinline__ void ring3_feed(WeightRing3& R, int half, int num, int den) {
  if (R.fill < 0) return;
  if (R.fill >= R.total && !R.more) return;  // (the next pass's first chunks: only if there is a next pass)
  const int n = ring3_pieces(R), h0 = n / 2;
  int want = half == 0 ? (h0 * num + den - 1) / den : h0 + ((n - h0) * num + den - 1) / den;
  if (want > n) want = n;
  for (; R.issued < want; ++R.issued) ring3_piece(R, R.fill, R.issued);
}
// the middle of chunk R.next (once per chunk)
__device__ __forceinline__ void ring3_mid(WeightRing3& R) {
  DYN_PHASE_T0
  ring3_feed(R, 1, 1, 1);  // (whatever is left of chunk R.next + 1: normally nothing)
  if (R.next + 1 < R.total || R.wrap) ring3_barrier();
  DYN_PHASE_WAIT(R, R.next);
  R.fill = (R.next + 2 < R.total || R.wrap) ? R.next + 2 : -1;
  R.issued = 0;
}
__device__ __forceinline__ void ring3_leave(WeightRing3& R) {
  if (R.wrap && R.next == R.total - 1) ring3_feed(R, 1, 1, 1);  // the window of the next pass's chunk 1 ends with this pass
  ++R.next;
}
// between two passes of a persistent workgroup: the next pass's chunks 0 and 1 have been requested into the slots this pass's chunks `total` and
// `total + 1` would have taken; the pass starts like a fresh stream (ring3_enter waits for chunk 0 and publishes it)
__device__ __forceinline__ void ring3_next_pass(WeightRing3& R) {
  float* s0[B6D_SLOTS];
  unsigned l0[B6D_SLOTS];
  for (int j = 0; j < B6D_SLOTS; ++j) { s0[j] = R.slot[j]; l0[j] = R.lds_slot[j]; }
  for (int j = 0; j < B6D_SLOTS; ++j) { R.slot[j] = s0[(R.total + j) % B6D_SLOTS]; R.lds_slot[j] = l0[(R.total + j) % B6D_SLOTS]; }
  R.next = 0;
  R.fill = -1;
  R.issued = 0;
}

__device__ __forceinline__ float relu1(float v) {
#if defined(__AMDGCN__)
  float r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(v));  // fmaxf() canonicalises its operand first (a second v_max_f32)
  return r;
#else
  return v > 0.f ? v : 0.f;
#endif
}

// One Linear layer, interleaved form.  NT output tiles (NT = 1: plain triples, everything else in front of them), NSLOTS input slots,
// feed(s) as in mlp_layer_b6 (called exactly once per slot, in slot order).  Units of U = 2 pairs (tiles t, t + 1 of one k-group):
//   gap 0: A fragments of pair P + AHEAD          M1(t0)
//   gap 1: A fragments of pair P + AHEAD + 1      M1(t1)
//   gap 2 .. 5: the unit's share of the next k-group's operand, in thirds of a pair (feed, feed, split); gap 5 also one DMA piece
//               M2(t0) | M2(t1) | M3(t0) | M3(t1)
// (The unit loop is a compile-time expansion, dyn_static_for: left to `#pragma unroll`, a 100-unit layer exceeds the unroller's size limit,
// stays a loop, and the accumulators it then indexes at run time move to scratch memory.)
template <class F, int... Is>
__device__ __forceinline__ void dyn_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void dyn_static_for(F&& f) {
  dyn_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
#define DYN_INLINE_LAMBDA __attribute__((always_inline))

// An MFMA pinned to its place in the instruction stream.  The intrinsic has no side effects, so nothing ties it to the scheduling barriers
// around it (hipcc moved the products of half a chunk behind the chunk's barrier and issued them as one block); two empty volatile asm
// statements do: the first redefines the B operand (the MFMA cannot rise above it), the second redefines the result (it cannot sink below),
// and volatile statements keep their order among themselves and against loads, stores, barriers and the scheduling barriers.
__device__ __forceinline__ f32x16 mfma_pinned(const u32x4v& a, u32x4v& b, f32x16 c) {
#if defined(__AMDGCN__)
  asm volatile("" : "+v"(b));  // (the B operand: redefined in place, one chain per k-group -- touching the A fragment instead makes hipcc copy it for its second use)
  f32x16 d = mfma_bf16(a, b, c);
  asm volatile("" : "+a"(d));
  return d;
#else
  return mfma_bf16(a, b, c);
#endif
}

template <int NT, int NSLOTS, int CP = B6_CHUNK_PAIRS, int AHEAD = B6D_AHEAD, class Feed>
__device__ __forceinline__ void mlp_layer_b6_duo(WeightRing3& R, f32x16 (&acc)[NT], Feed&& feed) {
  constexpr int NG = (NSLOTS + 7) / 8;
  constexpr int NP = NG * NT;             // pairs of the layer
  constexpr int U = NT >= 2 ? 2 : 1;      // pairs per unit
  constexpr int QN = AHEAD + U;
  constexpr int NM = 3 * 4 * U / NT;      // operand micro-steps per unit: 4 pairs of slots per k-group, three steps each (feed, feed, split)
  static_assert(CP % NT == 0 && (NT == 1 || NT == 2 || NT == 4 || NT == 8), "output tiles per layer; pairs per chunk (CP) a multiple of them");
  static_assert(AHEAD % 2 == 0 && AHEAD >= 2 && AHEAD <= CP / 2, "the look-ahead must stay inside the published half chunk");
  const int lane = threadIdx.x & 63;
  const int c0 = R.next;
  u32x4v bh, bm, bl, nh, nm, nl;
  b6_split_pairs<NSLOTS, 0, 4>(feed, 0, bh, bm, bl);
  nh = bh; nm = bm; nl = bl;
  float v0 = 0.f, v1 = 0.f;  // the pair of slots being prepared
  B6A q[QN];
  auto load = [&](int P) DYN_INLINE_LAMBDA {
    return b6_load_a(ring3_slot(R, c0 + P / CP) + (P % CP) * B6_PAIR_FLOATS, lane);
  };
  ring3_enter(R);
#if !defined(__AMDGCN__)
  // Emulator build (tests/emu): the SAME schedule -- ring calls, feed order, order of the products per accumulator -- as a run-time loop; the
  // compile-time expansion below (up to 100 units per layer, each with its own constants) takes a host compiler tens of minutes.
  for (int i = 0; i < AHEAD && i < NP; ++i) q[i % QN] = load(i);
  for (int P = 0; P < NP; P += U) {
    const int pr = P % CP, npc = NP - (P - pr) < CP ? NP - (P - pr) : CP, mid = (npc / 2) / U * U;
    const int g = P / NT, t0 = P % NT, t1 = t0 + U - 1, um = (P % NT) / U;
    if (pr == mid) ring3_mid(R);
    auto gap = [&](int K) {
      for (int m = 0; m < NM; ++m) {
        const int at = NT == 1 ? 0 : (NM <= 4 ? 2 + m : m * 6 / NM);
        if (at != K || g + 1 >= NG) continue;
        const int step = um * NM + m, p2 = step / 3, st = step % 3, s0 = (g + 1) * 8 + 2 * p2;
        if (st == 0) v0 = s0 < NSLOTS ? feed(s0) : 0.f;
        if (st == 1) v1 = s0 + 1 < NSLOTS ? feed(s0 + 1) : 0.f;
        if (st == 2) {
          unsigned h_, m_, l_;
          split3_pair(v0, v1, h_, m_, l_);
          nh[p2] = h_; nm[p2] = m_; nl[p2] = l_;
        }
      }
    };
    auto dma = [&](int done) {
      const int wend = mid - B6D_DMA_MARGIN;
      if (pr >= mid) ring3_feed(R, 0, pr + done - mid, npc - mid);
      else if (wend <= 0) ring3_feed(R, 1, 1, 1);
      else ring3_feed(R, 1, pr + done < wend ? pr + done : wend, wend);
    };
    if (P + AHEAD < NP) q[(P + AHEAD) % QN] = load(P + AHEAD);
    gap(0);
    const B6A a0 = q[P % QN], a1 = q[(P + U - 1) % QN];
#if DYN_SPLIT_TERMS == 6
    acc[t0] = mfma_bf16(a0.lo, bh, acc[t0]);
    if (U == 2) acc[t1] = mfma_bf16(a1.lo, bh, acc[t1]);
    acc[t0] = mfma_bf16(a0.hi, bl, acc[t0]);
    if (U == 2) acc[t1] = mfma_bf16(a1.hi, bl, acc[t1]);
    acc[t0] = mfma_bf16(a0.mid, bm, acc[t0]);
    if (U == 2) acc[t1] = mfma_bf16(a1.mid, bm, acc[t1]);
#endif
    acc[t0] = mfma_bf16(a0.mid, bh, acc[t0]);
    if (U == 2) {
      if (P + AHEAD + 1 < NP) q[(P + AHEAD + 1) % QN] = load(P + AHEAD + 1);
      gap(1);
      acc[t1] = mfma_bf16(a1.mid, bh, acc[t1]);
      gap(2);
    }
    acc[t0] = mfma_bf16(a0.hi, bm, acc[t0]);
    if (U == 2) {
      gap(3);
      dma(1);
      acc[t1] = mfma_bf16(a1.hi, bm, acc[t1]);
      gap(4);
    }
    acc[t0] = mfma_bf16(a0.hi, bh, acc[t0]);
    if (U == 2) {
      gap(5);
      dma(U);
      acc[t1] = mfma_bf16(a1.hi, bh, acc[t1]);
    } else {
      dma(U);
    }
    if (t1 == NT - 1) { bh = nh; bm = nm; bl = nl; }
    if (pr + U == npc) ring3_leave(R);
  }
  return;
#endif
  dyn_static_for<(AHEAD < NP ? AHEAD : NP)>([&](auto I) DYN_INLINE_LAMBDA { q[decltype(I)::value % QN] = load(decltype(I)::value); });
  __builtin_amdgcn_sched_barrier(0);
  dyn_static_for<NP / U>([&](auto I) DYN_INLINE_LAMBDA {
    constexpr int P = decltype(I)::value * U;
    constexpr int pr = P % CP;                                    // pair inside its chunk
    constexpr int npc = NP - (P - pr) < CP ? NP - (P - pr) : CP;  // pairs of this chunk
    constexpr int mid = (npc / 2) / U * U;                        // the chunk's barrier stands in front of this pair
    constexpr int g = P / NT, t0 = P % NT, t1 = t0 + U - 1;
    constexpr int um = (P % NT) / U;                              // unit within the k-group
    if constexpr (pr == mid) ring3_mid(R);
    // the micro-steps of gap k: thirds of a pair of slots of k-group g + 1
    auto gap = [&](auto K) DYN_INLINE_LAMBDA {
      dyn_static_for<NM>([&](auto M) DYN_INLINE_LAMBDA {
        constexpr int m = decltype(M)::value;
        constexpr int at = NT == 1 ? 0 : (NM <= 4 ? 2 + m : m * 6 / NM);
        if constexpr (at == decltype(K)::value && g + 1 < NG) {
          constexpr int step = um * NM + m, p2 = step / 3, st = step % 3, s0 = (g + 1) * 8 + 2 * p2;
          if constexpr (st == 0) v0 = s0 < NSLOTS ? feed(s0) : 0.f;
          if constexpr (st == 1) v1 = s0 + 1 < NSLOTS ? feed(s0 + 1) : 0.f;
          if constexpr (st == 2) {
            unsigned h_, m_, l_;
            split3_pair(v0, v1, h_, m_, l_);
            nh[p2] = h_; nm[p2] = m_; nl[p2] = l_;
          }
        }
      });
    };
    // the weight stream: this unit's share of the pieces of the chunk being filled (see ring3_feed), after `done` of the unit's U pairs
    auto dma = [&](auto D) DYN_INLINE_LAMBDA {
      constexpr int done = decltype(D)::value;
      constexpr int wend = mid - B6D_DMA_MARGIN;  // (first half of the chunk: the window closes here)
      if constexpr (pr >= mid) ring3_feed(R, 0, pr + done - mid, npc - mid);
      else if constexpr (wend <= 0) ring3_feed(R, 1, 1, 1);
      else ring3_feed(R, 1, pr + done < wend ? pr + done : wend, wend);
    };
    // ---- gap 0
    if constexpr (P + AHEAD < NP) q[(P + AHEAD) % QN] = load(P + AHEAD);
    gap(std::integral_constant<int, 0>{});
    const B6A& a0 = q[P % QN];
    const B6A& a1 = q[(P + U - 1) % QN];
#if DYN_SPLIT_TERMS == 6
    acc[t0] = mfma_pinned(a0.lo, bh, acc[t0]);
    if constexpr (U == 2) acc[t1] = mfma_pinned(a1.lo, bh, acc[t1]);
    acc[t0] = mfma_pinned(a0.hi, bl, acc[t0]);
    if constexpr (U == 2) acc[t1] = mfma_pinned(a1.hi, bl, acc[t1]);
    acc[t0] = mfma_pinned(a0.mid, bm, acc[t0]);
    if constexpr (U == 2) acc[t1] = mfma_pinned(a1.mid, bm, acc[t1]);
#endif
    acc[t0] = mfma_pinned(a0.mid, bh, acc[t0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (U == 2) {
      // ---- gap 1
      if constexpr (P + AHEAD + 1 < NP) q[(P + AHEAD + 1) % QN] = load(P + AHEAD + 1);
      gap(std::integral_constant<int, 1>{});
      acc[t1] = mfma_pinned(a1.mid, bh, acc[t1]);
      __builtin_amdgcn_sched_barrier(0);
      gap(std::integral_constant<int, 2>{});
    }
    acc[t0] = mfma_pinned(a0.hi, bm, acc[t0]);
    if constexpr (U == 2) {
      __builtin_amdgcn_sched_barrier(0);
      gap(std::integral_constant<int, 3>{});
      dma(std::integral_constant<int, 1>{});
      acc[t1] = mfma_pinned(a1.hi, bm, acc[t1]);
      __builtin_amdgcn_sched_barrier(0);
      gap(std::integral_constant<int, 4>{});
    }
    acc[t0] = mfma_pinned(a0.hi, bh, acc[t0]);
    if constexpr (U == 2) {
      __builtin_amdgcn_sched_barrier(0);
      gap(std::integral_constant<int, 5>{});
      dma(std::integral_constant<int, U>{});
      acc[t1] = mfma_pinned(a1.hi, bh, acc[t1]);
    } else {
      dma(std::integral_constant<int, U>{});
    }
    if constexpr (t1 == NT - 1) { bh = nh; bm = nm; bl = nl; }
    if constexpr (pr + U == npc) ring3_leave(R);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// One Linear layer whose packed pairs are RESIDENT in LDS (`w`: pair p at w + p * B6_PAIR_FLOATS, in consumption order): no ring, no barriers.  For a
// kernel whose whole weight set fits the LDS (k_static_blend: 104 KiB) and whose workgroups stay on the CU and walk many row tiles: the weights are
// copied once per workgroup instead of streamed once per 128 rows.  Schedule as mlp_layer_b6 (pair-granular pipeline, the next k-group's operand in slices).
template <int NT, int NSLOTS, int AHEAD = B6_AHEAD, class Feed>
__device__ __forceinline__ void mlp_layer_b6_lds(const float* w, f32x16 (&acc)[NT], Feed&& feed) {
  constexpr int NG = (NSLOTS + 7) / 8, NPR = NG * NT;
  static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "output tiles per layer");
  const int lane = threadIdx.x & 63;
  u32x4v bh, bm, bl, nh, nm, nl;
  b6_split_pairs<NSLOTS, 0, 4>(feed, 0, bh, bm, bl);
  nh = bh; nm = bm; nl = bl;
  B6A q[AHEAD + 1];
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < NPR) q[i] = b6_load_a(w + i * B6_PAIR_FLOATS, lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pr = 0; pr < NPR; ++pr) {
    const int g = pr / NT, t = pr % NT;
    if (pr + AHEAD < NPR) q[(pr + AHEAD) % (AHEAD + 1)] = b6_load_a(w + (pr + AHEAD) * B6_PAIR_FLOATS, lane);
    if (g + 1 < NG) {
      if (NT == 1) b6_split_pairs<NSLOTS, 0, 4>(feed, g + 1, nh, nm, nl);
      if (NT == 2 && t == 0) b6_split_pairs<NSLOTS, 0, 2>(feed, g + 1, nh, nm, nl);
      if (NT == 2 && t == 1) b6_split_pairs<NSLOTS, 2, 4>(feed, g + 1, nh, nm, nl);
      if (NT == 4 && t == 0) b6_split_pairs<NSLOTS, 0, 1>(feed, g + 1, nh, nm, nl);
      if (NT == 4 && t == 1) b6_split_pairs<NSLOTS, 1, 2>(feed, g + 1, nh, nm, nl);
      if (NT == 4 && t == 2) b6_split_pairs<NSLOTS, 2, 3>(feed, g + 1, nh, nm, nl);
      if (NT == 4 && t == 3) b6_split_pairs<NSLOTS, 3, 4>(feed, g + 1, nh, nm, nl);
      if (NT == 8 && t == 0) b6_split_pairs<NSLOTS, 0, 1>(feed, g + 1, nh, nm, nl);
      if (NT == 8 && t == 2) b6_split_pairs<NSLOTS, 1, 2>(feed, g + 1, nh, nm, nl);
      if (NT == 8 && t == 4) b6_split_pairs<NSLOTS, 2, 3>(feed, g + 1, nh, nm, nl);
      if (NT == 8 && t == 6) b6_split_pairs<NSLOTS, 3, 4>(feed, g + 1, nh, nm, nl);
    }
    const B6A& cur = q[pr % (AHEAD + 1)];
#if DYN_SPLIT_TERMS == 6
    acc[t] = mfma_bf16(cur.lo, bh, acc[t]);
    acc[t] = mfma_bf16(cur.hi, bl, acc[t]);
    acc[t] = mfma_bf16(cur.mid, bm, acc[t]);
#endif
    acc[t] = mfma_bf16(cur.mid, bh, acc[t]);
    acc[t] = mfma_bf16(cur.hi, bm, acc[t]);
    acc[t] = mfma_bf16(cur.hi, bh, acc[t]);
    if (t == NT - 1) { bh = nh; bm = nm; bl = nl; }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One output tile of an NT-tile layer whose packed chunks the wave reads straight from global memory into registers, no ring and no
// barriers (the layer's rows are shared by the whole workgroup and every wave needs a different eighth of the weights, so staging
// them through LDS would cost whole chunks of DMA and barriers for a handful of MFMAs per wave).  b6_tile_prefetch is issued early
// (its latency hides under whatever follows), b6_tile_apply consumes the registers.
template <int NSLOTS>
struct B6TileW {
  B6A a[(NSLOTS + 7) / 8];
};
template <int NT, int NSLOTS>
__device__ __forceinline__ void b6_tile_prefetch(const float* layer_chunks, int my_tile, B6TileW<NSLOTS>& w) {
  constexpr int NG = (NSLOTS + 7) / 8;
  constexpr int GPC = B6_CHUNK_PAIRS / NT;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int g = 0; g < NG; ++g) w.a[g] = b6_load_a(layer_chunks + (g / GPC) * B6_CHUNK + ((g % GPC) * NT + my_tile) * B6_PAIR_FLOATS, lane);
}
template <int NSLOTS, class Feed>
__device__ __forceinline__ void b6_tile_apply(const B6TileW<NSLOTS>& w, f32x16& acc, Feed&& feed) {
  constexpr int NG = (NSLOTS + 7) / 8;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    u32x4v bh, bm, bl;
    b6_split_pairs<NSLOTS, 0, 4>(feed, g, bh, bm, bl);
#if DYN_SPLIT_TERMS == 6
    acc = mfma_bf16(w.a[g].lo, bh, acc);
    acc = mfma_bf16(w.a[g].hi, bl, acc);
    acc = mfma_bf16(w.a[g].mid, bm, acc);
#endif
    acc = mfma_bf16(w.a[g].mid, bh, acc);
    acc = mfma_bf16(w.a[g].hi, bm, acc);
    acc = mfma_bf16(w.a[g].hi, bh, acc);
  }
}

// ---- the names the network kernels use for the two-slot ring and its layer loop ---------------------------------------
typedef WeightRing6 NetRing;
#define NET_CHUNK B6_CHUNK
#define net_ring_init ring6_init
#define net_ring_init_t(R, stream, total, lds, threads) ring6_init(R, stream, total, lds, 1 << 30, 0, threads)
#define net_ring_init_1(R, stream, total, lds, threads) ring6_init(R, stream, total, lds, 1 << 30, 0, threads, 1)
#define net_layer mlp_layer_b6
__host__ __device__ constexpr int net_layer_chunks(int NT, int NSLOTS) { return b6_layer_chunks(NT, NSLOTS); }
