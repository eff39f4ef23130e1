// The 2-D feature encoder that runs right before the per-ray path (reference ibrnet/feature_network.py:179-311): the part of
// ResNet.forward that is executed -- conv 7x7 / 2 (3 -> 64, reflect padding) -> InstanceNorm -> ReLU -> layer1 (three BasicBlocks of
// 3x3 convolutions, the first with stride 2 and a 1x1 / 2 shortcut, InstanceNorm after every convolution) -> 1x1 convolution with
// bias -> 32 coarse + 32 fine channels at 1/4 resolution.
//
// MI355X form.  Activations are channels-last fp32 ([N,H,W,64]: one pixel = 256 contiguous bytes), so the outputs ARE the
// [V,Hf,Wf,32] maps the gather kernel taps (no NCHW -> NHWC repack) and the source images are consumed as the data loader
// stores them ([V,H,W,3]).  Every convolution is an implicit GEMM on the split-product MFMA engine of dyn_mlp.h (fp32 operands
// as exact sums of two half floats, 3 MFMAs per K = 16, fp32 accumulate: fp32-class products), evaluated transposed like the
// networks: out^T [channels x pixels] = W [channels x K] . patch^T [K x pixels], K = (ky, kx, input channel).  One wavefront owns
// 32 consecutive output pixels of one row and all 64 output channels; the lane of a pixel gathers its own K-slices (8 consecutive
// input channels = 32 bytes) straight from L1/L2 -- a 72x128x64 map is 2.4 MB -- with reflect padding folded into the index.
// The whole weight set of a convolution (3x3x64x64 as half-float hi | mid images: 144 KiB) sits in LDS for the lifetime of a
// workgroup, which walks a strip of output rows of one image.
// InstanceNorm needs per-(image, channel) statistics of a convolution's whole output: every convolution adds per-channel sums and
// sums of squares of what it writes to a small fp64 table, and the CONSUMER of a tensor applies normalisation + affine + ReLU
// (+ residual) while it loads it (one fma + one max per value).  Block outputs are materialised once by a small elementwise kernel.
#include <math.h>
#include <string.h>

#include <vector>

#include "dyn_host.h"
#include "dyn_mlp.h"
#include "dyn_pack.h"

#define ENC_C 64            // channels of every activation
#define ENC_THREADS 512     // 8 waves = 8 tiles of 32 output pixels per workgroup iteration
#define ENC_EPS 1e-5        // nn.InstanceNorm2d default

// state-dict order of the tensors handed to dyn_encoder_pack
enum {
  EN_CONV1_W, EN_BN1_G, EN_BN1_B,
  EN_B0_C1_W, EN_B0_N1_G, EN_B0_N1_B, EN_B0_C2_W, EN_B0_N2_G, EN_B0_N2_B, EN_B0_DS_W, EN_B0_DS_G, EN_B0_DS_B,
  EN_B1_C1_W, EN_B1_N1_G, EN_B1_N1_B, EN_B1_C2_W, EN_B1_N2_G, EN_B1_N2_B,
  EN_B2_C1_W, EN_B2_N1_G, EN_B2_N1_B, EN_B2_C2_W, EN_B2_N2_G, EN_B2_N2_B,
  EN_OUT_W, EN_OUT_B, EN_NUM_TENSORS
};

// k-groups (16 K values each) of the three convolution shapes
#define K7_ROW 24                        /* conv 7x7: one kernel row = 7 px x 3 ch = 21 floats, padded to 24 */
#define K7_GROUPS ((7 * K7_ROW + 15) / 16)  /* 11 */
#define K3_GROUPS (9 * 4)
#define K1_GROUPS 4
constexpr size_t enc_image_floats(int groups) { return (size_t)groups * 2 * B6_PAIR_FLOATS; }  // 2 output tiles
// blob layout (floats): packed weight images, then the affine / bias tables
constexpr size_t EN_OFF_CONV1 = 0;
constexpr size_t EN_OFF_C3 = EN_OFF_CONV1 + enc_image_floats(K7_GROUPS);            // six 3x3 convolutions: b0c1 b0c2 b1c1 b1c2 b2c1 b2c2
constexpr size_t EN_OFF_DS = EN_OFF_C3 + 6 * enc_image_floats(K3_GROUPS);
constexpr size_t EN_OFF_OUT = EN_OFF_DS + enc_image_floats(K1_GROUPS);
constexpr size_t EN_OFF_AFFINE = EN_OFF_OUT + enc_image_floats(K1_GROUPS);          // 8 norms x [gamma 64 | beta 64]
constexpr size_t EN_OFF_BIAS = EN_OFF_AFFINE + 8 * 128;
constexpr size_t EN_BLOB_FLOATS = EN_OFF_BIAS + 64;
enum { NORM_BN1, NORM_B0N1, NORM_B0N2, NORM_B0DS, NORM_B1N1, NORM_B1N2, NORM_B2N1, NORM_B2N2 };

namespace {

// one convolution -> the engine's A images: pair (g, t) = [hi 1 KiB | mid 1 KiB], lane (n = l & 31, h = l >> 5), element e: W[32 t + n][k = 16 g + 8 h + e]
template <class KFn>
void pack_conv(float* dst, int groups, KFn&& w_of) {
  unsigned short* img = reinterpret_cast<unsigned short*>(dst);
  for (int g = 0; g < groups; ++g)
    for (int t = 0; t < 2; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const float w = w_of(32 * t + (lane & 31), 16 * g + 8 * (lane >> 5) + e);
          unsigned short hi, mid, lo;
          split_weight(w, hi, mid, lo);
          const size_t pair = (size_t)(g * 2 + t) * B6_PAIR_FLOATS * 2;  // in 16-bit units
          img[pair + lane * 8 + e] = hi;
          img[pair + 512 + lane * 8 + e] = mid;
#if DYN_SPLIT_PARTS == 3
          img[pair + 1024 + lane * 8 + e] = lo;
#endif
        }
}

}  // namespace

extern "C" size_t dyn_encoder_blob_floats(void) { return EN_BLOB_FLOATS; }

extern "C" int dyn_encoder_pack(const float* const* T, float* blob, size_t blob_floats) {
  DYN_REQUIRE(T && blob, "dyn_encoder_pack: null pointer");
  DYN_REQUIRE(blob_floats >= EN_BLOB_FLOATS, "dyn_encoder_pack: blob too small");
  for (int i = 0; i < EN_NUM_TENSORS; ++i) DYN_REQUIRE(T[i] != nullptr, "dyn_encoder_pack: tensor %d is NULL", i);
  g_pack_range_error = false;
  memset(blob, 0, EN_BLOB_FLOATS * sizeof(float));
  {
    const float* W = T[EN_CONV1_W];  // [64, 3, 7, 7]
    pack_conv(blob + EN_OFF_CONV1, K7_GROUPS, [=](int oc, int k) -> float {
      const int ky = k / K7_ROW, j = k % K7_ROW;
      if (ky >= 7 || j >= 21) return 0.f;
      const int kx = j / 3, ic = j % 3;
      return W[((oc * 3 + ic) * 7 + ky) * 7 + kx];
    });
  }
  const int c3[6] = {EN_B0_C1_W, EN_B0_C2_W, EN_B1_C1_W, EN_B1_C2_W, EN_B2_C1_W, EN_B2_C2_W};
  for (int c = 0; c < 6; ++c) {
    const float* W = T[c3[c]];  // [64, 64, 3, 3]
    pack_conv(blob + EN_OFF_C3 + c * enc_image_floats(K3_GROUPS), K3_GROUPS, [=](int oc, int k) -> float {
      const int tap = k / 64, ic = k % 64;
      return W[((oc * 64 + ic) * 3 + tap / 3) * 3 + tap % 3];
    });
  }
  {
    const float* W = T[EN_B0_DS_W];  // [64, 64, 1, 1]
    pack_conv(blob + EN_OFF_DS, K1_GROUPS, [=](int oc, int k) -> float { return W[oc * 64 + k]; });
    const float* Wo = T[EN_OUT_W];
    pack_conv(blob + EN_OFF_OUT, K1_GROUPS, [=](int oc, int k) -> float { return Wo[oc * 64 + k]; });
  }
  const int norms[8][2] = {{EN_BN1_G, EN_BN1_B}, {EN_B0_N1_G, EN_B0_N1_B}, {EN_B0_N2_G, EN_B0_N2_B}, {EN_B0_DS_G, EN_B0_DS_B},
                           {EN_B1_N1_G, EN_B1_N1_B}, {EN_B1_N2_G, EN_B1_N2_B}, {EN_B2_N1_G, EN_B2_N1_B}, {EN_B2_N2_G, EN_B2_N2_B}};
  for (int n = 0; n < 8; ++n)
    for (int c = 0; c < 64; ++c) {
      blob[EN_OFF_AFFINE + n * 128 + c] = T[norms[n][0]][c];
      blob[EN_OFF_AFFINE + n * 128 + 64 + c] = T[norms[n][1]][c];
    }
  for (int c = 0; c < 64; ++c) blob[EN_OFF_BIAS + c] = T[EN_OUT_B][c];
  DYN_REQUIRE(!g_pack_range_error, "dyn_encoder_pack: a weight is outside the half-float range of the split engine (|w| >= 65504 or not finite)");
  return 0;
}

// -------------------------------------------------------------------------------------------------------------------
// device side
// -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {  // padding_mode='reflect': -1 -> 1, n -> n - 2
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

// how a convolution reads its input: 0 the tensor as stored; 1 relu(IN(in)); 2 relu(IN(in) + in2) with in2 stored as is
enum { LOAD_PLAIN = 0, LOAD_NORM_RELU = 1, LOAD_NORM_ADD_RELU = 2 };

struct ConvArgs {
  const float* in;         // [N, Hin, Win, CIN] channels-last
  const float* in2;        // [N, Hin, Win, 64] or NULL (LOAD_NORM_ADD_RELU)
  const double* stats_in;  // [N][64][2] sum, sum of squares of `in` (modes 1, 2)
  const float* affine_in;  // [gamma 64 | beta 64] of the norm applied to `in`
  const float* wimg;       // packed weight images
  const float* bias;       // [64] or NULL
  float* out;              // [N, Hout, Wout, 64] channels-last, or (split outputs) the first 32 channels [N, Hout, Wout, 32]
  float* out_hi;           // NULL, or the last 32 channels [N, Hout, Wout, 32]
  double* stats_out;       // [N][64][2] accumulated here, or NULL
  int N, Hin, Win, Hout, Wout;
};

// statistics of one (image, channel) -> the fused multiply-add of InstanceNorm + affine: y = x * sc + sh
__device__ __forceinline__ void norm_coeff(const double* st, float gamma, float beta, double inv_n, float& sc, float& sh) {
  const double mean = st[0] * inv_n;
  double var = st[1] * inv_n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const double rstd = 1.0 / sqrt(var + ENC_EPS);
  sc = (float)(rstd * (double)gamma);
  sh = (float)((double)beta - mean * rstd * (double)gamma);
}

template <int KH, int KW, int STRIDE, int CIN, int MODE>
__global__ void __launch_bounds__(ENC_THREADS, 2) k_enc_conv(ConvArgs p) {
  constexpr int PAD = KH / 2;
  constexpr int GROUPS = CIN == 3 ? K7_GROUPS : KH * KW * 4;
  float* lds = reinterpret_cast<float*>(dyn_smem);
  float* wl = lds;                                     // weight images
  float* coef = lds + enc_image_floats(GROUPS);        // [sc 64 | sh 64] of the input norm
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  // weights -> LDS (lane-linear images, 16 bytes per thread per step)
  {
    // global -> LDS DMA (no register round trip, all pieces in flight at once): each wave moves contiguous 1 KiB pieces, the LDS image is the
    // packed stream's own layout
    constexpr int PIECES = (int)(enc_image_floats(GROUPS) / 256);  // 1 KiB pieces
    const float* g = p.wimg + lane * 4;
#pragma unroll
    for (int i = 0; i < (PIECES + ENC_THREADS / 64 - 1) / (ENC_THREADS / 64); ++i) {
      const int piece = i * (ENC_THREADS / 64) + wave;
      if (piece < PIECES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 256),
                                         (__attribute__((address_space(3))) void*)(wl + piece * 256), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }
  // This workgroup's share of the batch: a contiguous run of the N * Hout output rows (one workgroup per CU: no tail generation of
  // workgroups).  The run may cross an image boundary: the per-image norm coefficients are tabulated for both images it can touch.
  const long rows_total = (long)p.N * p.Hout;
  const long r_lo = rows_total * blockIdx.x / gridDim.x, r_hi = rows_total * (blockIdx.x + 1) / gridDim.x;
  const int n_first = (int)(r_lo / p.Hout);
  if (MODE != LOAD_PLAIN && tid < 128) {
    const int which = tid >> 6, c = tid & 63;
    const int n = n_first + which < p.N ? n_first + which : p.N - 1;
    float sc, sh;
    norm_coeff(p.stats_in + ((long)n * 64 + c) * 2, p.affine_in[c], p.affine_in[64 + c], 1.0 / ((double)p.Hin * p.Win), sc, sh);
    coef[which * 128 + c] = sc;
    coef[which * 128 + 64 + c] = sh;
  }
  __syncthreads();

  const int tiles_x = (p.Wout + 31) / 32;
  const long n_tiles = (r_hi - r_lo) * tiles_x;
  float s1[2][16], s2[2][16];  // this lane's running per-channel sums of what it wrote (channel 32 t + fi(r, h)) for image n_cur
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
  int n_cur = -1;
  // the 32 pixel-lanes of a half hold partial sums of the same 32 channels: butterfly over the lanes, then one fp64 atomic per channel
  auto flush_stats = [&]() {
    if (p.stats_out == nullptr || n_cur < 0) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = s1[t][r], b = s2[t][r];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
        if (j == 0) {
          double* st = p.stats_out + ((long)n_cur * 64 + 32 * t + dyn_fi(r, h)) * 2;
          atomicAdd(st, (double)a);
          atomicAdd(st + 1, (double)b);
        }
        s1[t][r] = 0.f; s2[t][r] = 0.f;
      }
  };

  for (long tile = wave; tile < n_tiles; tile += ENC_THREADS / 64) {
    const long grow = r_lo + tile / tiles_x;  // row of the flattened (image, output row) list
    const int n = (int)(grow / p.Hout), oy = (int)(grow - (long)n * p.Hout), ox = (int)(tile % tiles_x) * 32 + j;
    if (n != n_cur) { flush_stats(); n_cur = n; }
    const float* cf = coef + (n - n_first) * 128;
    const float* inb = p.in + (long)n * p.Hin * p.Win * CIN;
    const float* in2b = MODE == LOAD_NORM_ADD_RELU ? p.in2 + (long)n * p.Hin * p.Win * 64 : nullptr;
    const bool live = ox < p.Wout;
    const int oxc = live ? ox : p.Wout - 1;  // idle lanes shadow the last pixel (loads stay in bounds, nothing is stored)
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = p.bias != nullptr ? p.bias[32 * t + dyn_fi(r, h)] : 0.f;
    // one K-group (16 K values: this lane's 8) through the split engine against both output tiles
    auto mma_group = [&](int g, const float (&v)[8]) {
      u32x4v bh, bm, bl;
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        unsigned h_, m_, l_;
        split3_pair(v[2 * p2], v[2 * p2 + 1], h_, m_, l_);
        bh[p2] = h_; bm[p2] = m_; bl[p2] = l_;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const B6A a = b6_load_a(wl + (size_t)(g * 2 + t) * B6_PAIR_FLOATS, lane);
#if DYN_SPLIT_TERMS == 6
        acc[t] = mfma_bf16(a.lo, bh, acc[t]);
        acc[t] = mfma_bf16(a.hi, bl, acc[t]);
        acc[t] = mfma_bf16(a.mid, bm, acc[t]);
#endif
        acc[t] = mfma_bf16(a.mid, bh, acc[t]);
        acc[t] = mfma_bf16(a.hi, bm, acc[t]);
        acc[t] = mfma_bf16(a.hi, bh, acc[t]);
      }
    };
    if constexpr (CIN == 3) {
      // K = (ky, [kx, ic] padded to 24): this lane's 8 values of group g are k = 16 g + 8 h + e, i.e. 8-float chunk c = 2 g + h of the padded
      // rows: kernel row ky = c / 3, floats [8 (c % 3), + 8) of that row's 21 (+ 3 zero-weight) floats.  Group g + 1 is in flight under group g.
      auto load_group = [&](int g, float (&v)[8]) {
        const int c = 2 * g + h;
        const int ky = (c * 11) >> 5;          // c / 3 for c < 32
        const int j0 = (c - 3 * ky) * 8;
        const int iy = reflect_idx(oy * STRIDE + (ky < 7 ? ky : 6) - PAD, p.Hin);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int jj = j0 + e;
          const int kx = (jj * 11) >> 5;        // jj / 3 for jj < 32
          const int ic = jj - 3 * kx;
          const int ix = reflect_idx(oxc * STRIDE + (kx < 7 ? kx : 6) - PAD, p.Win);
          const float x = inb[((long)iy * p.Win + ix) * 3 + ic];
          v[e] = (ky < 7 && jj < 21) ? x : 0.f;
        }
      };
      float cur[8], nxt[8];
      load_group(0, cur);
#pragma unroll 1
      for (int g = 0; g < GROUPS; ++g) {
        if (g + 1 < GROUPS) load_group(g + 1, nxt);
        mma_group(g, cur);
#pragma unroll
        for (int e = 0; e < 8; ++e) cur[e] = nxt[e];
      }
    } else {
      // K order of the weight images: k-group g = tap * 4 + cg.  The channel group is the OUTER loop, so that the 16 norm coefficients of
      // the lane's 8 channels sit in registers for all taps; operand fetches (2 float4 = this lane's 8 channels of one input pixel) run
      // PF (tap, cg) steps ahead of the MFMAs that consume them.
      constexpr int TAPS = KH * KW, STEPS = 4 * TAPS, PF = MODE == LOAD_NORM_ADD_RELU ? 4 : 8;
      struct Op { float4 a, b, ra, rb; };
      // element offsets of the lane's half of each tap's input pixel (reflect padding resolved once per tile, not once per fetch)
      int tap_off[TAPS];
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int ky = tap / KW, kx = tap - ky * KW;
        const int iy = reflect_idx(oy * STRIDE + ky - PAD, p.Hin), ix = reflect_idx(oxc * STRIDE + kx - PAD, p.Win);
        tap_off[tap] = (iy * p.Win + ix) * 64 + h * 8;
      }
      auto load_op = [&](int step, Op& o) {
        const int cg = step / TAPS, tap = step - cg * TAPS;
        const int off = tap_off[tap] + cg * 16;
        o.a = *reinterpret_cast<const float4*>(inb + off);
        o.b = *reinterpret_cast<const float4*>(inb + off + 4);
        if (MODE == LOAD_NORM_ADD_RELU) {
          o.ra = *reinterpret_cast<const float4*>(in2b + off);
          o.rb = *reinterpret_cast<const float4*>(in2b + off + 4);
        }
      };
      Op ring[PF];
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (i < STEPS) load_op(i, ring[i]);
      float scv[8], shv[8];
#pragma unroll
      for (int step = 0; step < STEPS; ++step) {
        const int cg = step / TAPS, tap = step - cg * TAPS;
        if (MODE != LOAD_PLAIN && tap == 0) {
          const float4 sa = *reinterpret_cast<const float4*>(cf + cg * 16 + h * 8), sb = *reinterpret_cast<const float4*>(cf + cg * 16 + h * 8 + 4);
          const float4 ha = *reinterpret_cast<const float4*>(cf + 64 + cg * 16 + h * 8), hb = *reinterpret_cast<const float4*>(cf + 64 + cg * 16 + h * 8 + 4);
          scv[0] = sa.x; scv[1] = sa.y; scv[2] = sa.z; scv[3] = sa.w; scv[4] = sb.x; scv[5] = sb.y; scv[6] = sb.z; scv[7] = sb.w;
          shv[0] = ha.x; shv[1] = ha.y; shv[2] = ha.z; shv[3] = ha.w; shv[4] = hb.x; shv[5] = hb.y; shv[6] = hb.z; shv[7] = hb.w;
        }
        const Op o = ring[step % PF];
        if (step + PF < STEPS) load_op(step + PF, ring[step % PF]);
        float v[8] = {o.a.x, o.a.y, o.a.z, o.a.w, o.b.x, o.b.y, o.b.z, o.b.w};
        if (MODE != LOAD_PLAIN) {
          float add[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (MODE == LOAD_NORM_ADD_RELU) {
            add[0] = o.ra.x; add[1] = o.ra.y; add[2] = o.ra.z; add[3] = o.ra.w; add[4] = o.rb.x; add[5] = o.rb.y; add[6] = o.rb.z; add[7] = o.rb.w;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], scv[e], shv[e]) + add[e], 0.f);
        }
        mma_group(tap * 4 + cg, v);
      }
    }
    if (live) {
      const long pix = ((long)n * p.Hout + oy) * p.Wout + ox;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float* o = p.out_hi == nullptr ? p.out + pix * 64 + 32 * t : (t == 0 ? p.out : p.out_hi) + pix * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q)  // registers 4 q .. 4 q + 3 are channels 8 q + 4 h + (0..3) of the tile
          *reinterpret_cast<float4*>(o + 8 * q + 4 * h) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[t][r] += acc[t][r]; s2[t][r] = fmaf(acc[t][r], acc[t][r], s2[t][r]); }
      }
    }
  }
  flush_stats();
}

// block output: out = relu(IN(a) + (IN(b) | b))   (BasicBlock.forward, feature_network.py:66-85)
__global__ void __launch_bounds__(256) k_enc_block_out(const float* __restrict__ a, const double* __restrict__ stats_a, const float* __restrict__ aff_a,
                                                       const float* __restrict__ b, const double* __restrict__ stats_b, const float* __restrict__ aff_b,
                                                       long hw, float* __restrict__ out) {
  float* coef = reinterpret_cast<float*>(dyn_smem);  // [4][64]
  const int n = blockIdx.y;
  if (threadIdx.x < 64) {
    float sc, sh;
    norm_coeff(stats_a + ((long)n * 64 + threadIdx.x) * 2, aff_a[threadIdx.x], aff_a[64 + threadIdx.x], 1.0 / (double)hw, sc, sh);
    coef[threadIdx.x] = sc; coef[64 + threadIdx.x] = sh;
    if (stats_b != nullptr) {
      norm_coeff(stats_b + ((long)n * 64 + threadIdx.x) * 2, aff_b[threadIdx.x], aff_b[64 + threadIdx.x], 1.0 / (double)hw, sc, sh);
      coef[128 + threadIdx.x] = sc; coef[192 + threadIdx.x] = sh;
    }
  }
  __syncthreads();
  const long total4 = hw * 16;  // float4 elements per image
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i & 15) * 4;
    const float4 x = reinterpret_cast<const float4*>(a)[(long)n * total4 + i];
    float4 y = reinterpret_cast<const float4*>(b)[(long)n * total4 + i];
    const float xs[4] = {x.x, x.y, x.z, x.w};
    float ys[4] = {y.x, y.y, y.z, y.w}, o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (stats_b != nullptr) ys[e] = fmaf(ys[e], coef[128 + c + e], coef[192 + c + e]);
      o[e] = fmaxf(fmaf(xs[e], coef[c + e], coef[64 + c + e]) + ys[e], 0.f);
    }
    reinterpret_cast<float4*>(out)[(long)n * total4 + i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// -------------------------------------------------------------------------------------------------------------------
// host
// -------------------------------------------------------------------------------------------------------------------
struct EncWs {
  int H1, W1, H2, W2;
  size_t off_c1, off_t[4], off_stats, total;  // c1: [N,H1,W1,64]; t[0..3]: [N,H2,W2,64] scratch maps; stats: 9 tables of [N][64][2] doubles
};
static EncWs enc_ws(int N, int H, int W) {
  EncWs w;
  w.H1 = (H + 2 * 3 - 7) / 2 + 1; w.W1 = (W + 2 * 3 - 7) / 2 + 1;
  w.H2 = (w.H1 + 2 - 3) / 2 + 1; w.W2 = (w.W1 + 2 - 3) / 2 + 1;
  size_t o = 0;
  w.off_c1 = o; o += (size_t)N * w.H1 * w.W1 * 64;
  for (int i = 0; i < 4; ++i) { w.off_t[i] = o; o += (size_t)N * w.H2 * w.W2 * 64; }
  o = (o + 3) & ~(size_t)3;
  w.off_stats = o; o += (size_t)9 * N * 64 * 2 * 2;  // doubles counted as 2 floats
  w.total = o;
  return w;
}
extern "C" size_t dyn_encoder_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H < 16 || W < 16) return 0;
  return enc_ws(N, H, W).total * sizeof(float);
}
extern "C" int dyn_encoder_out_size(int H, int W, int* Hf, int* Wf) {
  DYN_REQUIRE(Hf && Wf && H >= 16 && W >= 16, "dyn_encoder_out_size: bad argument");
  const EncWs w = enc_ws(1, H, W);
  *Hf = w.H2; *Wf = w.W2;
  return 0;
}

template <int KH, int KW, int STRIDE, int CIN, int MODE>
static int launch_conv(int slot, const char* name, ConvArgs a, hipStream_t stream) {
  constexpr int GROUPS = CIN == 3 ? K7_GROUPS : KH * KW * 4;
  const size_t lds = (enc_image_floats(GROUPS) + 256) * sizeof(float);
  // one workgroup per CU (the weight images take the LDS of a CU), each a contiguous share of the batch's output rows
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const long rows_total = (long)a.N * a.Hout;
  const long want = n_cu > a.N ? n_cu : a.N;  // a share never spans more than two images (the kernel tabulates the norm coefficients of two)
  const dim3 grid((unsigned)(rows_total < want ? rows_total : want));
  DYN_LAUNCH(slot, name, (k_enc_conv<KH, KW, STRIDE, CIN, MODE>), grid, dim3(ENC_THREADS), lds, stream, a);
  return 0;
}

extern "C" int dyn_encoder_forward(const DynEncoderParams* q, void* stream_) {
  DYN_REQUIRE(q, "dyn_encoder_forward: null params");
  DYN_REQUIRE(q->blob && q->images && q->coarse && q->fine && q->workspace, "dyn_encoder_forward: null pointer");
  DYN_REQUIRE(q->N > 0 && q->H >= 16 && q->W >= 16, "dyn_encoder_forward: need N > 0 and images of at least 16 x 16");
  const EncWs w = enc_ws(q->N, q->H, q->W);
  DYN_REQUIRE(q->workspace_bytes >= w.total * sizeof(float), "dyn_encoder_forward: workspace too small (%zu < %zu bytes)", q->workspace_bytes,
              w.total * sizeof(float));
  hipStream_t stream = (hipStream_t)stream_;
  float* ws = (float*)q->workspace;
  const float* B = q->blob;
  double* stats = reinterpret_cast<double*>(ws + w.off_stats);
  auto st = [&](int i) { return stats + (size_t)i * q->N * 64 * 2; };
  auto aff = [&](int n) { return B + EN_OFF_AFFINE + n * 128; };
  auto c3 = [&](int c) { return B + EN_OFF_C3 + c * enc_image_floats(K3_GROUPS); };
  if (hipMemsetAsync(stats, 0, (size_t)9 * q->N * 64 * 2 * sizeof(double), stream) != hipSuccess) {
    dyn_set_error("dyn_encoder_forward: hipMemsetAsync failed");
    return DYN_E_LAUNCH;
  }
  float *c1 = ws + w.off_c1, *t0 = ws + w.off_t[0], *t1 = ws + w.off_t[1], *t2 = ws + w.off_t[2], *t3 = ws + w.off_t[3];
  const long hw2 = (long)w.H2 * w.W2;
  const dim3 ew_grid(64, q->N), ew_blk(256);
  ConvArgs a;
  int rc;
  // conv1 7x7 / 2 on the images as stored ([N,H,W,3]); its norm + ReLU is applied by its two consumers
  a = ConvArgs{q->images, nullptr, nullptr, nullptr, B + EN_OFF_CONV1, nullptr, c1, nullptr, st(0), q->N, q->H, q->W, w.H1, w.W1};
  if ((rc = launch_conv<7, 7, 2, 3, LOAD_PLAIN>(DYN_K_ENC_CONV7, "k_enc_conv7", a, stream))) return rc;
  // layer1.0: conv3x3 / 2 and the 1x1 / 2 shortcut, both on relu(IN(c1))
  a = ConvArgs{c1, nullptr, st(0), aff(NORM_BN1), c3(0), nullptr, t0, nullptr, st(1), q->N, w.H1, w.W1, w.H2, w.W2};
  if ((rc = launch_conv<3, 3, 2, 64, LOAD_NORM_RELU>(DYN_K_ENC_CONV3, "k_enc_conv3s2", a, stream))) return rc;
  a = ConvArgs{c1, nullptr, st(0), aff(NORM_BN1), B + EN_OFF_DS, nullptr, t1, nullptr, st(3), q->N, w.H1, w.W1, w.H2, w.W2};
  if ((rc = launch_conv<1, 1, 2, 64, LOAD_NORM_RELU>(DYN_K_ENC_CONV1, "k_enc_conv1s2", a, stream))) return rc;
  a = ConvArgs{t0, nullptr, st(1), aff(NORM_B0N1), c3(1), nullptr, t2, nullptr, st(2), q->N, w.H2, w.W2, w.H2, w.W2};
  if ((rc = launch_conv<3, 3, 1, 64, LOAD_NORM_RELU>(DYN_K_ENC_CONV3, "k_enc_conv3", a, stream))) return rc;
  // out0 = relu(IN(conv2) + IN(shortcut)) -> t3
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "k_enc_block_out", k_enc_block_out, ew_grid, ew_blk, 1024, stream, t2, st(2), aff(NORM_B0N2), t1, st(3), aff(NORM_B0DS), hw2, t3);
  // layer1.1 on out0 (t3): conv1 -> t0, conv2 -> t1, out1 = relu(IN(conv2) + out0) -> t2
  a = ConvArgs{t3, nullptr, nullptr, nullptr, c3(2), nullptr, t0, nullptr, st(4), q->N, w.H2, w.W2, w.H2, w.W2};
  if ((rc = launch_conv<3, 3, 1, 64, LOAD_PLAIN>(DYN_K_ENC_CONV3, "k_enc_conv3", a, stream))) return rc;
  a = ConvArgs{t0, nullptr, st(4), aff(NORM_B1N1), c3(3), nullptr, t1, nullptr, st(5), q->N, w.H2, w.W2, w.H2, w.W2};
  if ((rc = launch_conv<3, 3, 1, 64, LOAD_NORM_RELU>(DYN_K_ENC_CONV3, "k_enc_conv3", a, stream))) return rc;
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "k_enc_block_out", k_enc_block_out, ew_grid, ew_blk, 1024, stream, t1, st(5), aff(NORM_B1N2), t3, (const double*)nullptr, (const float*)nullptr, hw2, t2);
  // layer1.2 on out1 (t2): conv1 -> t0, conv2 -> t1; out2 = relu(IN(conv2) + out1) is formed by the 1x1 output convolution as it loads
  a = ConvArgs{t2, nullptr, nullptr, nullptr, c3(4), nullptr, t0, nullptr, st(6), q->N, w.H2, w.W2, w.H2, w.W2};
  if ((rc = launch_conv<3, 3, 1, 64, LOAD_PLAIN>(DYN_K_ENC_CONV3, "k_enc_conv3", a, stream))) return rc;
  a = ConvArgs{t0, nullptr, st(6), aff(NORM_B2N1), c3(5), nullptr, t1, nullptr, st(7), q->N, w.H2, w.W2, w.H2, w.W2};
  if ((rc = launch_conv<3, 3, 1, 64, LOAD_NORM_RELU>(DYN_K_ENC_CONV3, "k_enc_conv3", a, stream))) return rc;
  a = ConvArgs{t1, t2, st(7), aff(NORM_B2N2), B + EN_OFF_OUT, B + EN_OFF_BIAS, q->coarse, q->fine, nullptr, q->N, w.H2, w.W2, w.H2, w.W2};
  if ((rc = launch_conv<1, 1, 1, 64, LOAD_NORM_ADD_RELU>(DYN_K_ENC_CONV1, "k_enc_out_conv", a, stream))) return rc;
  return 0;
}

// ===================================================================================================================
// Training form (SURVEY 8f-3 / train.py:272-281: the reference optimises feature_net too).  The inference kernels above keep nothing;
// the backward pass needs every convolution's input and output again, so the training form is built like the networks' (dyn_train.hip):
// every convolution is an explicit im2col (channels-last patches, K = (ky, kx, ic)) + dyn_train_gemm in its three roles, and small row
// kernels do what sits between -- InstanceNorm (+ affine, + residual, + ReLU) forward and backward with per-(image, channel) statistics
// in fp64 tables, the scatter of patch gradients back onto the input map.  Host side: dynibar_amd/train_encoder.py.
// ===================================================================================================================
struct EncGeom {
  int N, Hin, Win, C, KH, KW, stride, pad, Hout, Wout;
};
// col[row, tap * C + c] = in[n, reflect(oy * stride - pad + ky), reflect(ox * stride - pad + kx), c]; row = (n * Hout + oy) * Wout + ox
// C not a multiple of four (the images' three channels): one thread per four consecutive elements of a patch row (a 16-byte store; the
// image is small and cached, the patch matrix is what costs) -- ldc a multiple of four, the tail beyond KH * KW * C written as zeros
__global__ void __launch_bounds__(256) k_enc_im2col_any(EncGeom q, const float* __restrict__ in, float* __restrict__ col, long ldc) {
  const int q4 = (int)(ldc / 4), K = q.KH * q.KW * q.C;
  const long total = (long)q.N * q.Hout * q.Wout * q4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int piece = (int)(idx % q4);
  const long row = idx / q4;
  const int ox = (int)(row % q.Wout), oy = (int)((row / q.Wout) % q.Hout), n = (int)(row / ((long)q.Wout * q.Hout));
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * piece + e;
    if (j < K) {
      const int tap = j / q.C, c = j - tap * q.C;
      const int iy = reflect_idx(oy * q.stride - q.pad + tap / q.KW, q.Hin), ix = reflect_idx(ox * q.stride - q.pad + tap % q.KW, q.Win);
      v[e] = in[(((long)n * q.Hin + iy) * q.Win + ix) * q.C + c];
    } else {
      v[e] = 0.f;
    }
  }
  reinterpret_cast<float4*>(col + row * ldc)[piece] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void __launch_bounds__(256) k_enc_im2col(EncGeom q, const float* __restrict__ in, float* __restrict__ col, long ldc) {
  const int cq = q.C / 4;  // float4 pieces per tap (C % 4 == 0; k_enc_im2col_any otherwise)
  const long total = (long)q.N * q.Hout * q.Wout * q.KH * q.KW * cq;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int piece = (int)(idx % cq);
  const long rt = idx / cq;
  const int tap = (int)(rt % (q.KH * q.KW));
  const long row = rt / (q.KH * q.KW);
  const int ox = (int)(row % q.Wout), oy = (int)((row / q.Wout) % q.Hout), n = (int)(row / ((long)q.Wout * q.Hout));
  const int iy = reflect_idx(oy * q.stride - q.pad + tap / q.KW, q.Hin), ix = reflect_idx(ox * q.stride - q.pad + tap % q.KW, q.Win);
  const float* src = in + (((long)n * q.Hin + iy) * q.Win + ix) * q.C;
  float* dst = col + row * ldc + (long)tap * q.C;
  reinterpret_cast<float4*>(dst)[piece] = reinterpret_cast<const float4*>(src)[piece];
}
// the adjoint: din[n, iy, ix, c] += sum over the (row, tap) pairs whose patch element is this pixel.  Gather form, one thread per (input
// pixel, four channels): a tap (ky, kx) reads row index v = oy * stride - pad + ky, mirrored into the map by reflect_idx, so pixel iy is
// read through v = iy, through v = -iy (top border, 1 <= iy <= pad) and through v = 2 (H - 1) - iy (bottom border) -- at most three
// candidates per axis, each valid when (v + pad - ky) is a multiple of the stride inside the output.  (A scatter with fp32 atomics, the
// first version, took 1.15 ms per 3x3 convolution of 18 quarter-resolution maps: 95 M atomics; this one reads the patch gradients once.)
__device__ __forceinline__ int enc_mirror_candidates(int i, int n, int pad, int (&v)[3]) {
  int c = 0;
  v[c++] = i;
  if (i >= 1 && i <= pad) v[c++] = -i;
  if (i <= n - 2 && i >= n - 1 - pad) v[c++] = 2 * (n - 1) - i;
  return c;
}
__global__ void __launch_bounds__(256) k_enc_col2im(EncGeom q, const float* __restrict__ dcol, long ldc, float* __restrict__ din) {
  const int cq = q.C / 4;
  const long total = (long)q.N * q.Hin * q.Win * cq;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int piece = (int)(idx % cq);
  const long pix = idx / cq;
  const int ix = (int)(pix % q.Win), iy = (int)((pix / q.Win) % q.Hin), n = (int)(pix / ((long)q.Win * q.Hin));
  int vy[3], vx[3];
  const int ny = enc_mirror_candidates(iy, q.Hin, q.pad, vy), nx = enc_mirror_candidates(ix, q.Win, q.pad, vx);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < q.KH; ++ky)
    for (int a = 0; a < ny; ++a) {
      const int ty = vy[a] + q.pad - ky;
      if (ty < 0 || ty % q.stride != 0 || ty / q.stride >= q.Hout) continue;
      const int oy = ty / q.stride;
      for (int kx = 0; kx < q.KW; ++kx)
        for (int b = 0; b < nx; ++b) {
          const int tx = vx[b] + q.pad - kx;
          if (tx < 0 || tx % q.stride != 0 || tx / q.stride >= q.Wout) continue;
          const long row = ((long)n * q.Hout + oy) * q.Wout + tx / q.stride;
          const float4 v = reinterpret_cast<const float4*>(dcol + row * ldc + (long)(ky * q.KW + kx) * q.C)[piece];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
  float4* dst = reinterpret_cast<float4*>(din + pix * q.C) + piece;
  const float4 o = *dst;
  *dst = make_float4(o.x + acc.x, o.y + acc.y, o.z + acc.z, o.w + acc.w);
}
static int enc_geom_check(const EncGeom& q, const char* who) {
  DYN_REQUIRE(q.N > 0 && q.Hin > 0 && q.Win > 0 && q.C > 0 && q.KH > 0 && q.KW > 0 && q.stride > 0 && q.pad >= 0 && q.pad < q.Hin && q.pad < q.Win,
              "%s: bad geometry", who);
  DYN_REQUIRE(q.Hout == (q.Hin + 2 * q.pad - q.KH) / q.stride + 1 && q.Wout == (q.Win + 2 * q.pad - q.KW) / q.stride + 1, "%s: output size %d x %d does not match", who,
              q.Hout, q.Wout);
  return 0;
}
extern "C" int dyn_enc_im2col(const float* in, int N, int Hin, int Win, int C, int KH, int KW, int stride, int pad, int Hout, int Wout, float* col,
                              long ldc, void* stream) {
  DYN_REQUIRE(in && col && ldc >= (long)KH * KW * C, "dyn_enc_im2col: bad arguments");
  const EncGeom q{N, Hin, Win, C, KH, KW, stride, pad, Hout, Wout};
  if (int rc = enc_geom_check(q, "dyn_enc_im2col")) return rc;
  DYN_REQUIRE((ldc & 3) == 0 && ((uintptr_t)col & 15) == 0 && ((C & 3) != 0 || ((uintptr_t)in & 15) == 0),
              "dyn_enc_im2col: ldc a multiple of 4 floats, 16-byte-aligned patch matrix (and map, when C is a multiple of 4)");
  if ((C & 3) != 0) {
    const long total4 = (long)N * Hout * Wout * (ldc / 4);
    DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_im2col", k_enc_im2col_any, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, in, col, ldc);
    return 0;
  }
  const long total = (long)N * Hout * Wout * KH * KW * (C / 4);
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_im2col", k_enc_im2col, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, in, col, ldc);
  return 0;
}
extern "C" int dyn_enc_col2im(const float* dcol, long ldc, int N, int Hin, int Win, int C, int KH, int KW, int stride, int pad, int Hout, int Wout,
                              float* din, void* stream) {
  DYN_REQUIRE(dcol && din && ldc >= (long)KH * KW * C && (C & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)dcol | (uintptr_t)din) & 15) == 0, "dyn_enc_col2im: bad arguments (C, ldc multiples of 4, aligned)");
  const EncGeom q{N, Hin, Win, C, KH, KW, stride, pad, Hout, Wout};
  if (int rc = enc_geom_check(q, "dyn_enc_col2im")) return rc;
  const long total = (long)N * Hin * Win * (C / 4);
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_col2im", k_enc_col2im, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, dcol, ldc, din);
  return 0;
}

// ---- InstanceNorm over [N, HW, 64] maps.  grid (chunks of ENC_IN_CHUNK pixels, N), 256 threads = 16 float4 channel groups x 16 pixel lanes ----
#define ENC_IN_CHUNK 256  // pixels per block: 18 quarter-resolution maps give 648 blocks (1024 left a third of the CUs idle)
// stats[n][c] = {sum x, sum x^2} (fp64, zeroed by the caller)
__global__ void __launch_bounds__(256) k_enc_in_stats(const float4* __restrict__ x, long HW, double* __restrict__ stats) {
  float* red = reinterpret_cast<float*>(dyn_smem);  // [16 pixel lanes][64 ch][2]
  const int n = blockIdx.y, g = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const long p0 = (long)blockIdx.x * ENC_IN_CHUNK, p1 = p0 + ENC_IN_CHUNK < HW ? p0 + ENC_IN_CHUNK : HW;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
  for (long p = p0 + pl; p < p1; p += 16) {
    const float4 v = x[((long)n * HW + p) * 16 + g];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
  }
  float* r = red + (pl * 64 + 4 * g) * 2;
  r[0] = s.x; r[1] = s2.x; r[2] = s.y; r[3] = s2.y; r[4] = s.z; r[5] = s2.z; r[6] = s.w; r[7] = s2.w;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
    double t = 0.0;
    for (int l = 0; l < 16; ++l) t += (double)red[(l * 64 + c) * 2 + k];
    atomicAdd(stats + ((long)n * 64 + c) * 2 + k, t);
  }
}
// per-block table of the (image, channel) normalisation: mean, rstd (and gamma * rstd, beta - mean * gamma * rstd) from the fp64 sums
__device__ __forceinline__ void enc_in_table(const double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int n,
                                             long HW, float* tab) {  // tab [4][64]: mean, rstd, sc, sh
  if (threadIdx.x < 64) {
    const int c = threadIdx.x;
    const double inv = 1.0 / (double)HW, mean = stats[((long)n * 64 + c) * 2] * inv;
    double var = stats[((long)n * 64 + c) * 2 + 1] * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + ENC_EPS);
    tab[c] = (float)mean;
    tab[64 + c] = (float)rstd;
    tab[128 + c] = (float)(rstd * (double)gamma[c]);
    tab[192 + c] = (float)((double)beta[c] - mean * rstd * (double)gamma[c]);
  }
  __syncthreads();
}
// y = relu?(IN(x) * gamma + beta (+ res))
__global__ void __launch_bounds__(256) k_enc_in_apply(const float4* __restrict__ x, const double* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float4* __restrict__ res, int relu, long HW,
                                                      float4* __restrict__ y) {
  float* tab = reinterpret_cast<float*>(dyn_smem);
  const int n = blockIdx.y, g = threadIdx.x & 15, pl = threadIdx.x >> 4;
  enc_in_table(stats, gamma, beta, n, HW, tab);
  const float4 sc = *reinterpret_cast<const float4*>(tab + 128 + 4 * g), sh = *reinterpret_cast<const float4*>(tab + 192 + 4 * g);
  const long p0 = (long)blockIdx.x * ENC_IN_CHUNK, p1 = p0 + ENC_IN_CHUNK < HW ? p0 + ENC_IN_CHUNK : HW;
  for (long p = p0 + pl; p < p1; p += 16) {
    const long i = ((long)n * HW + p) * 16 + g;
    const float4 v = x[i];
    float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (res != nullptr) {
      const float4 r = res[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    y[i] = o;
  }
}
// backward, first pass: sums2[n][c] = {sum dyr, sum dyr * xhat} with dyr = relu ? dy * (y > 0) : dy, xhat = (x - mean) * rstd
__global__ void __launch_bounds__(256) k_enc_in_bwd_stats(const float4* __restrict__ dy, const float4* __restrict__ y, int relu,
                                                          const float4* __restrict__ x, const double* __restrict__ stats, long HW,
                                                          double* __restrict__ sums2) {
  float* tab = reinterpret_cast<float*>(dyn_smem);  // [256] table, then [16][64][2] partial sums
  float* red = tab + 256;
  const int n = blockIdx.y, g = threadIdx.x & 15, pl = threadIdx.x >> 4;
  if (threadIdx.x < 64) {
    const int c = threadIdx.x;
    const double inv = 1.0 / (double)HW, mean = stats[((long)n * 64 + c) * 2] * inv;
    double var = stats[((long)n * 64 + c) * 2 + 1] * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    tab[c] = (float)mean;
    tab[64 + c] = (float)(1.0 / sqrt(var + ENC_EPS));
  }
  __syncthreads();
  const float4 mean = *reinterpret_cast<const float4*>(tab + 4 * g), rstd = *reinterpret_cast<const float4*>(tab + 64 + 4 * g);
  const long p0 = (long)blockIdx.x * ENC_IN_CHUNK, p1 = p0 + ENC_IN_CHUNK < HW ? p0 + ENC_IN_CHUNK : HW;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
  for (long p = p0 + pl; p < p1; p += 16) {
    const long i = ((long)n * HW + p) * 16 + g;
    float4 d = dy[i];
    if (relu) {
      const float4 o = y[i];
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    const float4 v = x[i];
    s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    s2.x += d.x * (v.x - mean.x) * rstd.x; s2.y += d.y * (v.y - mean.y) * rstd.y;
    s2.z += d.z * (v.z - mean.z) * rstd.z; s2.w += d.w * (v.w - mean.w) * rstd.w;
  }
  float* r = red + (pl * 64 + 4 * g) * 2;
  r[0] = s.x; r[1] = s2.x; r[2] = s.y; r[3] = s2.y; r[4] = s.z; r[5] = s2.z; r[6] = s.w; r[7] = s2.w;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
    double t = 0.0;
    for (int l = 0; l < 16; ++l) t += (double)red[(l * 64 + c) * 2 + k];
    atomicAdd(sums2 + ((long)n * 64 + c) * 2 + k, t);
  }
}
// backward, second pass: dx = gamma * rstd * (dyr - S0 / HW - xhat * S1 / HW); dres (may be NULL) = dyr; the first block of an image adds
// its sums to dgamma (S1) and dbeta (S0)
__global__ void __launch_bounds__(256) k_enc_in_bwd_apply(const float4* __restrict__ dy, const float4* __restrict__ y, int relu,
                                                          const float4* __restrict__ x, const double* __restrict__ stats,
                                                          const double* __restrict__ sums2, const float* __restrict__ gamma, long HW,
                                                          float4* __restrict__ dx, float4* __restrict__ dres, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  float* tab = reinterpret_cast<float*>(dyn_smem);  // [5][64]: mean, rstd, gamma * rstd, S0 / HW, S1 / HW
  const int n = blockIdx.y, g = threadIdx.x & 15, pl = threadIdx.x >> 4;
  if (threadIdx.x < 64) {
    const int c = threadIdx.x;
    const double inv = 1.0 / (double)HW, mean = stats[((long)n * 64 + c) * 2] * inv;
    double var = stats[((long)n * 64 + c) * 2 + 1] * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + ENC_EPS);
    const double S0 = sums2[((long)n * 64 + c) * 2], S1 = sums2[((long)n * 64 + c) * 2 + 1];
    tab[c] = (float)mean;
    tab[64 + c] = (float)rstd;
    tab[128 + c] = (float)(rstd * (double)gamma[c]);
    tab[192 + c] = (float)(S0 * inv);
    tab[256 + c] = (float)(S1 * inv);
    if (blockIdx.x == 0) {
      atomicAdd(dgamma + c, (float)S1);
      atomicAdd(dbeta + c, (float)S0);
    }
  }
  __syncthreads();
  const float4 mean = *reinterpret_cast<const float4*>(tab + 4 * g), rstd = *reinterpret_cast<const float4*>(tab + 64 + 4 * g);
  const float4 gs = *reinterpret_cast<const float4*>(tab + 128 + 4 * g), m0 = *reinterpret_cast<const float4*>(tab + 192 + 4 * g);
  const float4 m1 = *reinterpret_cast<const float4*>(tab + 256 + 4 * g);
  const long p0 = (long)blockIdx.x * ENC_IN_CHUNK, p1 = p0 + ENC_IN_CHUNK < HW ? p0 + ENC_IN_CHUNK : HW;
  for (long p = p0 + pl; p < p1; p += 16) {
    const long i = ((long)n * HW + p) * 16 + g;
    float4 d = dy[i];
    if (relu) {
      const float4 o = y[i];
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
    if (dres != nullptr) dres[i] = d;
    const float4 v = x[i];
    dx[i] = make_float4(gs.x * (d.x - m0.x - (v.x - mean.x) * rstd.x * m1.x), gs.y * (d.y - m0.y - (v.y - mean.y) * rstd.y * m1.y),
                        gs.z * (d.z - m0.z - (v.z - mean.z) * rstd.z * m1.z), gs.w * (d.w - m0.w - (v.w - mean.w) * rstd.w * m1.w));
  }
}
static bool enc_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
extern "C" int dyn_enc_in_stats(const float* x, int N, long HW, double* stats, void* stream) {
  DYN_REQUIRE(x && stats && N > 0 && HW > 0 && enc_al16(x), "dyn_enc_in_stats: bad arguments");
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_in_stats", k_enc_in_stats, dim3((unsigned)((HW + ENC_IN_CHUNK - 1) / ENC_IN_CHUNK), N), dim3(256), 16 * 64 * 2 * 4,
             (hipStream_t)stream, reinterpret_cast<const float4*>(x), HW, stats);
  return 0;
}
extern "C" int dyn_enc_in_apply(const float* x, const double* stats, const float* gamma, const float* beta, const float* res, int relu, int N, long HW,
                                float* y, void* stream) {
  DYN_REQUIRE(x && stats && gamma && beta && y && N > 0 && HW > 0 && enc_al16(x) && enc_al16(y) && (res == nullptr || enc_al16(res)), "dyn_enc_in_apply: bad arguments");
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_in_apply", k_enc_in_apply, dim3((unsigned)((HW + ENC_IN_CHUNK - 1) / ENC_IN_CHUNK), N), dim3(256), 256 * 4,
             (hipStream_t)stream, reinterpret_cast<const float4*>(x), stats, gamma, beta, reinterpret_cast<const float4*>(res), relu, HW,
             reinterpret_cast<float4*>(y));
  return 0;
}
extern "C" int dyn_enc_in_bwd(const float* dy, const float* y, int relu, const float* x, const double* stats, const float* gamma, int N, long HW,
                              double* sums2, float* dx, float* dres, float* dgamma, float* dbeta, void* stream) {
  DYN_REQUIRE(dy && x && stats && gamma && sums2 && dx && dgamma && dbeta && N > 0 && HW > 0 && (!relu || y != nullptr), "dyn_enc_in_bwd: bad arguments");
  DYN_REQUIRE(enc_al16(dy) && enc_al16(x) && enc_al16(dx) && (y == nullptr || enc_al16(y)) && (dres == nullptr || enc_al16(dres)), "dyn_enc_in_bwd: 16-byte alignment");
  const dim3 grid((unsigned)((HW + ENC_IN_CHUNK - 1) / ENC_IN_CHUNK), N);
  if (hipMemsetAsync(sums2, 0, (size_t)N * 64 * 2 * sizeof(double), (hipStream_t)stream) != hipSuccess) {
    dyn_set_error("dyn_enc_in_bwd: hipMemsetAsync failed");
    return DYN_E_LAUNCH;
  }
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_in_bwd", k_enc_in_bwd_stats, grid, dim3(256), (256 + 16 * 64 * 2) * 4, (hipStream_t)stream,
             reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), relu, reinterpret_cast<const float4*>(x), stats, HW, sums2);
  DYN_LAUNCH(DYN_K_ENC_BLOCK, "dyn_enc_in_bwd", k_enc_in_bwd_apply, grid, dim3(256), 320 * 4, (hipStream_t)stream, reinterpret_cast<const float4*>(dy),
             reinterpret_cast<const float4*>(y), relu, reinterpret_cast<const float4*>(x), stats, sums2, gamma, HW, reinterpret_cast<float4*>(dx),
             reinterpret_cast<float4*>(dres), dgamma, dbeta);
  return 0;
}
