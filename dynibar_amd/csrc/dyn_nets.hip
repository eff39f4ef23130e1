// K3: the per-point multi-view networks (reference ibrnet/mlp_network.py) as register-resident fp32 MFMA chains.
// See dyn_mlp.h for the engine.  DynibarStatic (mlp_network.py:423-527) is three launches:
//   A  k_static_views : per point-view chain  ray_dir_fc -> [x ref_feature] -> mean/var -> base_fc -> vis_fc -> vis_fc2,
//                       then the visibility-weighted mean/var over the views of each point   (one wave = 32 point-views)
//   B  k_static_points: per point chain  geometry_fc -> 4-head ray attention over the S samples of a ray -> LayerNorm ->
//                       out_geometry_fc (sigma) and the point part of rgb_fc.0               (one wave = 32 points of one ray)
//   C  k_static_blend : per point-view  rgb_fc -> masked softmax over views -> blend of the source colours
// Between A and C the 128-wide per-view feature x is parked in HBM in the lanes' own register order (512 B per point-view).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#include "dyn_host.h"
#include "dyn_mlp.h"
#include "dyn_pack.h"

// ===================================================================================================================
// host-side packing
// ===================================================================================================================
namespace {

// value of the packed A operand for (output tile t, row i, k-step s, half h)
using SlotFn = std::function<float(int t, int i, int s, int h)>;

// B6 engine image of one layer (dyn_mlp.h): per (k-group of 8 slots, output tile) three lane-linear 1 KiB parts [hi | mid | lo]
thread_local int g_pack_chunk_pairs = B6_CHUNK_PAIRS;  // pairs per chunk of the stream being packed (the point kernels' streams use PTS_CP); per host thread: two threads may pack at once
void pack_layer_b6(std::vector<float>& out, int NT, int NSLOTS, const SlotFn& fn) {
  const int CPAIRS = g_pack_chunk_pairs, CFLOATS = CPAIRS * B6_PAIR_FLOATS;
  const int NG = (NSLOTS + 7) / 8, GPC = CPAIRS / NT, NCH = (NG + GPC - 1) / GPC;
  const size_t base = out.size();
  out.resize(base + (size_t)NCH * CFLOATS, 0.f);
  unsigned short* img = reinterpret_cast<unsigned short*>(out.data() + base);
  for (int c = 0; c < NCH; ++c)
    for (int gi = 0; gi < GPC; ++gi) {
      const int g = c * GPC + gi;
      if (g >= NG) continue;
      for (int t = 0; t < NT; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int s = g * 8 + e;
            if (s >= NSLOTS) continue;
            const float w = fn(t, lane & 31, s, lane >> 5);
            unsigned short hi, mid, lo;
            split_weight(w, hi, mid, lo);
            const size_t pair = (size_t)c * CFLOATS * 2 + (size_t)(gi * NT + t) * B6_PAIR_FLOATS * 2;  // in 16-bit units
            img[pair + 0 * 512 + lane * 8 + e] = hi;
            img[pair + 1 * 512 + lane * 8 + e] = mid;
            if (DYN_SPLIT_PARTS == 3) img[pair + 2 * 512 + lane * 8 + e] = lo;
          }
    }
}

void pack_net_layer(std::vector<float>& out, int NT, int NSLOTS, const SlotFn& fn) {
  pack_layer_b6(out, NT, NSLOTS, fn);
}

// input feature of a chained layer: k-step s, half h -> feature of the previous layer's output (D layout)
inline int chain_feature(int s, int h) { return 32 * (s / 16) + dyn_fi(s % 16, h); }

// Linear [nout, nin] (+bias) fed by a previous layer's D-layout output of `nin` features, bias in the extra k-step.
// col_of maps a chained feature index to the weight column (identity by default).
SlotFn chained(const float* W, const float* b, int nout, int nin, int ld, int col0 = 0) {
  return [=](int t, int i, int s, int h) -> float {
    const int n = 32 * t + i;
    if (n >= nout) return 0.f;
    const int nsteps_in = nin / 2;
    if (s < nsteps_in) return W[(size_t)n * ld + col0 + chain_feature(s, h)];
    if (s == nsteps_in && h == 0 && b != nullptr) return b[n];
    return 0.f;
  };
}

// [2][n] table of a single-output Linear over a D-layout activation of n features (times `scale`: DYN_ELU_PRE / DYN_ELU_POST, dyn_mlp.h)
void pack_rowtab(std::vector<float>& out, const float* w, int n, double scale = 1.0) {
  for (int h = 0; h < 2; ++h)
    for (int k = 0; k < n / 2; ++k) out.push_back((float)((double)w[chain_feature(k, h)] * scale));
}

// a chained layer whose weight slots (s < nsteps_w) and bias slot take different factors
SlotFn scaled_wb(SlotFn fn, int nsteps_w, double kw, double kb) {
  return [=](int t, int i, int s, int h) -> float { return (float)((double)fn(t, i, s, h) * (s < nsteps_w ? kw : kb)); };
}

// a layer image times a constant (the scaled-domain ELU's pack-time factors, dyn_mlp.h: elu_s)
SlotFn scaled(SlotFn fn, double scale) {
  if (scale == 1.0) return fn;
  return [=](int t, int i, int s, int h) -> float { return (float)((double)fn(t, i, s, h) * scale); };
}

}  // namespace

// ===================================================================================================================
// DynibarStatic
// ===================================================================================================================
// state-dict order of the tensors handed to dyn_static_net_pack (names as in mlp_network.py:319-421)
enum {
  ST_RAYDIR0_W, ST_RAYDIR0_B, ST_RAYDIR2_W, ST_RAYDIR2_B, ST_REFFEAT_W, ST_REFFEAT_B, ST_BASE0_W, ST_BASE0_B, ST_BASE2_W, ST_BASE2_B,
  ST_VIS0_W, ST_VIS0_B, ST_VIS2_W, ST_VIS2_B, ST_VISB0_W, ST_VISB0_B, ST_VISB2_W, ST_VISB2_B, ST_GEO0_W, ST_GEO0_B, ST_GEO2_W,
  ST_GEO2_B, ST_WQ, ST_WK, ST_WV, ST_FC, ST_LN_G, ST_LN_B, ST_OG0_W, ST_OG0_B, ST_OG2_W, ST_OG2_B, ST_RGB0_W, ST_RGB0_B,
  ST_RGB2_W, ST_RGB2_B, ST_RGB4_W, ST_RGB4_B, ST_S, ST_NUM_TENSORS
};

// k-step counts / tiles of every layer (shared by the packer and the kernels)
/* ray_dir_fc.0 (round 5: in two parts).  Its 103 inputs are [PE(pts) 33 | PE(source-ray Pluecker) 66 | ray_diff 4]: the first 33 are the same for
 * every view of a point, so their columns are applied once per point for the whole workgroup (L1P, like base_fc.0's statistics) and only the
 * per-view columns remain per row (L1V): 5 k-groups x 8 tiles x 3 products per wave instead of 7 (-39 of 894 MFMAs, -13 of 393 weight pairs streamed). */
#define SA_L1P_STEPS 17  /* per point: 15 cos|sin pairs of pts + (x, y), (z, -) */
#define SA_L1V_STEPS 36  /* per view: 30 cos|sin pairs of the Pluecker coordinates + 6 raw pairs (Pluecker 6, ray_diff 4, bias, -) */
#define SA_L2_STEPS 128  /* ray_dir_fc.2: 256 (bias = accumulator init) */
#define SA_NX 37         /* registers holding the 70-channel per-view feature (18 gathered + 16 + 3 computed) */
#define SA_L3_STEPS (3 * SA_NX + 1)
#define SA_L3P_STEPS (2 * SA_NX)  /* per-point part of base_fc.0: [mean | var] slots, evaluated once per point for the whole workgroup */
#define SA_L3V_STEPS (SA_NX + 1)  /* per-view part: the 70 channels + bias */
#define SA_L4_STEPS 128  /* base_fc.2 (bias = accumulator init) */
#define SA_L5_STEPS 64   /* vis_fc.0 / vis_fc.2 / vis_fc2.0 */
constexpr int SA_CHUNKS = net_layer_chunks(8, SA_L1P_STEPS) + net_layer_chunks(8, SA_L1V_STEPS) + net_layer_chunks(2, SA_L2_STEPS) + net_layer_chunks(8, SA_L3P_STEPS) + net_layer_chunks(8, SA_L3V_STEPS) +
                          net_layer_chunks(4, SA_L4_STEPS) + 3 * net_layer_chunks(4, SA_L5_STEPS);
// The point kernels (k_net_points) run one wave per SIMD on the interleaved layer loop with the three-slot ring (dyn_mlp.h, round 4).  Their LDS also
// holds the ray attention's K / V images (32.5 KiB), so their weight streams are packed in chunks of PTS_CP = 16 pairs (32 KiB; 3 slots = 96 KiB,
// what the two 48 KiB slots took).  DYN_POINTS_DUO = 0: the round-3 form (A/B builds; the 6-term bf16 build keeps it: its pairs are 3 KiB).
#ifndef DYN_POINTS_DUO
#define DYN_POINTS_DUO (DYN_SPLIT_TERMS == 3 ? 1 : 0)
#endif
#ifndef DYN_POINTS_PERSIST
#define DYN_POINTS_PERSIST 0
#endif
#if DYN_POINTS_DUO
#define PTS_CP 16
#define PTS_CHUNK (PTS_CP * B6_PAIR_FLOATS)
#define PTS_RING_SLOTS B6D_SLOTS
__host__ __device__ constexpr int pts_layer_chunks(int NT, int NSLOTS) { return b6_layer_chunks(NT, NSLOTS, PTS_CP); }
typedef WeightRing3 PtsRing;
#define pts_ring_init(R, stream, total, lds) ring3_init(R, stream, total, lds, DYN_NET_THREADS, PTS_CP)
#define PTS_LAYER(NT, NSLOTS) mlp_layer_b6_duo<NT, NSLOTS, PTS_CP>
#else
#define PTS_CHUNK NET_CHUNK
#define PTS_RING_SLOTS 2
__host__ __device__ constexpr int pts_layer_chunks(int NT, int NSLOTS) { return net_layer_chunks(NT, NSLOTS); }
typedef NetRing PtsRing;
#define pts_ring_init(R, stream, total, lds) net_ring_init_t(R, stream, total, lds, DYN_NET_THREADS)
#define PTS_LAYER(NT, NSLOTS) net_layer<NT, NSLOTS>
#endif
constexpr int SB_CHUNKS = pts_layer_chunks(8, 129) + pts_layer_chunks(4, 129) + 4 * pts_layer_chunks(4, 64) + 2 * pts_layer_chunks(4, 65);
#define SC_L11_STEPS 67
constexpr int SC_CHUNKS = net_layer_chunks(4, SC_L11_STEPS) + net_layer_chunks(2, 64);
// constant tables (floats): A: vis row [2][64] @0, vis_fc2.2 row [2][64] @128, b_vis @256, b_vis2 @257, |s| @258
#define SA_CT 848   /* + bias tables: base_fc.2 @272, vis_fc.0 @400, vis_fc.2 @528, vis_fc2.0 @656, ray_dir_fc.2 @784 (64) */
// B: ln gamma [2][64], ln beta [2][64], out_geometry_fc.2 row [2][64], its bias
#define SB_CT 400
// C: rgb_fc.4 row [2][32], bias
#define SC_CT 144   /* + bias table of rgb_fc.2 @80 (64) */
constexpr size_t ST_OFF_A = 0;
constexpr size_t ST_OFF_B = ST_OFF_A + (size_t)SA_CHUNKS * NET_CHUNK;
constexpr size_t ST_OFF_C = ST_OFF_B + (size_t)SB_CHUNKS * PTS_CHUNK;
constexpr size_t ST_OFF_CTA = ST_OFF_C + (size_t)SC_CHUNKS * NET_CHUNK;
constexpr size_t ST_OFF_CTB = ST_OFF_CTA + SA_CT;
constexpr size_t ST_OFF_CTC = ST_OFF_CTB + SB_CT;
constexpr size_t ST_OFF_REF = ST_OFF_CTC + SC_CT;  // ref_feature_fc.0: [35][66] then [35]
constexpr size_t ST_BLOB_FLOATS = ST_OFF_REF + 35 * 66 + 36;

// ===================================================================================================================
// DynibarDynamic (mlp_network.py:129-316): layer programs and blob layout
// ===================================================================================================================
enum {
  DT_RAYDIR0_W, DT_RAYDIR0_B, DT_RAYDIR2_W, DT_RAYDIR2_B, DT_BASE0_W, DT_BASE0_B, DT_BASE2_W, DT_BASE2_B, DT_VIS0_W, DT_VIS0_B, DT_VIS2_W,
  DT_VIS2_B, DT_VISB0_W, DT_VISB0_B, DT_VISB2_W, DT_VISB2_B, DT_GEO0_W, DT_GEO0_B, DT_GEO2_W, DT_GEO2_B, DT_WQ, DT_WK, DT_WV, DT_FC, DT_LN_G,
  DT_LN_B, DT_REFPTS0_W, DT_REFPTS0_B, DT_REFPTS2_W, DT_REFPTS2_B, DT_OG0_W, DT_OG0_B, DT_OG2_W, DT_OG2_B, DT_RGB0_W, DT_RGB0_B, DT_RGB2_W,
  DT_RGB2_B, DT_RGB4_W, DT_RGB4_B, DT_NUM_TENSORS
};
#define DA_NX 18                      /* registers holding the 35-channel per-view feature */
#define DA_L3_STEPS (3 * DA_NX + 1)   /* base_fc.0: x | mean | var | bias */
#define DA_L3P_STEPS (2 * DA_NX)
#define DA_L3V_STEPS (DA_NX + 1)
constexpr int DA_CHUNKS = net_layer_chunks(8, DA_L3P_STEPS) + net_layer_chunks(8, DA_L3V_STEPS) + net_layer_chunks(4, SA_L4_STEPS) + 3 * net_layer_chunks(4, SA_L5_STEPS);
constexpr int DB_CHUNKS = pts_layer_chunks(8, 129) + pts_layer_chunks(4, 129) + 4 * pts_layer_chunks(4, 64) + pts_layer_chunks(8, 81) +
                          pts_layer_chunks(4, 129) + pts_layer_chunks(4, 65) + pts_layer_chunks(4, 78) + pts_layer_chunks(2, 65);
// B table: ln gamma @0, ln beta @128, out_geometry_fc.2 row @256, its bias @384, rgb_fc.4 biases @385..387, rgb_fc.4 rows [3][2][32] @400
#define DB_CT 592
constexpr size_t DY_OFF_A = 0;
constexpr size_t DY_OFF_B = DY_OFF_A + (size_t)DA_CHUNKS * NET_CHUNK;
constexpr size_t DY_OFF_CTA = DY_OFF_B + (size_t)DB_CHUNKS * PTS_CHUNK;
constexpr size_t DY_OFF_CTB = DY_OFF_CTA + SA_CT;
constexpr size_t DY_OFF_POSENC = DY_OFF_CTB + DB_CT;                 // [256 positions][2][64]
constexpr size_t DY_OFF_TIME = DY_OFF_POSENC + 256 * 128;            // ray_dir_fc: W0 [256,21], b0 [256], W2 [35,256], b2 [35]
constexpr size_t DY_BLOB_FLOATS = DY_OFF_TIME + 256 * 21 + 256 + 35 * 256 + 36;
// channel (0..34, or -1) of the per-view feature held by register q of a lane of half h
__host__ __device__ constexpr int da_c35(int q, int h) { return h == 0 ? q : (q < 17 ? 18 + q : -1); }

// channel (0..69, or -1) of the 70-wide per-view feature held by register q of a lane of half h
__host__ __device__ constexpr int sa_c70(int q, int h) {
  if (q < 18) return h == 0 ? q : (q < 17 ? 18 + q : -1);
  if (q < 34) return 35 + dyn_fi(q - 18, h);
  return h == 0 ? 35 + 32 + (q - 34) : -1;
}

// reference column of ray_dir_fc.0's input for k-step s, half h  (-1: unused, -2: bias): the per-point part ...
static int sa_l1p_col(int s, int h) {
  if (s < 15) {
    const int c = s / 5, fi = s % 5;  // coordinate of pts, frequency index
    return 3 + (h * 5 + fi) * 3 + c;
  }
  const int k = (s - 15) * 2 + h;  // raw list: pts(3), -
  return k < 3 ? k : -1;
}
// ... and the per-view part
static int sa_l1v_col(int s, int h) {
  if (s < 30) {
    const int c = s / 5, fi = s % 5;  // Pluecker coordinate 0..5, frequency index
    return 33 + 6 + (h * 5 + fi) * 6 + c;
  }
  const int k = (s - 30) * 2 + h;  // raw list: Pluecker(6), ray_diff(4), ONE, -
  if (k < 6) return 33 + k;
  if (k < 10) return 99 + (k - 6);
  return k == 10 ? -2 : -1;
}

extern "C" size_t dyn_static_net_blob_floats(void) { return ST_BLOB_FLOATS; }

extern "C" int dyn_static_net_pack(const float* const* T, int F, float* blob, size_t blob_floats) {
  DYN_REQUIRE(T && blob, "dyn_static_net_pack: null pointer");
  DYN_REQUIRE(F == 32, "dyn_static_net_pack: the kernels are specialised for 32 feature channels (coarse_feat_dim = fine_feat_dim = 32)");
  DYN_REQUIRE(blob_floats >= ST_BLOB_FLOATS, "dyn_static_net_pack: blob too small");
  for (int i = 0; i < ST_NUM_TENSORS; ++i) DYN_REQUIRE(T[i] != nullptr, "dyn_static_net_pack: tensor %d is NULL", i);
  std::vector<float> o;
  g_pack_range_error = false;
  o.reserve(ST_BLOB_FLOATS);
  // ---- A ----
  {
    const float *W = T[ST_RAYDIR0_W], *b = T[ST_RAYDIR0_B];
    // (ELU_PRE / ELU_POST: the layers whose outputs only feed ELUs hold log2(e) times their value, their consumers ln(2) -- elu_s, dyn_mlp.h)
    pack_net_layer(o, 8, SA_L1P_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i, c = sa_l1p_col(s, h);
      return c >= 0 ? W[n * 103 + c] : 0.f;
    }, DYN_ELU_PRE));
    pack_net_layer(o, 8, SA_L1V_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i, c = sa_l1v_col(s, h);
      return c >= 0 ? W[n * 103 + c] : (c == -2 ? b[n] : 0.f);
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 2, SA_L2_STEPS, scaled(chained(T[ST_RAYDIR2_W], nullptr, 35, 256, 256), DYN_ELU_POST));
  {
    // base_fc.0 (mlp_network.py:477-481, input [mean | var | x]): the [mean | var] columns act on per-point statistics and are
    // evaluated once per point for the whole workgroup, the x columns (+ bias) per view
    const float *W = T[ST_BASE0_W], *b = T[ST_BASE0_B];
    pack_net_layer(o, 8, SA_L3P_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int c = sa_c70(s % SA_NX, h);
      return c < 0 ? 0.f : W[(32 * t + i) * 210 + (s / SA_NX) * 70 + c];
    }, DYN_ELU_PRE));
    pack_net_layer(o, 8, SA_L3V_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s == SA_NX) return h == 0 ? b[n] : 0.f;
      const int c = sa_c70(s, h);
      return c < 0 ? 0.f : W[n * 210 + 140 + c];
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 4, SA_L4_STEPS, scaled(chained(T[ST_BASE2_W], nullptr, 128, 256, 256), DYN_ELU_POST));
  pack_net_layer(o, 4, SA_L5_STEPS, scaled(chained(T[ST_VIS0_W], nullptr, 128, 128, 128), DYN_ELU_PRE));
  pack_net_layer(o, 4, SA_L5_STEPS, scaled(chained(T[ST_VIS2_W], nullptr, 128, 128, 128), DYN_ELU_POST));  // rows 0..127 = x_res
  pack_net_layer(o, 4, SA_L5_STEPS, scaled(chained(T[ST_VISB0_W], nullptr, 128, 128, 128), DYN_ELU_PRE));
  DYN_REQUIRE(o.size() == ST_OFF_B, "static pack: A stream size mismatch");
  // ---- B ----
#if DYN_POINTS_DUO
  g_pack_chunk_pairs = PTS_CP;
#endif
  {
    const float *W = T[ST_GEO0_W], *b = T[ST_GEO0_B];
    pack_net_layer(o, 8, 129, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < 128) return W[n * 257 + (s < 64 ? 0 : 128) + chain_feature(s % 64, h)];  // mean | var
      return h == 0 ? W[n * 257 + 256] : b[n];                                          // mean of the weights | bias
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 4, 129, scaled_wb(chained(T[ST_GEO2_W], T[ST_GEO2_B], 128, 256, 256), 128, DYN_ELU_POST, 1.0));
  pack_net_layer(o, 4, 64, chained(T[ST_WQ], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 64, chained(T[ST_WK], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 64, chained(T[ST_WV], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 64, chained(T[ST_FC], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 65, scaled(chained(T[ST_OG0_W], T[ST_OG0_B], 128, 128, 128), DYN_ELU_PRE));
  pack_net_layer(o, 4, 65, scaled(chained(T[ST_RGB0_W], T[ST_RGB0_B], 128, 128, 261), DYN_ELU_PRE));  // columns 0..127 = globalfeat part
  g_pack_chunk_pairs = B6_CHUNK_PAIRS;
  DYN_REQUIRE(o.size() == ST_OFF_C, "static pack: B stream size mismatch");
  // ---- C ----
  {
    const float* W = T[ST_RGB0_W];
    pack_net_layer(o, 4, SC_L11_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < 64) return W[n * 261 + 128 + chain_feature(s, h)];
      const int k = (s - 64) * 2 + h;  // vis, ray_diff[0..3]
      return k < 5 ? W[n * 261 + 256 + k] : 0.f;
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 2, 64, scaled(chained(T[ST_RGB2_W], nullptr, 64, 128, 128), DYN_ELU_POST * DYN_ELU_PRE));
  DYN_REQUIRE(o.size() == ST_OFF_CTA, "static pack: C stream size mismatch");
  // ---- constant tables ----
  pack_rowtab(o, T[ST_VIS2_W] + 128 * 128, 128, DYN_ELU_POST);
  pack_rowtab(o, T[ST_VISB2_W], 128, DYN_ELU_POST);
  o.push_back(T[ST_VIS2_B][128]);
  o.push_back(T[ST_VISB2_B][0]);
  o.push_back(fabsf(T[ST_S][0]));
  o.resize(ST_OFF_CTA + 272, 0.f);
  pack_rowtab(o, T[ST_BASE2_B], 128);
  pack_rowtab(o, T[ST_VIS0_B], 128, DYN_ELU_PRE);
  pack_rowtab(o, T[ST_VIS2_B], 128);
  pack_rowtab(o, T[ST_VISB0_B], 128, DYN_ELU_PRE);
  {
    float b64[64] = {0.f};
    for (int i = 0; i < 35; ++i) b64[i] = T[ST_RAYDIR2_B][i];
    pack_rowtab(o, b64, 64);
  }
  o.resize(ST_OFF_CTB, 0.f);
  pack_rowtab(o, T[ST_LN_G], 128);
  pack_rowtab(o, T[ST_LN_B], 128);
  pack_rowtab(o, T[ST_OG2_W], 128, DYN_ELU_POST);
  o.push_back(T[ST_OG2_B][0]);
  o.resize(ST_OFF_CTC, 0.f);
  pack_rowtab(o, T[ST_RGB4_W], 64, DYN_ELU_POST);
  o.push_back(T[ST_RGB4_B][0]);
  o.resize(ST_OFF_CTC + 80, 0.f);
  pack_rowtab(o, T[ST_RGB2_B], 64, DYN_ELU_PRE);
  o.resize(ST_OFF_REF, 0.f);
  for (int i = 0; i < 35 * 66; ++i) o.push_back(T[ST_REFFEAT_W][i]);
  for (int i = 0; i < 35; ++i) o.push_back(T[ST_REFFEAT_B][i]);
  o.resize(ST_BLOB_FLOATS, 0.f);
  DYN_REQUIRE(!g_pack_range_error, "dyn_static_net_pack: a weight is outside the half-float range of the split engine (|w| >= 65504 or not finite)");
  for (size_t i = 0; i < ST_BLOB_FLOATS; ++i) blob[i] = o[i];
  return 0;
}

// -------------------------------------------------------------------------------------------------------------------
// workspace layout (floats) of one dyn_static_net call
// -------------------------------------------------------------------------------------------------------------------
static bool dense_views(int V) {
  static const int mode = getenv("DYN_DENSE_VIEWS") ? atoi(getenv("DYN_DENSE_VIEWS")) : -1;  // developer A/B: 0 never, 1 whenever V >= 9
  if (mode == 0) return false;
  if (mode == 1) return V >= 9;
  // measured (4096/2048 rays x 64 samples, k_static_views + k_static_blend): dense rows win where the lane segments would be more than a
  // quarter padding -- 9..12 views (vs 16 lanes) and 17..26 views (vs 32 lanes); 13..15 and 27..31 views keep the segments
  return (V >= 9 && V <= 12) || (V >= 17 && V <= 26);
}
struct StaticWs {
  bool dense, ragged;
  long n_pts, n_tiles_a, n_tiles_b;
  int PT, TPR;  // points per A tile; B tiles per ray (1, 2, 4 or -- rays of more than 128 samples -- a multiple of 4)
  size_t off_x, off_vis, off_gin, off_nvalid, off_hg, off_ref, total;  // dynamic net: off_x/off_vis/off_hg unused, off_ref = time feature
  size_t off_qkvg;  // long rays only: per B tile [g | q | k | v][4][4][64 lanes][4] (the two-pass point chain hands these over)
  // ragged dense rows (round 6): the plan of a launch -- see k_ragged_points
  long n_seg, n_wg_max;
  size_t off_bits, off_emin, off_segcnt, off_segstart, off_wgstart;  // [n_pts] u32, [n_pts] f32, [n_seg] i32, [n_seg][RAG_SEG_WGS] i32, [1 + n_wg_max + 1] i32 (n_wg first)
  size_t off_rowtab, off_ptab;  // per workgroup: [256] u16 (point | view << 8 of every row) and [RAG_PTAB] i32 (row offsets of its points, then n_rows, first point, points)
};
// Ragged dense rows (round 6).  A (point, view) row whose mask is 0 contributes exactly nothing to any output of the reference: its pooling weight, its
// visibility and its blending weight are products with the mask (mlp_network.py:463-471, 484-488) and its logit is filled with -1e9 before the softmax
// (:523-525), i.e. exp(-1e9 - max) = 0 in fp32.  The reference and rounds 1-5 evaluate such rows anyway: 11-14 % of the rows of the synthetic scenes.  In the
// dense-rows flavour (V = 9..12, 17..26: the 11 static views of the Nvidia evaluation) a workgroup's 256 rows are now the VALID rows of consecutive points --
// as many points as fit, at most 32 (the per-point tiles of base_fc.0 / ray_dir_fc.0 are one 32-column MFMA tile) -- so the matrix work, the parked x and the
// blend's stream shrink by the masked fraction.  A point without any valid view keeps ONE row (its view 0, mask 0): every point then owns a column and a
// record, and what the reference does for it -- sigma -1e9, colour = the plain mean of the V gathered colours, the uniform softmax over equal logits --
// is reproduced by the blend from rgb_feat.  The anti-alias pooling weight needs min_v exp(|s| (d_v - 1)) over ALL views, masked ones included
// (mlp_network.py:465-468): the plan carries it per point.
// The plan (four small launches per network call): k_ragged_points: per point the valid-view bit mask and that minimum; k_ragged_segments: one thread per
// segment of RAG_SEG consecutive points packs its points greedily into workgroups (rows <= 256, points <= 32); k_ragged_scatter: exclusive scan of the
// segments' workgroup counts and the flat list wg_start[0 .. n_wg]; k_ragged_tables: per workgroup the row table (row -> point, view) and the row offsets
// of its points, so that a consumer workgroup starts with ONE round trip to memory (its own table) in front of its input loads.  The launch grid is the
// upper bound n_wg_max; surplus workgroups leave at once.
#define RAG_SEG 1024      /* points per planning segment (its last workgroup is partly filled: ~1.2 % of the rows at 11 views) */
#define RAG_SEG_WGS 160   /* workgroups a segment can need: ceil(1024 / 7) (26 views: >= 9 points fit 256 rows; short of 32 points only by the row limit) + slack */
#define RAG_MAX_PTS 32
#define RAG_PTAB 36       /* ints per workgroup: pbase[0..32], then first point, number of points, 0 */

#define DYN_MAX_SAMPLES 256  // samples per ray the point chain is built for (sinusoid table, LDS key blocks of 128)
// Per-point records handed between the view kernels and the point kernel are lane-coalesced for the point kernel: record i (a float4)
// of the point at (ray, sample) sits at [tile = ray * TPR + sample / 32][i][lane = sample % 32 + 32 * half].  (A point-major layout
// made every wave of the point kernel pull 64 distinct cache lines per load instruction and thrash its L1: 42 k cycles of prologue.)
#define SB_GIN_RECS 33  // geometry_fc input per (point, half): 16 x mean, 16 x var, then [mean weight | 1], 0, 0, 0
#define SB_HG_RECS 16   // point part of rgb_fc.0 per (point, half)

#ifndef DYN_BLEND_WS
#define DYN_BLEND_WS (DYN_SPLIT_TERMS == 3 ? 1 : 0)
#endif
static bool ragged_rows_enabled() {
  static const int off = getenv("DYN_RAGGED") ? (atoi(getenv("DYN_RAGGED")) == 0) : 0;  // developer A/B: DYN_RAGGED=0 evaluates every row (rounds 1-5)
  static const int blend_stream = getenv("DYN_BLEND_STREAM") != nullptr;                  // (the streaming blend of round 3 has no ragged form)
  return DYN_BLEND_WS && !off && !blend_stream;  // (the 6-term build keeps the streaming blend: no ragged rows there)
}
static StaticWs static_ws(int R, int S, int V, bool dynamic = false) {
  StaticWs w;
  w.n_pts = (long)R * S;
  // points per wave: views are padded to a power-of-two lane segment; or (dense rows: view counts from 9 that are not a power of two)
  // points per 256-row workgroup, whose 8 waves are 8 tiles of the parked-feature buffer
  w.dense = dense_views(V);
  w.ragged = w.dense && ragged_rows_enabled();
  w.PT = w.dense ? 256 / V : 32 / (V <= 4 ? 4 : (V <= 8 ? 8 : (V <= 16 ? 16 : 32)));
  w.n_tiles_a = w.dense ? ((w.n_pts + w.PT - 1) / w.PT) * 8 : (w.n_pts + w.PT - 1) / w.PT;
  w.n_seg = (w.n_pts + RAG_SEG - 1) / RAG_SEG;
  // a ragged workgroup holds at least 256 / V points unless its segment ends: never more workgroups than the regular partition + one per segment
  w.n_wg_max = w.ragged ? (w.n_pts + w.PT - 1) / w.PT + w.n_seg : 0;
  if (w.ragged) w.n_tiles_a = w.n_wg_max * 8;
  int tpr = (S + 31) / 32;
  w.TPR = tpr <= 1 ? 1 : (tpr <= 2 ? 2 : ((tpr + 3) / 4) * 4);
  w.n_tiles_b = (long)R * w.TPR;
  size_t o = 0;
  w.off_x = o; o += dynamic ? 0 : (size_t)w.n_tiles_a * 64 * 64;
  w.off_vis = o; o += dynamic ? 0 : (size_t)w.n_tiles_a * 128;  // per tile [32 rows] float4 {vis2, r, g, b}: what the blend needs of a row besides x
  w.off_gin = o; o += (size_t)w.n_tiles_b * SB_GIN_RECS * 256;
  w.off_nvalid = o; o += (size_t)((w.n_pts + 3) & ~3L);
  w.off_hg = o; o += dynamic ? 0 : (size_t)w.n_tiles_b * SB_HG_RECS * 256;
  w.off_ref = o; o += dynamic ? 64 : (size_t)R * 36;
  o = (o + 3) & ~(size_t)3;
  w.off_qkvg = o; o += w.TPR > 4 ? (size_t)w.n_tiles_b * 4 * 4096 : 0;
  w.off_bits = o; o += w.ragged ? (size_t)((w.n_pts + 3) & ~3L) : 0;
  w.off_emin = o; o += w.ragged ? (size_t)((w.n_pts + 3) & ~3L) : 0;
  w.off_segcnt = o; o += w.ragged ? (size_t)((w.n_seg + 3) & ~3L) : 0;
  w.off_segstart = o; o += w.ragged ? (size_t)w.n_seg * RAG_SEG_WGS : 0;
  w.off_wgstart = o; o += w.ragged ? (size_t)((w.n_wg_max + 2 + 3) & ~3L) : 0;
  w.off_rowtab = o; o += w.ragged ? (size_t)w.n_wg_max * 128 : 0;
  w.off_ptab = o; o += w.ragged ? (size_t)w.n_wg_max * RAG_PTAB : 0;
  w.total = o;
  return w;
}

extern "C" size_t dyn_static_net_workspace_bytes(int R, int S, int V) {
  if (R <= 0 || S <= 0 || V <= 0 || V > 32) return 0;
  return static_ws(R, S, V).total * sizeof(float);
}

// -------------------------------------------------------------------------------------------------------------------
// Fourier features, Pluecker coordinates
// -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unit3(float x, float y, float z, float& ox, float& oy, float& oz) {
  const float d = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);  // F.normalize(eps=1e-12)
  ox = x / d; oy = y / d; oz = z / d;
}

// cos | sin of 2^f x for f = 0..NF-1 (the reference's octave PeriodicEmbed, mlp_network.py:530-555): one accurate sincosf and
// NF-1 double-angle steps.  2^f x is exact in fp32, so the only difference from evaluating each octave directly is the recurrence's
// round-off (about 2^f ulp, <= 2e-6 at the fifth octave), far inside the 1e-4 budget, for a fifth of the transcendental work.
// sin and cos of a scene coordinate (|x| up to a few hundred): two-step Cody-Waite reduction by pi/2 with fma (the products are
// exact, the residual error is |y| * 1e-15) and the cephes single-precision minimax polynomials on [-pi/4, pi/4] (< 2 ulp).
// A third of the instructions of the general-range library sincosf, which cost 4.5 % of the view kernel (9 calls per row).
__device__ __forceinline__ void sincos_small(float x, float& s, float& c) {
  const float y = rintf(x * 0.63661977236758134f);
  const int q = (int)y;
  float r = fmaf(y, -1.57079637050628662109375f, x);   // float(pi/2)
  r = fmaf(y, 4.37113900018624283e-8f, r);              // float(pi/2) - pi/2
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(z, -0.5f, 1.0f));
  const bool swap = q & 1;
  const float sv = swap ? pc : ps, cv = swap ? ps : pc;
  // quadrant signs: sin flips in quadrants 2, 3; cos flips in quadrants 1, 2
  s = __uint_as_float(__float_as_uint(sv) ^ ((unsigned)(q & 2) << 30));
  c = __uint_as_float(__float_as_uint(cv) ^ ((unsigned)((q + 1) & 2) << 30));
}

template <int NF>
__device__ __forceinline__ void octave_embed(float x, int h, float* out) {
  float sn, cs;
  sincos_small(x, sn, cs);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    out[f] = h == 0 ? cs : sn;
    const float s2 = 2.0f * sn * cs;
    cs = cs * cs - sn * sn;
    sn = s2;
  }
}

// ref_feature_fc.0(PE(ref Pluecker)) per ray (mlp_network.py:434,456; render_ray.py:372-377): [R,36]
// One block = REF_RAYS rays: the 66 embedded Pluecker values of a ray are formed ONCE (rounds 1-5: by each of the ray's 35 output threads again) and parked in LDS, then one
// thread per (ray, channel) adds its 66 products in the reference's order -- the same cosf / sinf values, the same sums, a thirty-fifth of the transcendental work.
// (One block per CU and all 512 registers of a lane: no other kernel's waves beside it -- csrc/dyn_mlp.h, DYN_EXCLUSIVE_CU.)
#define REF_RAYS 8
__global__ void __launch_bounds__(256, 1) k_static_ref_feat(const float* __restrict__ ray_o, const float* __restrict__ ray_d, const float* __restrict__ Wref, int R,
                                  float* __restrict__ ref_feat) {
  DYN_CLAIM_REGISTER_FILE();
  float* pe = reinterpret_cast<float*>(dyn_smem);  // [REF_RAYS][66]: the 6 raw coordinates, then cos(2^f c_k) for f, k, then sin(2^f c_k)
  const int r0 = blockIdx.x * REF_RAYS;
  for (int t = threadIdx.x; t < REF_RAYS * 66; t += 256) {
    const int rl = t / 66, jj = t - rl * 66;
    const int r = r0 + rl < R ? r0 + rl : R - 1;
    float d[3], c[6];
    unit3(ray_d[r * 3], ray_d[r * 3 + 1], ray_d[r * 3 + 2], d[0], d[1], d[2]);
    const float ox = ray_o[r * 3], oy = ray_o[r * 3 + 1], oz = ray_o[r * 3 + 2];
    c[0] = d[0]; c[1] = d[1]; c[2] = d[2];
    c[3] = oy * d[2] - oz * d[1];
    c[4] = oz * d[0] - ox * d[2];
    c[5] = ox * d[1] - oy * d[0];
    if (dyn_ref_cross_axis(R) != DYN_CROSS_XYZ) {  // a chunk of exactly 3 rays: the reference's torch.cross runs over the rays (csrc/dyn_device.h)
      float m[3];
      dyn_ref_moment_over_rays(ray_o, ray_d, r, m);
      c[3] = m[0]; c[4] = m[1]; c[5] = m[2];
    }
    const int e = jj < 6 ? jj : (jj - 6) % 6;  // the coordinate
    float ck = c[0];
    ck = e == 1 ? c[1] : ck; ck = e == 2 ? c[2] : ck; ck = e == 3 ? c[3] : ck; ck = e == 4 ? c[4] : ck; ck = e == 5 ? c[5] : ck;
    float v = ck;
    if (jj >= 6) {
      const int f = ((jj - 6) / 6) % 5, fn = (jj - 6) / 30;
      const float a = (float)(1 << f) * ck;
      v = fn == 0 ? cosf(a) : sinf(a);
    }
    pe[t] = v;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < REF_RAYS * 36; t += 256) {
    const int rl = t / 36, ch = t - rl * 36;
    if (r0 + rl >= R) continue;
    float acc = 0.f;
    if (ch < 35) {
      const float* w = Wref + ch * 66;
      const float* x = pe + rl * 66;
      acc = Wref[35 * 66 + ch];
      for (int j = 0; j < 66; ++j) acc = fmaf(w[j], x[j], acc);  // (raw coordinates first, then cos, then sin: the order of the reference's concatenation and of rounds 1-5)
    }
    ref_feat[(long)(r0 + rl) * 36 + ch] = acc;
  }
}

struct StaticArgs {
  int R, S, V, PT, TPR;
  int anti_alias, mask_rgb;
  long n_pts, n_tiles_a, n_tiles_b;
  float shift;            // dynamic net: subtracted from sigma (mlp_network.py:295)
  const float* blob;
  const float* ray_d;     // [R,3]   (dynamic net: viewing direction features)
  const float* pts;       // [R,S,3]
  const float* rgb_feat;  // [R,S,V,35]
  const float* ray_diff;  // [R,S,V,4]
  const float* mask;      // [R,S,V]
  const float* centers;   // [V,16]: source camera centre at [12..14]
  float* raw;             // [R,S,4]
  float* ws;
  StaticWs o;
  // ragged dense rows: the plan of this launch (null / unused in the other flavours)
  const unsigned* rg_bits;         // [n_pts] bit v set: mask[point, v] != 0
  const float* rg_emin;            // [n_pts] min over all V views of exp(|s| (ray_diff.w - 1)) (anti-alias pooling only)
  const int* rg_wg;                // [0]: n_wg, [1 .. n_wg + 1]: first point of every workgroup
  const unsigned short* rg_rowtab; // [n_wg][256]
  const int* rg_ptab;              // [n_wg][RAG_PTAB]
};
// float4 index of record 0 of (point, half) in a [tile][n_rec][64 lanes] buffer (see SB_GIN_RECS); record i is i * 64 further
__device__ __forceinline__ long point_rec(const StaticArgs& p, long point, int h, int n_rec) {
  const long ray = point / p.S;
  const int smp = (int)(point - ray * p.S);
  return ((ray * p.TPR + (smp >> 5)) * n_rec) * 64 + (smp & 31) + 32 * h;
}

// -------------------------------------------------------------------------------------------------------------------
// base_fc.0 of the view chains.  Its input is [mean | var | x]: the weighted mean / variance over the views are per-POINT values,
// so their 4 NX columns of the weight matrix are applied once per point instead of once per view: the 8 waves pool their points'
// statistics in LDS (32 points at 8 views), wave w evaluates output tile w for all pooled points, the result comes back through LDS
// as the initial value of every view's accumulators, and only the NX + 1 per-view slots remain per wave.
// (V <= 4, 64 pooled points, does not fit the LDS budget and keeps the statistics slots per view.)
// LDS: pool [2 NX slots][2 halves][32 points], res [256 features][32 points].
// -------------------------------------------------------------------------------------------------------------------
#define POOL_FLOATS(NX) (2 * (NX) * 2 * 32)
// res: the per-point tiles on their way back to the rows, POINT-major [32 points][256 features + 4] (round 5; rounds 1-4: feature-major, 16 + 128 four-byte
// LDS accesses per lane and exchange): a lane's four consecutive registers are four consecutive features, so a tile leaves in 4 and comes back in 32
// sixteen-byte accesses; the stride of 260 floats puts the 16 lanes of a quarter-wave 4 banks apart on the way in, and on the way out the lanes of a
// point read one address (a broadcast).
#define RES_STRIDE 260
#define RES_FLOATS (RES_STRIDE * 32)
__device__ __forceinline__ void res_put(float* res, int wave, int j, int h, const f32x16& acc) {
  float4* d = reinterpret_cast<float4*>(res + j * RES_STRIDE + wave * 32 + 4 * h);
#pragma unroll
  for (int q = 0; q < 4; ++q) d[2 * q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}
__device__ __forceinline__ void res_get(const float* res, int col, int h, f32x16 (&a1)[8]) {
  const float4* s4 = reinterpret_cast<const float4*>(res + col * RES_STRIDE + 4 * h);
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = s4[t * 8 + 2 * q];
      a1[t][4 * q] = v.x; a1[t][4 * q + 1] = v.y; a1[t][4 * q + 2] = v.z; a1[t][4 * q + 3] = v.w;
    }
}

// ---- dense rows (VSEG == 0): any number of views without padding ------------------------------------------------------------------
// The lane segments above want the views of a point in a power-of-two group of lanes, so 11 views occupy 16 lanes and 31 % of the
// matrix work is padding.  In the dense flavour the 256 rows of a view workgroup are simply the first PTW * V point-views of its
// PTW = 256 / V points, in order: a point's rows sit at arbitrary lanes and may straddle two waves.  Cross-view reductions then go
// through LDS: rows deposit their terms in a [slot][row] table, a barrier, and (point, slot) tasks spread over the 512 threads add up the
// V rows of their point.  More barriers than the DPP butterflies (about 18 per pass of the view chain), none of the padding.
#define DENSE_STRIDE 258   /* row stride of a slot in the reduction tables (256 rows + 2: the two halves of a wave land on different banks) */
#define DENSE_SCALARS 512  /* floats at the end of the workgroup's LDS for scalar rounds: [2][256] */
#define DENSE_EXTRA 2304   /* floats added to the view kernels' LDS in the dense flavour (table space behind `res`) */
struct DenseRows {
  int V, PTW, rw, p_local, view, base;  // base: first row of this row's point (clamped for the idle tail rows)
  int cnt;                              // rows of this row's point: V, or -- ragged -- its valid views (>= 1)
  float* scal;                          // [2][256]
  const int* pbase;                     // ragged: [PTW + 1] first row of every point of the workgroup (LDS); regular: null (point p starts at row p V)
  long point0;                          // first point of the workgroup
  int n_rows;                           // rows in use (regular: PTW V)
};
__device__ __forceinline__ DenseRows dense_rows(int V, int PTW, float* scal, long wgi = -1) {
  DenseRows d;
  d.V = V; d.PTW = PTW; d.scal = scal;
  d.rw = (threadIdx.x >> 6) * 32 + (threadIdx.x & 31);
  d.p_local = d.rw / V;
  d.view = d.rw - d.p_local * V;
  d.base = (d.p_local < PTW ? d.p_local : PTW - 1) * V;
  d.cnt = V;
  d.pbase = nullptr;
  d.point0 = (wgi >= 0 ? wgi : (long)blockIdx.x) * PTW;
  d.n_rows = PTW * V;
  return d;
}
// first row / number of rows of point `pnt` of the workgroup (the (point, slot) task loops of the LDS reductions)
__device__ __forceinline__ int dense_pt_base(const DenseRows& d, int pnt) { return d.pbase != nullptr ? d.pbase[pnt] : pnt * d.V; }
__device__ __forceinline__ int dense_pt_cnt(const DenseRows& d, int pnt) { return d.pbase != nullptr ? d.pbase[pnt + 1] - d.pbase[pnt] : d.V; }

// Ragged rows of workgroup (or persistent unit) `wgi` from the plan's tables: row -> (point of the workgroup, view), the row offsets of its points.
// No barrier inside and ONE round trip to memory: every thread reads its own row's entry, the workgroup's scalars (rows, first point, points) are uniform
// loads, and the row offsets of the thread's own point follow from its entry -- they are first needed by the cross-view reductions, long after the input loads.
// `pbase` (LDS, RAG_PTAB ints, or null for a kernel without (point, slot) task loops): the copy the task loops read; the caller has a workgroup barrier between
// this call and the first such loop (k_static_views: the exchange of the per-point ray_dir_fc.0 tile; k_dynamic_views: the first barrier of the pooled statistics).
// `pre`: the row's table entry and the unit's scalars if the caller has loaded them already (the persistent blend requests the next unit's one unit ahead).
struct RaggedHead {
  unsigned short ri;
  int n_rows, point0, np;
};
__device__ __forceinline__ RaggedHead ragged_head(const unsigned short* rowtab, const int* ptab, long wgi, int rw) {
  RaggedHead h;
  h.ri = rowtab[wgi * 256 + rw];
  h.n_rows = ptab[wgi * RAG_PTAB + 32];
  h.point0 = ptab[wgi * RAG_PTAB + 33];
  h.np = ptab[wgi * RAG_PTAB + 34];
  return h;
}
__device__ __forceinline__ DenseRows ragged_rows(int V, const unsigned short* rowtab, const int* ptab, long wgi, float* scal, int* pbase,
                                                 const RaggedHead* pre = nullptr) {
  const int tid = threadIdx.x;
  const int* pt = ptab + wgi * RAG_PTAB;
  DenseRows d;
  d.rw = (tid >> 6) * 32 + (tid & 31);
  const RaggedHead hd = pre != nullptr ? *pre : ragged_head(rowtab, ptab, wgi, d.rw);
  const unsigned short ri = hd.ri;
  if (pbase != nullptr && tid < RAG_PTAB) pbase[tid] = pt[tid];
  d.V = V; d.scal = scal; d.pbase = pbase != nullptr ? pbase : pt;  // (non-null marks the ragged flavour; without an LDS copy the offsets are read where they lie)
  d.n_rows = hd.n_rows;
  d.point0 = hd.point0;
  d.PTW = hd.np;
  const bool live = d.rw < d.n_rows;
  d.p_local = live ? (ri & 0xff) : RAG_MAX_PTS + 1;  // idle rows: beyond every point (they shadow the last point's rows like the regular flavour's tail rows)
  d.view = live ? (ri >> 8) : 0;
  const int pc = live ? d.p_local : d.PTW - 1;
  d.base = pt[pc];
  d.cnt = pt[pc + 1] - d.base;
  return d;
}

// ---- the plan of a ragged launch ----
// e - 1 of the anti-alias pooling weight (mlp_network.py:463-466); one function so that the plan's minimum over all views and the rows' own values are the same bits.
// The reference forms (e_v - min_v e) with e = exp(|s| (dot - 1)) ~ 1: a difference of nearly equal numbers whose fp32 roundings (6e-8 each, more with a 1-ulp exp) are
// all of its error -- 1e-3 relative on weights of 1e-5 apart, and what tests/parity.check_static_net's conditioning allowance is for.  expm1(a_v) - min_v expm1(a) is the
// same difference without the rounding of e: against the float64 values of the reference's formula the density logits of a trained-scale net moved from 3 x the fp32
// reference's own error to below it (tools/fp64_probe.py).
__device__ __forceinline__ float aa_exp(float s_abs, float dot) { return expm1f(s_abs * (dot - 1.0f)); }

// one block = 256 consecutive points: their V masks and V ray_diff records are contiguous runs, read coalesced (16 bytes per lane for the records) and parked in LDS,
// then one thread per point walks its V entries there (a thread per point reading its own 44-byte / 176-byte strided runs took ~100 us per call)
__global__ void __launch_bounds__(256) k_ragged_points(long n_pts, int V, const float* __restrict__ mask, const float4* __restrict__ ray_diff, const float* __restrict__ s_abs,
                                                       unsigned* __restrict__ bits, float* __restrict__ emin) {
  float* dw = reinterpret_cast<float*>(dyn_smem);                      // [256 V] ray_diff.w (anti-alias pooling only)
  unsigned char* mk = reinterpret_cast<unsigned char*>(dw + 256 * V);  // [256 V] mask != 0
  const long p0 = (long)blockIdx.x * 256;
  const int np = (int)(n_pts - p0 < 256 ? n_pts - p0 : 256), n = np * V;
  const bool aa = s_abs != nullptr;
  for (int i = threadIdx.x; i < n; i += 256) {
    mk[i] = mask[p0 * V + i] != 0.f ? 1 : 0;
    if (aa) dw[i] = ray_diff[p0 * V + i].w;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t >= np) return;
  unsigned b = 0u;
  float m = 3.0e38f;
  const float sa = aa ? s_abs[0] : 0.f;
  for (int v = 0; v < V; ++v) {
    if (mk[t * V + v]) b |= 1u << v;
    if (aa) m = fminf(m, aa_exp(sa, dw[t * V + v]));
  }
  bits[p0 + t] = b;
  if (aa) emin[p0 + t] = m;
}

// one block per segment: greedy packing of the segment's points into workgroups (rows <= 256, points <= RAG_MAX_PTS).  The block reads the segment's bit masks
// coalesced, forms the prefix sums of the row counts in LDS, every thread finds by bisection where a workgroup STARTING at its points would end (the greedy rule
// depends on the start only), and one thread follows that chain from point 0: ~40 dependent LDS reads instead of a 1024-step walk (a thread per segment reading
// global memory point by point took 0.3 ms -- 14 ms per frame, most of what ragged rows had gained; the walk over LDS bytes 47-63 us)
__global__ void __launch_bounds__(256) k_ragged_segments(long n_pts, long n_seg, const unsigned* __restrict__ bits, int* __restrict__ seg_cnt, int* __restrict__ seg_start) {
  int* pre = reinterpret_cast<int*>(dyn_smem);                           // [RAG_SEG + 1] exclusive prefix sums of the row counts
  int* part = pre + RAG_SEG + 4;                                         // [256] per-thread sums / their scan
  unsigned short* nxt = reinterpret_cast<unsigned short*>(part + 256);   // [RAG_SEG] first point of the workgroup behind the one that starts here
  const int tid = threadIdx.x;
  const long seg = blockIdx.x;
  const long p0 = seg * RAG_SEG;
  const int n = (int)((p0 + RAG_SEG < n_pts ? p0 + RAG_SEG : n_pts) - p0);
  int r4[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // thread t owns points 4 t .. 4 t + 3
    const int i = 4 * tid + j;
    const unsigned b = i < n ? bits[p0 + i] : 0u;
    r4[j] = i < n ? (b != 0u ? __popc(b) : 1) : 0;
    sum += r4[j];
  }
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan of the 256 partial sums
    const int t = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  int run = part[tid] - sum;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pre[4 * tid + j] = run;
    run += r4[j];
  }
  if (tid == 255) pre[RAG_SEG] = run;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = 4 * tid + j;
    if (i < n) {
      // the largest e in (i, min(i + RAG_MAX_PTS, n)] with pre[e] - pre[i] <= 256 (e = i + 1 always qualifies: a point has at most 32 rows)
      int lo = i + 1, hi = i + RAG_MAX_PTS < n ? i + RAG_MAX_PTS : n;
      const int base = pre[i];
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] - base <= 256) lo = mid; else hi = mid - 1;
      }
      nxt[i] = (unsigned short)lo;
    }
  }
  __syncthreads();
  if (tid != 0) return;
  int* out = seg_start + seg * RAG_SEG_WGS;
  int n_wg = 0;
  for (int i = 0; i < n; i = nxt[i]) out[n_wg++] = i;
  seg_cnt[seg] = n_wg;
}

// one block: exclusive scan of the segments' workgroup counts, then wg[0] = n_wg, wg[1 + i] = first point of workgroup i, wg[1 + n_wg] = n_pts
__global__ void __launch_bounds__(1024) k_ragged_scatter(long n_pts, long n_seg, const int* __restrict__ seg_cnt, const int* __restrict__ seg_start, int* __restrict__ wg) {
  int* part = reinterpret_cast<int*>(dyn_smem);  // [1024] + the running total behind it
  int& carry = part[1024];
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (long s0 = 0; s0 < n_seg; s0 += 1024) {
    const long seg = s0 + tid;
    const int n = seg < n_seg ? seg_cnt[seg] : 0;
    part[tid] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      const int t = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += t;
      __syncthreads();
    }
    const int base = carry + part[tid] - n;
    if (seg < n_seg)
      for (int i = 0; i < n; ++i) wg[1 + base + i] = (int)(seg * RAG_SEG) + seg_start[seg * RAG_SEG_WGS + i];
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  if (tid == 0) {
    wg[0] = carry;
    wg[1 + carry] = (int)n_pts;
  }
}

// one wave per workgroup of the plan: its row table and the row offsets of its points
__global__ void __launch_bounds__(64) k_ragged_tables(const unsigned* __restrict__ bits_g, const int* __restrict__ wg, unsigned short* __restrict__ rowtab, int* __restrict__ ptab) {
  const long wgi = blockIdx.x;
  if (wgi >= wg[0]) return;
  const int tid = threadIdx.x, i = tid & 31;
  const int p0 = wg[1 + wgi], np = wg[2 + wgi] - p0;
  const bool own = tid < 32 && i < np;  // lanes 0..31 own the points, the upper half idles along (the shuffles want the whole wave)
  const unsigned bits = own ? bits_g[p0 + i] : 0u;
  const int n = own ? (bits != 0u ? __popc(bits) : 1) : 0;
  int incl = n;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const int t = __shfl_up(incl, off, 32);
    if (i >= off) incl += t;
  }
  unsigned short* rt = rowtab + wgi * 256;
  int* pt = ptab + wgi * RAG_PTAB;
  const int total = __shfl(incl, 31);
  for (int r = total + tid; r < 256; r += 64) rt[r] = 0;  // idle rows
  if (tid < 32) {
    pt[i] = incl - n;
    if (i == 31) { pt[32] = incl; pt[33] = p0; pt[34] = np; pt[35] = 0; }
    int r = incl - n;
    if (own) {
      if (bits == 0u) {
        rt[r] = (unsigned short)i;  // the placeholder row of a point without valid views: its view 0 (mask 0)
      } else {
        for (unsigned b = bits; b != 0u; b &= b - 1u) rt[r++] = (unsigned short)(i | ((__ffs(b) - 1) << 8));
      }
    }
  }
}

// all-reduce of one value per row over the rows of the row's point (two values per round with the second table)
template <class Op>
__device__ __forceinline__ float dense_all(const DenseRows& d, float v, float ident, Op op) {
  if ((threadIdx.x & 32) == 0) d.scal[d.rw] = v;
  __syncthreads();
  // four loads in flight per step: the trip count is a run-time value, and one dependent LDS round trip per view would cost ~100 cycles each
  const float* src = d.scal + d.base;
  float a0 = ident, a1 = ident, a2 = ident, a3 = ident;
  int k = 0;
  // (the trip count is the point's own row count -- the view count, a compile-time constant, in the regular specialised kernels.  A fixed trip count of V with the
  //  entries beyond the point's rows predicated off was measured for the ragged flavour in round 6: every load in flight at once, but 5-6 % SLOWER view kernel)
  for (; k + 4 <= d.cnt; k += 4) { a0 = op(a0, src[k]); a1 = op(a1, src[k + 1]); a2 = op(a2, src[k + 2]); a3 = op(a3, src[k + 3]); }
  for (; k < d.cnt; ++k) a0 = op(a0, src[k]);
  __syncthreads();
  return op(op(a0, a1), op(a2, a3));
}
__device__ __forceinline__ void dense_all2(const DenseRows& d, float v0, float v1, float& s0, float& s1) {
  if ((threadIdx.x & 32) == 0) { d.scal[d.rw] = v0; d.scal[256 + d.rw] = v1; }
  __syncthreads();
  const float* src = d.scal + d.base;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  int k = 0;
  for (; k + 2 <= d.cnt; k += 2) { a0 += src[k]; a1 += src[k + 1]; b0 += src[256 + k]; b1 += src[256 + k + 1]; }
  for (; k < d.cnt; ++k) { a0 += src[k]; b0 += src[256 + k]; }
  __syncthreads();
  s0 = a0 + a1; s1 = b0 + b1;
}
template <int VSEG>
__device__ __forceinline__ float views_sum(const DenseRows& d, float v) {
  if constexpr (VSEG == 0) return dense_all(d, v, 0.f, [](float a, float b) { return a + b; });
  else return seg_sum<VSEG>(v, 0, 0);
}
template <int VSEG>
__device__ __forceinline__ float views_min(const DenseRows& d, float v) {
  if constexpr (VSEG == 0) return dense_all(d, v, 3.0e38f, [](float a, float b) { return fminf(a, b); });
  else return seg_min<VSEG>(v, 0, 0);
}
template <int VSEG>
__device__ __forceinline__ float views_max(const DenseRows& d, float v) {
  if constexpr (VSEG == 0) return dense_all(d, v, -3.0e38f, [](float a, float b) { return fmaxf(a, b); });
  else return seg_max<VSEG>(v, 0, 0);
}

// base_fc.0's per-point statistics in the dense flavour: weighted mean and variance of the NX x 2 per-view slots over the views of
// each point, into `pool` (the B operand of the per-point MFMA).  The rows deposit their slots ROW-major in 16-byte quads ([row][half]
// [quad], row stride 36 floats: the 16 lanes of a quad access start 4 banks apart, conflict-free), a (point, half, quad) task then
// walks the V rows of its point twice (mean, then sum w (x - mean)^2 -- the same two sweeps as the lane-segment flavour) with 16-byte
// loads.  The table behind `pool` holds 4 quads per half, so the ceil(NX / 4) quads go in rounds.
// row k (< pn, the point's own row count) of a point's rows in a [row][stride] table of 16-byte quads, and its weight
__device__ __forceinline__ float4 dense_quad(const float* src, int k, int stride, int) { return *reinterpret_cast<const float4*>(src + k * stride); }
__device__ __forceinline__ float dense_w(const float* w, int k, int) { return w[k]; }

template <int NX>
__device__ __forceinline__ void dense_pool_stats(const DenseRows& d, const float (&xin)[NX], float wgt, float* pool, float* tab) {
  constexpr int QPH = 4, RS = 2 * 4 * QPH + 4;
  constexpr int NQ = (NX + 3) / 4, ROUNDS = (NQ + QPH - 1) / QPH;
  const int tid = threadIdx.x, h = (tid >> 5) & 1;
  float* wrow = d.scal;  // the pooling weight of every row; published by the first barrier below
  if (h == 0) wrow[d.rw] = wgt;
  auto X = [&](int i) { return i < NX ? xin[i < NX ? i : 0] : 0.f; };
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    const int nq = (NQ - round * QPH < QPH) ? NQ - round * QPH : QPH;
    float4* mine = reinterpret_cast<float4*>(tab + d.rw * RS + h * (4 * QPH));
#pragma unroll
    for (int qq = 0; qq < QPH; ++qq) {
      const int q = round * QPH + qq;
      if (q < NQ) mine[qq] = make_float4(X(4 * q), X(4 * q + 1), X(4 * q + 2), X(4 * q + 3));
    }
    __syncthreads();
    const int total = d.PTW * 2 * nq;
    for (int task = tid; task < total; task += DYN_VIEW_THREADS) {
      const int pnt = task / (2 * nq), rem = task - pnt * (2 * nq), hh = rem / nq, qq = rem - hh * nq;
      const int pb = dense_pt_base(d, pnt), pn = dense_pt_cnt(d, pnt);
      const float* src = tab + pb * RS + hh * (4 * QPH) + qq * 4;
      const float* w = wrow + pb;
      float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
      int k = 0;
      for (; k + 2 <= pn; k += 2) {
        const float4 x0 = dense_quad(src, k, RS, pn), x1 = dense_quad(src, k + 1, RS, pn);
        const float w0 = dense_w(w, k, pn), w1 = dense_w(w, k + 1, pn);
        m0.x = fmaf(w0, x0.x, m0.x); m0.y = fmaf(w0, x0.y, m0.y); m0.z = fmaf(w0, x0.z, m0.z); m0.w = fmaf(w0, x0.w, m0.w);
        m1.x = fmaf(w1, x1.x, m1.x); m1.y = fmaf(w1, x1.y, m1.y); m1.z = fmaf(w1, x1.z, m1.z); m1.w = fmaf(w1, x1.w, m1.w);
      }
      if (k < pn) {
        const float4 x0 = dense_quad(src, k, RS, pn);
        const float w0 = dense_w(w, k, pn);
        m0.x = fmaf(w0, x0.x, m0.x); m0.y = fmaf(w0, x0.y, m0.y); m0.z = fmaf(w0, x0.z, m0.z); m0.w = fmaf(w0, x0.w, m0.w);
      }
      const float mean[4] = {m0.x + m1.x, m0.y + m1.y, m0.z + m1.z, m0.w + m1.w};
      float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
      for (k = 0; k + 2 <= pn; k += 2) {
        const float4 x0 = dense_quad(src, k, RS, pn), x1 = dense_quad(src, k + 1, RS, pn);
        const float w0 = dense_w(w, k, pn), w1 = dense_w(w, k + 1, pn);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w}, a1[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = a0[e] - mean[e], d1 = a1[e] - mean[e];
          v0[e] = fmaf(w0, d0 * d0, v0[e]);
          v1[e] = fmaf(w1, d1 * d1, v1[e]);
        }
      }
      if (k < pn) {
        const float4 x0 = dense_quad(src, k, RS, pn);
        const float w0 = dense_w(w, k, pn);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = a0[e] - mean[e];
          v0[e] = fmaf(w0, d0 * d0, v0[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = 4 * (round * QPH + qq) + e;
        if (q < NX) {
          pool[(q * 2 + hh) * 32 + pnt] = mean[e];
          pool[((NX + q) * 2 + hh) * 32 + pnt] = v0[e] + v1[e];
        }
      }
    }
    __syncthreads();
  }
}

template <int VSEG, int NX>
__device__ __forceinline__ void base_fc0(NetRing& ring, const float* pooled_w, const float (&xin)[NX], float wgt, int V, int view, int p_local,
                                         float* pool, f32x16 (&a1)[8], const DenseRows* dr = nullptr, float wsum = 0.f) {
  constexpr int PHASE_KID = 0;
  (void)PHASE_KID;
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const float one_h0 = h == 0 ? 1.0f : 0.0f;
  if constexpr (VSEG == 0) {
    // dense rows: the statistics come through the LDS tables, the rest is the pooled form below with col = the point's index in the workgroup
    float* res = pool + POOL_FLOATS(NX);
    B6TileW<2 * NX> pw;
    b6_tile_prefetch<8, 2 * NX>(pooled_w, wave, pw);
    dense_pool_stats<NX>(*dr, xin, wgt, pool, res);
    if (threadIdx.x < 2 * 2 * NX)
      for (int c = dr->PTW; c < 32; ++c) pool[threadIdx.x * 32 + c] = 0.f;
    __syncthreads();
    f32x16 accp[1];
    acc_zero(accp);
    b6_tile_apply<2 * NX>(pw, accp[0], [&](int s) { return pool[(s * 2 + h) * 32 + j]; });
    res_put(res, wave, j, h, accp[0]);
    __syncthreads();
    const int col = dr->p_local < dr->PTW ? dr->p_local : dr->PTW - 1;
    res_get(res, col, h, a1);
  } else if (VSEG >= 8) {
    static_assert(DYN_VIEW_THREADS / 64 == 8, "one output tile of base_fc.0 per wave");
    constexpr int PT = 32 / VSEG;
    float* res = pool + POOL_FLOATS(NX);
    const int col = wave * PT + p_local;
    // this wave's output tile of the per-point part: weights straight from the packed stream into registers, in flight during the statistics
    B6TileW<2 * NX> pw;
    b6_tile_prefetch<8, 2 * NX>(pooled_w, wave, pw);
    // all statistics first, one predicated block of stores after: a branch per feature would split the DPP reductions into
    // basic blocks and keep their lane moves from folding into the adds
    float pm[NX], pvr[NX];
#pragma unroll
    for (int q0 = 0; q0 + 4 <= NX; q0 += 4) {  // four features at a time (seg_sum4)
      float m4[4], v4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) m4[e] = xin[q0 + e] * wgt;
      seg_sum4<VSEG>(m4[0], m4[1], m4[2], m4[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = xin[q0 + e] - m4[e];
        v4[e] = wgt * (d * d);
      }
      seg_sum4<VSEG>(v4[0], v4[1], v4[2], v4[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { pm[q0 + e] = m4[e]; pvr[q0 + e] = v4[e]; }
    }
#pragma unroll
    for (int q = NX & ~3; q < NX; ++q) {
      pm[q] = seg_sum<VSEG>(xin[q] * wgt, V, 0);
      const float d = xin[q] - pm[q];
      pvr[q] = seg_sum<VSEG>(wgt * (d * d), V, 0);
    }
    if (view == 0) {
#pragma unroll
      for (int q = 0; q < NX; ++q) {
        pool[(q * 2 + h) * 32 + col] = pm[q];
        pool[((NX + q) * 2 + h) * 32 + col] = pvr[q];
      }
    }
    if (8 * PT < 32 && threadIdx.x < 2 * 2 * NX)  // unused point columns: keep them finite (they are computed and discarded)
      for (int c = 8 * PT; c < 32; ++c) pool[threadIdx.x * 32 + c] = 0.f;
    DYN_PHASE(6);
    __syncthreads();
    f32x16 accp[1];
    acc_zero(accp);
    b6_tile_apply<2 * NX>(pw, accp[0], [&](int s) { return pool[(s * 2 + h) * 32 + j]; });
    DYN_PHASE(7);
    res_put(res, wave, j, h, accp[0]);
    __syncthreads();
    res_get(res, col, h, a1);
    DYN_PHASE(8);
  } else {
    acc_zero(a1);
    net_layer<8, 2 * NX>(ring, a1, [&](int s) {
      const int q = s < NX ? s : s - NX;
      const float m = seg_sum<VSEG>(xin[q] * wgt, V, 0);
      if (s < NX) return m;
      const float d = xin[q] - m;
      return seg_sum<VSEG>(wgt * (d * d), V, 0);
    });
  }
  net_layer<8, NX + 1>(ring, a1, [&](int s) { return s < NX ? xin[s] : one_h0; });
  DYN_PHASE(9);
}

// -------------------------------------------------------------------------------------------------------------------
// shared tail of the per point-view chains (static: mlp_network.py:483-494, dynamic: :266-282):
// base_fc.2 -> vis_fc -> vis_fc2 -> visibility-weighted mean / variance over the views -> geometry_fc input rows
// a1: base_fc.0 output before its ELU (256 features); every ELU is applied in the feed of the layer that consumes it, so that it
// interleaves with that layer's MFMAs.  Constant table: vis row @0, vis_fc2.2 row @128, b_vis @256, b_vis2 @257.
// -------------------------------------------------------------------------------------------------------------------
template <int VSEG, bool STORE_X>
__device__ __forceinline__ void views_tail(NetRing& ring, f32x16 (&a1)[8], const StaticArgs& p, const float* ctab, float wgt, float msk,
                                           long tile, long point, bool valid, int view, int seg_base, const DenseRows* dr = nullptr,
                                           float* lds_base = nullptr, int v_const = 0) {
  constexpr int PHASE_KID = 0;
  (void)PHASE_KID;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int V = v_const > 0 ? v_const : p.V;  // (a compile-time count in the kernels instantiated for one: the reduction loops over the views unroll)
  f32x16 x[4];
  {
    acc_init_bias<4>(x, ctab + 272);
    net_layer<4, SA_L4_STEPS>(ring, x, [&](int s) { return elu_s(a1[s / 16][s % 16]); });
    DYN_PHASE(11);
  }
  float vis;
  {
    f32x16 a5[4], a6[4];
    acc_init_bias<4>(a5, ctab + 400);
    DYN_PHASE(12);
    net_layer<4, SA_L5_STEPS>(ring, a5, [&](int s) {
      const float r = elu1(x[s / 16][s % 16]);  // x = ELU(base_fc.2)
      x[s / 16][s % 16] = r;
      return r * wgt;
    });
    DYN_PHASE(13);
    acc_init_bias<4>(a6, ctab + 528);
    net_layer<4, SA_L5_STEPS>(ring, a6, [&](int s) {
      const float r = elu_s(a5[s / 16][s % 16]);
      a5[s / 16][s % 16] = r;
      return r;
    });
    DYN_PHASE(14);
    vis = sigmoid1(elu1(row_dot<4>(a5, ctab) + ctab[256])) * msk;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[t][r] += elu1(a6[t][r]);
  }
  float vis2;
  {
    f32x16 a7[4];
    acc_init_bias<4>(a7, ctab + 656);
    DYN_PHASE(15);
    net_layer<4, SA_L5_STEPS>(ring, a7, [&](int s) { return x[s / 16][s % 16] * vis; });
    DYN_PHASE(16);
    acc_elu_s(a7);
    vis2 = sigmoid1(row_dot<4>(a7, ctab + 128) + ctab[257]) * msk;
  }
  DYN_PHASE(17);
  // ---- outputs: x and vis2 in lane order, visibility-weighted statistics per point ----
  const bool tile_ok = tile < p.n_tiles_a;
  if (STORE_X && tile_ok) {
    float4* xw = reinterpret_cast<float4*>(p.ws + p.o.off_x) + tile * 16 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) nt_store4<4>(xw + (t * 4 + q) * 64, make_float4(x[t][q * 4], x[t][q * 4 + 1], x[t][q * 4 + 2], x[t][q * 4 + 3]));
    if (h == 0) nt_store1<4>(p.ws + p.o.off_vis + (tile * 32 + (lane & 31)) * 4, vis2);  // (.x of the row's record; its colour went out with the gather loads)
  }
  DYN_PHASE(18);
  if constexpr (VSEG == 0) {
    // ---- dense rows: visibility-weighted statistics through LDS tables (the whole workgroup LDS is free now: no weights are left to stream) ----
    const DenseRows& d = *dr;
    const int tid = threadIdx.x;
    float vsum, nvalid;
    dense_all2(d, vis2, msk, vsum, nvalid);
    const float w2 = vis2 / (vsum + 1e-8f);
    const float wmean = (vsum / (vsum + 1e-8f)) / (float)V;  // mean over the views of w2 (its sum, over V)
    // x goes into LDS ROW-major in 16-byte quads (row stride 132 floats: 16 lanes of a quad access start 4 banks apart, conflict-free);
    // a (point, quad) task walks the V rows of its point twice -- mean, then sum w (x - mean)^2, as the lane segments do -- and
    // stores both quads of the point record with one 16-byte store each.  One table pass, one barrier.
    constexpr int XS = 132;
    float* tab = lds_base;             // [256 rows][XS]
    float* wrow = lds_base + 256 * XS;  // [256] w2 of every row
    const long point0 = d.point0;
    // (the last dense_all2 barrier also retired every wave's reads of the constant tables and of the weight ring)
    {
      float4* mine = reinterpret_cast<float4*>(tab + d.rw * XS) + h;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[8 * t + 2 * q] = make_float4(x[t][q * 4], x[t][q * 4 + 1], x[t][q * 4 + 2], x[t][q * 4 + 3]);
      if (h == 0) wrow[d.rw] = w2;
    }
    __syncthreads();
    for (int task = tid; task < d.PTW * 32; task += DYN_VIEW_THREADS) {
      const int pnt = task >> 5, qd = task & 31;  // quad qd = 8 t + 2 q + half: features 32 t + 8 q + 4 half + (0..3)
      const int pb = dense_pt_base(d, pnt), pn = dense_pt_cnt(d, pnt);  // (regular flavour: pnt V and V -- a constant in the instantiations with a compile-time view count)
      const float* src = tab + pb * XS + qd * 4;
      const float* w = wrow + pb;
      float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
      int k = 0;
      for (; k + 2 <= pn; k += 2) {
        const float4 x0 = dense_quad(src, k, XS, pn), x1 = dense_quad(src, k + 1, XS, pn);
        const float w0 = dense_w(w, k, pn), w1 = dense_w(w, k + 1, pn);
        m0.x = fmaf(w0, x0.x, m0.x); m0.y = fmaf(w0, x0.y, m0.y); m0.z = fmaf(w0, x0.z, m0.z); m0.w = fmaf(w0, x0.w, m0.w);
        m1.x = fmaf(w1, x1.x, m1.x); m1.y = fmaf(w1, x1.y, m1.y); m1.z = fmaf(w1, x1.z, m1.z); m1.w = fmaf(w1, x1.w, m1.w);
      }
      if (k < pn) {
        const float4 x0 = dense_quad(src, k, XS, pn);
        const float w0 = dense_w(w, k, pn);
        m0.x = fmaf(w0, x0.x, m0.x); m0.y = fmaf(w0, x0.y, m0.y); m0.z = fmaf(w0, x0.z, m0.z); m0.w = fmaf(w0, x0.w, m0.w);
      }
      const float mean[4] = {m0.x + m1.x, m0.y + m1.y, m0.z + m1.z, m0.w + m1.w};
      float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
      for (k = 0; k + 2 <= pn; k += 2) {
        const float4 x0 = dense_quad(src, k, XS, pn), x1 = dense_quad(src, k + 1, XS, pn);
        const float w0 = dense_w(w, k, pn), w1 = dense_w(w, k + 1, pn);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w}, b0[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = a0[e] - mean[e], d1 = b0[e] - mean[e];
          v0[e] = fmaf(w0, d0 * d0, v0[e]);
          v1[e] = fmaf(w1, d1 * d1, v1[e]);
        }
      }
      if (k < pn) {
        const float4 x0 = dense_quad(src, k, XS, pn);
        const float w0 = dense_w(w, k, pn);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = a0[e] - mean[e];
          v0[e] = fmaf(w0, d0 * d0, v0[e]);
        }
      }
      const long pt = point0 + pnt;
      if (pt < p.n_pts) {
        const int g = 4 * (qd >> 3) + ((qd >> 1) & 3);  // record 4 t + q of half qd & 1
        float4* gin = reinterpret_cast<float4*>(p.ws + p.o.off_gin) + point_rec(p, pt, qd & 1, SB_GIN_RECS);
        nt_store4<4>(gin + g * 64, make_float4(mean[0], mean[1], mean[2], mean[3]));
        nt_store4<4>(gin + (16 + g) * 64, make_float4(v0[0] + v1[0], v0[1] + v1[1], v0[2] + v1[2], v0[3] + v1[3]));
      }
    }
    if (valid && d.rw == d.base) {  // the point's first row (view 0 in the regular flavour)
      float4* gin = reinterpret_cast<float4*>(p.ws + p.o.off_gin) + point_rec(p, point, h, SB_GIN_RECS);
      gin[32 * 64] = make_float4(h == 0 ? wmean : 1.0f, 0.f, 0.f, 0.f);
      if (h == 0) p.ws[p.o.off_nvalid + point] = nvalid;
    }
  } else {
  const float w2 = vis2 / (seg_sum<VSEG>(vis2, V, seg_base) + 1e-8f);
  const float wmean = seg_sum<VSEG>(w2, V, seg_base) / (float)V;
  const float nvalid = seg_sum<VSEG>(msk, V, seg_base);
  float4* gin = reinterpret_cast<float4*>(p.ws + p.o.off_gin) + (valid ? point_rec(p, point, h, SB_GIN_RECS) : 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float m[4], vv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = x[t][q * 4 + e] * w2;
      seg_sum4<VSEG>(m[0], m[1], m[2], m[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = x[t][q * 4 + e] - m[e];
        vv[e] = w2 * (d * d);
      }
      seg_sum4<VSEG>(vv[0], vv[1], vv[2], vv[3]);
      const int g = t * 4 + q;
      const int sel = g & (VSEG - 1);  // spread the 32 row groups over the segment's real lanes
      const bool mine = valid && (view == (sel < V ? sel : 0));
      if (mine) {
        nt_store4<4>(gin + g * 64, make_float4(m[0], m[1], m[2], m[3]));
        nt_store4<4>(gin + (16 + g) * 64, make_float4(vv[0], vv[1], vv[2], vv[3]));
      }
    }
  if (valid && view == 0) {
    gin[32 * 64] = make_float4(h == 0 ? wmean : 1.0f, 0.f, 0.f, 0.f);
    if (h == 0) p.ws[p.o.off_nvalid + point] = nvalid;
  }
  }
}

// ===================================================================================================================
// A: per point-view chain
// ===================================================================================================================
// VC > 0 (dense flavour only): the view count as a compile-time constant -- the per-row division by V, the (point, slot) task loops of the LDS reductions
// and their loads unroll (round 5: instantiated for the 11 static views of the Nvidia evaluation, the kernel that is 42 % of its frame)
// RAG (dense flavour only): ragged rows -- the workgroup's rows are the valid (point, view) pairs of the points the launch's plan gives it (k_ragged_*)
template <int VSEG, int VC = 0, bool RAG = false>
__global__ void __launch_bounds__(DYN_VIEW_THREADS, 2) k_static_views(StaticArgs p) {
  static_assert(VC == 0 || VSEG == 0, "a compile-time view count is a dense-rows specialisation");
  static_assert(!RAG || VSEG == 0, "ragged rows are a dense-rows flavour");
  if (RAG && (int)blockIdx.x >= p.rg_wg[0]) return;  // the grid is the plan's upper bound
  constexpr int PHASE_KID = 0;
  (void)PHASE_KID;
  float* lds = reinterpret_cast<float*>(dyn_smem);
  float* ctab = lds + 2 * NET_CHUNK;  // [SA_CT]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  for (int i = tid; i < SA_CT; i += DYN_VIEW_THREADS) ctab[i] = p.blob[ST_OFF_CTA + i];
  NetRing ring;
  DYN_PHASE(0);
  // the per-point layers (ray_dir_fc.0's pts columns, base_fc.0's statistics columns: VSEG >= 8 and the dense flavour) are read straight from the stream
  // by each wave, not through the ring: the ring starts behind L1P and skips L3P.  VSEG = 4 (64 points per workgroup: the tiles do not fit the LDS)
  // runs both as ordinary per-view layers through the ring.
  constexpr bool POOLED = VSEG >= 8 || VSEG == 0;
  constexpr int SA_L1P_CHUNKS = net_layer_chunks(8, SA_L1P_STEPS);
  constexpr int SA_POOLED_AT = SA_L1P_CHUNKS + net_layer_chunks(8, SA_L1V_STEPS) + net_layer_chunks(2, SA_L2_STEPS);
  constexpr int LDS_FLOATS = 2 * NET_CHUNK + SA_CT + POOL_FLOATS(SA_NX) + RES_FLOATS + (VSEG == 0 ? DENSE_EXTRA : 0);

  const int V = VC > 0 ? VC : p.V;
  const long tile = (long)blockIdx.x * (DYN_VIEW_THREADS / 64) + wave;
  // VSEG > 0: views occupy a power-of-two segment of VSEG >= V lanes (PT = 32 / VSEG points per wave); lanes view >= V are padding.
  // VSEG == 0 (dense rows): the workgroup's 256 rows are the point-views of its PT = 256 / V points in order, no padding between points --
  // or (RAG) the valid point-views of the plan's points for this workgroup (the row offsets of its points: RAG_PTAB ints of LDS behind everything else).
  const DenseRows dr = RAG ? ragged_rows(V, p.rg_rowtab, p.rg_ptab, blockIdx.x, lds + LDS_FLOATS - DENSE_SCALARS, reinterpret_cast<int*>(lds + LDS_FLOATS))
                           : dense_rows(V, VC > 0 ? 256 / VC : p.PT, lds + LDS_FLOATS - DENSE_SCALARS);
  const int PT = RAG ? dr.PTW : (VC > 0 ? 256 / VC : p.PT);
  const int p_local = VSEG == 0 ? dr.p_local : j / (VSEG == 0 ? 1 : VSEG);
  const int view = VSEG == 0 ? dr.view : (j & (VSEG - 1));
  const long point = VSEG == 0 ? dr.point0 + p_local : tile * PT + p_local;
  const bool valid = (VSEG == 0 ? p_local < PT : view < V) && (point < p.n_pts);
  const int seg_base = 0;
  const long pv = valid ? point * V + view : 0;
  // (the ring's first chunk is requested BEHIND the ragged flavour's table loads: the row entries then arrive ahead of 48 KiB of weights)
  net_ring_init(ring, p.blob + ST_OFF_A + (POOLED ? (size_t)SA_L1P_CHUNKS * NET_CHUNK : 0), SA_CHUNKS - (POOLED ? SA_L1P_CHUNKS : 0), lds,
                SA_POOLED_AT - (POOLED ? SA_L1P_CHUNKS : 0), POOLED ? net_layer_chunks(8, SA_L3P_STEPS) : 0, DYN_VIEW_THREADS);

  // ---- gather the lane's inputs ----
  float msk = valid ? p.mask[pv] : 0.f;
  float4 rd = valid ? reinterpret_cast<const float4*>(p.ray_diff)[pv] : make_float4(0.f, 0.f, 0.f, 0.f);
  float px = 0.f, py = 0.f, pz = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  if (valid) {
    px = p.pts[point * 3]; py = p.pts[point * 3 + 1]; pz = p.pts[point * 3 + 2];
    cx = p.centers[view * 16 + 12]; cy = p.centers[view * 16 + 13]; cz = p.centers[view * 16 + 14];
  }
  // Exactly 3 views / a chunk of exactly 3 rays / 3 samples per ray: the reference's torch.cross (render_ray.py:392, no dim) runs over that axis, not over
  // xyz (csrc/dyn_device.h).  Those moments are formed here, before the first layer's accumulators are live; every other shape skips the branch.
  const int cross_axis = dyn_src_cross_axis(V, p.R, p.S);
  float mq[3] = {0.f, 0.f, 0.f};
  if (cross_axis != DYN_CROSS_XYZ && valid) {
    const int ray = (int)(point / p.S), smp = (int)(point - (long)ray * p.S);
    dyn_src_moment_over_axis(cross_axis, view, ray, smp,
        [&](int, int rr, int ss, float (&q)[3]) { const float* s3 = p.pts + ((long)rr * p.S + ss) * 3; q[0] = s3[0]; q[1] = s3[1]; q[2] = s3[2]; },
        [&](int vv, float (&c3)[3]) { c3[0] = p.centers[vv * 16 + 12]; c3[1] = p.centers[vv * 16 + 13]; c3[2] = p.centers[vv * 16 + 14]; }, mq);
  }
  float xin[SA_NX];
  f32x16 a1[8];
  const bool h0 = h == 0;
  if constexpr (POOLED) {
    // ---- ray_dir_fc.0, per-point part: output tile `wave` for the workgroup's points (column j = the point's index in the workgroup) ----
    B6TileW<SA_L1P_STEPS> pw1;
    b6_tile_prefetch<8, SA_L1P_STEPS>(p.blob + ST_OFF_A, wave, pw1);
    const int npw = VSEG == 0 ? PT : (DYN_VIEW_THREADS / 64) * PT;  // points of this workgroup
    const long qp = VSEG == 0 ? dr.point0 + j : (long)blockIdx.x * npw + j;
    const bool qok = j < npw && qp < p.n_pts;
    const float qx = qok ? p.pts[qp * 3] : 0.f, qy = qok ? p.pts[qp * 3 + 1] : 0.f, qz = qok ? p.pts[qp * 3 + 2] : 0.f;
    float in1p[SA_L1P_STEPS];
    octave_embed<5>(qx, h, in1p);
    octave_embed<5>(qy, h, in1p + 5);
    octave_embed<5>(qz, h, in1p + 10);
    in1p[15] = h0 ? qx : qy;
    in1p[16] = h0 ? qz : 0.f;
    f32x16 accp[1];
    acc_zero(accp);
    b6_tile_apply<SA_L1P_STEPS>(pw1, accp[0], [&](int s) { return in1p[s]; });
    float* res = ctab + SA_CT + POOL_FLOATS(SA_NX);
    res_put(res, wave, j, h, accp[0]);
    __syncthreads();
    res_get(res, VSEG == 0 ? (dr.p_local < dr.PTW ? dr.p_local : dr.PTW - 1) : wave * PT + p_local, h, a1);
  } else {
    float in1p[SA_L1P_STEPS];
    octave_embed<5>(px, h, in1p);
    octave_embed<5>(py, h, in1p + 5);
    octave_embed<5>(pz, h, in1p + 10);
    in1p[15] = h0 ? px : py;
    in1p[16] = h0 ? pz : 0.f;
    acc_zero(a1);
    net_layer<8, SA_L1P_STEPS>(ring, a1, [&](int s) { return in1p[s]; });
  }
  {
    // ---- per-view part: Pluecker coordinates of the source ray through the sample (render_ray.py:380-396), ray_diff, bias ----
    float c6[6];
    unit3(px - cx, py - cy, pz - cz, c6[0], c6[1], c6[2]);
    c6[3] = cy * c6[2] - cz * c6[1];
    c6[4] = cz * c6[0] - cx * c6[2];
    c6[5] = cx * c6[1] - cy * c6[0];
    if (cross_axis != DYN_CROSS_XYZ) { c6[3] = mq[0]; c6[4] = mq[1]; c6[5] = mq[2]; }
    float in1[SA_L1V_STEPS];
#pragma unroll
    for (int c = 0; c < 6; ++c) octave_embed<5>(c6[c], h, in1 + c * 5);
    // raw inputs [Pluecker | ray_diff | 1 | -], even entries to half 0, odd ones to half 1.  Written as selects between scalars: a
    // select between two elements of a local array becomes a lane-indexed load of the array from scratch memory (64 B per lane)
    in1[30] = h0 ? c6[0] : c6[1];
    in1[31] = h0 ? c6[2] : c6[3];
    in1[32] = h0 ? c6[4] : c6[5];
    in1[33] = h0 ? rd.x : rd.y;
    in1[34] = h0 ? rd.z : rd.w;
    in1[35] = h0 ? 1.0f : 0.0f;
    DYN_PHASE(1);
    net_layer<8, SA_L1V_STEPS>(ring, a1, [&](int s) { return in1[s]; });
    DYN_PHASE(2);
  }
  {
    f32x16 a2[2];
    acc_init_bias<2>(a2, ctab + 784);
    // The gathered colours / features are first needed after ray_dir_fc: their loads are issued from inside the layer's feed (half
    // way through it), which keeps 18 registers free during the 256-wide first layer and hides the HBM latency under this layer.
    // (requested half way through the layer; right behind the layer's first ring acquire -- slot 8 -- measured 1 % slower in round 6)
    net_layer<2, SA_L2_STEPS>(ring, a2, [&](int s) {
      if (s == 64) {
#pragma unroll
        for (int q = 0; q < 18; ++q) {
          const int ch = h == 0 ? q : 18 + q;
          xin[q] = (valid && ch < 35) ? nt_load1<2>(p.rgb_feat + pv * 35 + ch) : 0.f;
        }
      }
      return elu_s(a1[s / 16][s % 16]);  // ELU of ray_dir_fc.0 where it is consumed
    });
    const float* rf = p.ws + p.o.off_ref + (valid ? (point / p.S) * 36 : 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) xin[18 + r] = a2[0][r] * rf[dyn_fi(r, h)];
#pragma unroll
    for (int r = 0; r < 3; ++r) xin[34 + r] = h == 0 ? a2[1][r] * rf[32 + r] : 0.f;
    // The source colour of the row goes into the row's record next to vis2 (round 5): the blend kernel used to fetch it from rgb_feat, 12 bytes out of
    // every 140-byte row -- i.e. nearly every 128-byte line of that 294 MB tensor a second time.  (.yzw here, .x = vis2 at the end of the chain.)
    if (h == 0 && tile < p.n_tiles_a) {
      float* rec = p.ws + p.o.off_vis + (tile * 32 + j) * 4;
      nt_store1<4>(rec + 1, xin[0]); nt_store1<4>(rec + 2, xin[1]); nt_store1<4>(rec + 3, xin[2]);
    }
  }
  DYN_PHASE(4);
  if (p.mask_rgb) {
    const float s3 = (xin[0] + xin[1]) + xin[2];  // the colour channels live in the h = 0 lanes
    msk *= (__shfl(s3, j) > 1e-3f) ? 1.0f : 0.0f;
  }
  // ---- pooling weights (mlp_network.py:462-471) ----
  float wgt;
  if (p.anti_alias) {
    // padding lanes (and, in the dense flavour, the idle tail rows, which shadow the last point) never set the minimum over the views
    const float e = ((VSEG == 0 ? p_local < PT : view < V)) ? aa_exp(ctab[258], rd.w) : 3.0e38f;
    // (ragged rows: the minimum runs over ALL views of the point, masked ones included -- mlp_network.py:465-468 --, so it comes from the plan)
    const float emin = RAG ? (valid ? p.rg_emin[point] : e) : views_min<VSEG>(dr, e);
    wgt = (e - emin) * msk;
  } else {
    wgt = msk;
  }
  const float wraw = views_sum<VSEG>(dr, wgt);
  wgt = wgt / (wraw + 1e-8f);

  DYN_PHASE(5);
  base_fc0<VSEG, SA_NX>(ring, p.blob + ST_OFF_A + (size_t)SA_POOLED_AT * NET_CHUNK, xin, wgt, V, view, p_local, ctab + SA_CT, a1, &dr, wraw / (wraw + 1e-8f));
  DYN_PHASE(10);
  views_tail<VSEG, true>(ring, a1, p, ctab, wgt, msk, tile, point, valid, view, seg_base, &dr, lds, VC);
  DYN_PHASE(20);
}

// ===================================================================================================================
// B: per point chain with the ray attention
// ===================================================================================================================
#define SB_KL_FLOATS 4096         // K of one head, lane-native: [4 tiles][4][64 lanes][4]
#define SB_VL_LD 132              // V of one head: [32 features][128 points + pad]
#define SB_VL_FLOATS (32 * SB_VL_LD)

// workgroup barrier that publishes LDS writes only: `__syncthreads()` also waits for vmcnt(0), i.e. for the weight ring's DMA pieces in flight
// (round 4: the attention's eight barriers per pass no longer drain the ring's queue)
__device__ __forceinline__ void lds_barrier() {
#if defined(__AMDGCN__) && DYN_POINTS_DUO
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
#else
  __syncthreads();
#endif
}

// DYN = false: DynibarStatic (no positional encoding; outputs sigma and the point part of rgb_fc.0)
// DYN = true : DynibarDynamic (+ sinusoid positional encoding, ref_pts_fc, rgb_fc on [feature | PE(view dir)]; outputs raw [R,S,4])
// PHASE 0: the whole chain in one launch (rays of <= 128 samples: a ray's tiles sit in one workgroup and K / V never leave the chip).
// Longer rays run it in two launches: PHASE 1 ends after the Q/K/V projections and stores g, q, k, v per tile; PHASE 2 picks them up,
// streams the ray's keys through LDS in blocks of 128 with a running softmax, and finishes the chain.
constexpr int SB_CHUNKS_QKV = pts_layer_chunks(8, 129) + pts_layer_chunks(4, 129) + 3 * pts_layer_chunks(4, 64);
static int dyn_cu_count() {
  static int n_cu[DYN_MAX_DEVICES] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int slot = (dev >= 0 && dev < DYN_MAX_DEVICES) ? dev : 0;
  if (n_cu[slot] == 0) {
    hipDeviceProp_t prop;
    n_cu[slot] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu[slot];
}

// grid of k_net_points<., 0>: one workgroup per row tile -- or, in the persistent build (-DDYN_POINTS_PERSIST=1), one per CU, each walking every n_cu-th tile
static dim3 points_grid(dim3 full) {
#if DYN_POINTS_DUO && DYN_POINTS_PERSIST
  static int n_cu[DYN_MAX_DEVICES] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int slot = (dev >= 0 && dev < DYN_MAX_DEVICES) ? dev : 0;
  if (n_cu[slot] == 0) {
    hipDeviceProp_t prop;
    n_cu[slot] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  static const int no_persist = getenv("DYN_POINTS_NO_PERSIST") != nullptr;  // developer A/B: one workgroup per row tile, like the long-ray phases
  return (no_persist || full.x <= (unsigned)n_cu[slot]) ? full : dim3((unsigned)n_cu[slot]);
#else
  return full;
#endif
}

template <bool DYN, int PHASE>
__global__ void __launch_bounds__(DYN_NET_THREADS, 1) k_net_points(StaticArgs p) {
  constexpr int PHASE_KID = 1;
  (void)PHASE_KID;
  DYN_PHASE(0);
  DYN_CLAIM_REGISTER_FILE();
  float* lds = reinterpret_cast<float*>(dyn_smem);
  float* ctab = lds + PTS_RING_SLOTS * PTS_CHUNK;  // [SB_CT] / [DB_CT]
  float* Kl = ctab + (DYN ? DB_CT : SB_CT);
  float* Vl = Kl + SB_KL_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  constexpr int CT = DYN ? DB_CT : SB_CT;
  for (int i = tid; i < CT; i += DYN_NET_THREADS) ctab[i] = p.blob[(DYN ? DY_OFF_CTB : ST_OFF_CTB) + i];
  PtsRing ring;
  if (PHASE == 2)
    pts_ring_init(ring, p.blob + (DYN ? DY_OFF_B : ST_OFF_B) + (size_t)SB_CHUNKS_QKV * PTS_CHUNK, (DYN ? DB_CHUNKS : SB_CHUNKS) - SB_CHUNKS_QKV, lds);
  else
    pts_ring_init(ring, p.blob + (DYN ? DY_OFF_B : ST_OFF_B), PHASE == 1 ? SB_CHUNKS_QKV : (DYN ? DB_CHUNKS : SB_CHUNKS), lds);
  DYN_PHASE_RING_KID(ring, 1);
  // The three-slot ring's own barrier waits for vmcnt only (ring3_barrier) and gfx950's s_barrier does not imply an LDS wait: publish the constant
  // table to the other waves here, once per workgroup, instead of relying on the first full barrier of the attention coming before its first read.
  lds_barrier();

  const int TPR = p.TPR;
  const float one_h0 = h == 0 ? 1.0f : 0.0f;
  // -DDYN_POINTS_PERSIST=1 (measured in round 4, NOT the default): PHASE 0 as a persistent kernel -- one workgroup per CU walks row tiles blockIdx.x,
  // + gridDim.x, ...; the tail of a pass pulls the next tile's geometry_fc inputs into the L2 (one dword per 16 bytes, summed into a value nobody
  // reads; holding the 129 inputs themselves through the tail layers costs 80 spilled registers) and requests the next pass's first weight chunks.
  // Motive: cycle stamps show a workgroup's first 15 k of ~104 k cycles going into its own start (132 KB of inputs at the rate a CU gets from HBM,
  // the first weight chunks) with nothing else resident on the CU.  Result: k_net_points 487 us against 465-475 us with one workgroup per row tile,
  // frame +2 ms: the prefetch doubles the L2 -> CU traffic of the inputs and the static tile assignment loses the dispatcher's balancing.
  constexpr bool PERSIST = DYN_POINTS_PERSIST && DYN_POINTS_DUO && PHASE == 0;
  const long n_wg = (p.n_tiles_b + 3) / 4;
  // the geometry_fc inputs of row tile wgi: this lane's 33 float4 records
  auto gin_src = [&](long wgi, bool& valid_) DYN_INLINE_LAMBDA {
    const long tile_ = wgi * 4 + wave;
    const long ray_ = tile_ / TPR;
    valid_ = (ray_ < p.R) && ((int)(tile_ - ray_ * TPR) * 32 + j < p.S);
    return reinterpret_cast<const float4*>(p.ws + p.o.off_gin) + (tile_ < p.n_tiles_b ? tile_ : 0) * SB_GIN_RECS * 64 + lane;
  };
#if DYN_POINTS_DUO
  ring.wrap = PERSIST ? 1 : 0;
#endif
  long wgi = blockIdx.x;
  do {  // (a loop only in the persistent form)
  const bool more = PERSIST && wgi + gridDim.x < n_wg;
#if DYN_POINTS_DUO
  ring.more = more;
#endif
  const long tile = wgi * 4 + wave;
  const long ray = tile / TPR;
  const int kt_self = (int)(tile - ray * TPR);
  const int smp = kt_self * 32 + j;
  const bool valid = (ray < p.R) && (smp < p.S);
  const long point = valid ? ray * p.S + smp : 0;
  const float nvalid = valid ? p.ws[p.o.off_nvalid + point] : 0.f;

  float4* hand = reinterpret_cast<float4*>(p.ws + p.o.off_qkvg) + (PHASE == 0 ? 0 : tile * 4096) + lane;  // [which][t][q][64 lanes]
  f32x16 g[4];
  if (PHASE == 2) {
    if (ray < p.R) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = hand[(t * 4 + q) * 64];
          g[t][q * 4] = v.x; g[t][q * 4 + 1] = v.y; g[t][q * 4 + 2] = v.z; g[t][q * 4 + 3] = v.w;
        }
    } else {
      acc_zero(g);
    }
  } else {
    f32x16 a9[8];
    {
      float gin[129];
      bool gv;
      const float4* src = gin_src(wgi, gv);
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const float4 v = gv ? nt_load4<4>(src + i * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
        gin[i * 4] = v.x; gin[i * 4 + 1] = v.y; gin[i * 4 + 2] = v.z; gin[i * 4 + 3] = v.w;
      }
      gin[128] = gv ? src[32 * 64].x : (h == 1 ? 1.0f : 0.f);
      acc_zero(a9);
      PTS_LAYER(8, 129)(ring, a9, [&](int s) { return gin[s]; });
    }
    acc_zero(g);
    PTS_LAYER(4, 129)(ring, g, [&](int s) { return s < 128 ? elu_s(a9[s / 16][s % 16]) : one_h0; });  // ELUs ride in the consumer's feed
    acc_elu(g);
  }
  if (DYN && PHASE != 2) {
    // globalfeat + pos_encoding (mlp_network.py:284): table rows in D-layout order [position][half][64]
    const float4* pe = reinterpret_cast<const float4*>(p.blob + DY_OFF_POSENC + ((valid ? smp : 0) * 2 + h) * 64);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = pe[t * 4 + q];
        g[t][q * 4] += v.x; g[t][q * 4 + 1] += v.y; g[t][q * 4 + 2] += v.z; g[t][q * 4 + 3] += v.w;
      }
  }
  // ---- multi-head self-attention over the samples of the ray (mlp_network.py:13-31, 56-104) ----
  DYN_PHASE(1);  // geometry_fc done
  f32x16 att[4];
  const float inv_temp = 1.0f / 5.656854249492381f;  // d_k ** 0.5
  const bool q_ok = nvalid > 1.0f;                    // mask = (num_valid_obs > 1), applied along the query axis
  if (PHASE == 2) {
    // ---- long rays: keys / values of the whole ray come back from the workspace, 128 at a time, under a running softmax ----
    f32x16 qh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = hand[(16 + t * 4 + q) * 64];
        qh[t][q * 4] = v.x * inv_temp; qh[t][q * 4 + 1] = v.y * inv_temp; qh[t][q * 4 + 2] = v.z * inv_temp; qh[t][q * 4 + 3] = v.w * inv_temp;
      }
    const float4* ray_hand = reinterpret_cast<const float4*>(p.ws + p.o.off_qkvg) + ray * TPR * 4096;  // tile kt of the ray at + kt * 4096
    const int n_blocks = TPR / 4;
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      float run_max = -3.0e38f, run_sum = 0.f;
      f32x16 oh;
#pragma unroll
      for (int r = 0; r < 16; ++r) oh[r] = 0.f;
      for (int kb = 0; kb < n_blocks; ++kb) {
        __syncthreads();  // the previous block's K/V images are no longer read
        {
          const float4* kv = ray_hand + (long)(kb * 4 + wave) * 4096 + lane;  // this wave copies key tile kb * 4 + wave
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            reinterpret_cast<float4*>(Kl)[(wave * 4 + q) * 64 + lane] = kv[(32 + hd * 4 + q) * 64];
            const float4 v = kv[(48 + hd * 4 + q) * 64];
            Vl[dyn_fi(q * 4 + 0, h) * SB_VL_LD + wave * 32 + j] = v.x;
            Vl[dyn_fi(q * 4 + 1, h) * SB_VL_LD + wave * 32 + j] = v.y;
            Vl[dyn_fi(q * 4 + 2, h) * SB_VL_LD + wave * 32 + j] = v.z;
            Vl[dyn_fi(q * 4 + 3, h) * SB_VL_LD + wave * 32 + j] = v.w;
          }
        }
        __syncthreads();
        f32x16 sc[4];
        acc_zero(sc);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 a = reinterpret_cast<const float4*>(Kl)[(kt * 4 + q) * 64 + lane];
            sc[kt] = mfma32(a.x, qh[hd][q * 4 + 0], sc[kt]);
            sc[kt] = mfma32(a.y, qh[hd][q * 4 + 1], sc[kt]);
            sc[kt] = mfma32(a.z, qh[hd][q * 4 + 2], sc[kt]);
            sc[kt] = mfma32(a.w, qh[hd][q * 4 + 3], sc[kt]);
          }
        float mx = run_max;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool key_ok = (kb * 4 + kt) * 32 + dyn_fi(r, h) < p.S;
            float v = q_ok ? sc[kt][r] : -1e9f;
            v = key_ok ? v : -3.0e38f;
            sc[kt][r] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float rescale = __expf(run_max - mx);  // 0 for the first block
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = sc[kt][r] > -1.0e38f ? __expf(sc[kt][r] - mx) : 0.f;
            sc[kt][r] = e;
            sum += e;
          }
        sum += __shfl_xor(sum, 32);
        run_sum = run_sum * rescale + sum;
        run_max = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) oh[r] *= rescale;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(Vl + j * SB_VL_LD + kt * 32 + 8 * q + 4 * h);
            oh = mfma32(a.x, sc[kt][q * 4 + 0], oh);
            oh = mfma32(a.y, sc[kt][q * 4 + 1], oh);
            oh = mfma32(a.z, sc[kt][q * 4 + 2], oh);
            oh = mfma32(a.w, sc[kt][q * 4 + 3], oh);
          }
      }
      const float inv = 1.0f / run_sum;
#pragma unroll
      for (int r = 0; r < 16; ++r) att[hd][r] = oh[r] * inv;
    }
  } else {
    f32x16 qh[4], kh[4], vh[4];
    acc_zero(qh); acc_zero(kh); acc_zero(vh);
    PTS_LAYER(4, 64)(ring, qh, [&](int s) { return g[s / 16][s % 16]; });
    PTS_LAYER(4, 64)(ring, kh, [&](int s) { return g[s / 16][s % 16]; });
    PTS_LAYER(4, 64)(ring, vh, [&](int s) { return g[s / 16][s % 16]; });
    if (PHASE == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          hand[(t * 4 + q) * 64] = make_float4(g[t][q * 4], g[t][q * 4 + 1], g[t][q * 4 + 2], g[t][q * 4 + 3]);
          hand[(16 + t * 4 + q) * 64] = make_float4(qh[t][q * 4], qh[t][q * 4 + 1], qh[t][q * 4 + 2], qh[t][q * 4 + 3]);
          hand[(32 + t * 4 + q) * 64] = make_float4(kh[t][q * 4], kh[t][q * 4 + 1], kh[t][q * 4 + 2], kh[t][q * 4 + 3]);
          hand[(48 + t * 4 + q) * 64] = make_float4(vh[t][q * 4], vh[t][q * 4 + 1], vh[t][q * 4 + 2], vh[t][q * 4 + 3]);
        }
      return;
    }
    const int wave0 = wave - kt_self;                   // first wave of this ray inside the workgroup
    DYN_PHASE(2);  // Q, K, V projections done
    // The attention matmuls on the split engine too (round 2; the native fp32 MFMA needs 16 instructions of 64 cycles per 32 x 32 x 32 block,
    // the split engine 6 of 32).  scores^T [key x query] = K . q^T: A = a key tile's K in ITS lanes' register order (feature of slot e of group
    // m = fi(8 m + e, h): the same enumeration on both operands), deposited in LDS as hi | mid half-float images; B = this wave's q.
    // out^T [feature x query] = V^T . P: A = V^T rows read from the [feature][key] table and split on the fly, B = the probabilities.
    auto split8 = [&](const float (&v)[8], u32x4v& hi, u32x4v& mid) {
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        unsigned h_, m_, l_;
        split3_pair(v[2 * p2], v[2 * p2 + 1], h_, m_, l_);
        hi[p2] = h_; mid[p2] = m_;
      }
    };
    u32x4v* Kimg = reinterpret_cast<u32x4v*>(Kl);  // [key tile (wave)][group m][hi | mid][64 lanes]
#pragma unroll
    for (int hd = 0; hd < 4; ++hd) {
      lds_barrier();  // the previous head's K/V images are no longer read
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float kv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) kv[e] = kh[hd][8 * m + e];
        u32x4v khi, kmid;
        split8(kv, khi, kmid);
        Kimg[((wave * 2 + m) * 2 + 0) * 64 + lane] = khi;
        Kimg[((wave * 2 + m) * 2 + 1) * 64 + lane] = kmid;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Vl[dyn_fi(r, h) * SB_VL_LD + wave * 32 + j] = vh[hd][r];
      u32x4v qhi[2], qmid[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float qv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = qh[hd][8 * m + e] * inv_temp;
        split8(qv, qhi[m], qmid[m]);
      }
      lds_barrier();
      f32x16 sc[4];
      acc_zero(sc);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < TPR) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const u32x4v ahi = Kimg[(((wave0 + kt) * 2 + m) * 2 + 0) * 64 + lane], amid = Kimg[(((wave0 + kt) * 2 + m) * 2 + 1) * 64 + lane];
            sc[kt] = mfma_bf16(amid, qhi[m], sc[kt]);
            sc[kt] = mfma_bf16(ahi, qmid[m], sc[kt]);
            sc[kt] = mfma_bf16(ahi, qhi[m], sc[kt]);
          }
        }
      // softmax over the keys; register r of half h is key kt*32 + fi(r,h)
      // (key tiles beyond the ray's own -- kt >= TPR, a uniform condition -- are skipped altogether: a 64-sample ray has two of the four)
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < TPR) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool key_ok = kt * 32 + dyn_fi(r, h) < p.S;
            float v = q_ok ? sc[kt][r] : -1e9f;
            v = key_ok ? v : -3.0e38f;
            sc[kt][r] = v;
            mx = fmaxf(mx, v);
          }
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < TPR) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = sc[kt][r] > -1.0e38f ? __expf(sc[kt][r] - mx) : 0.f;
            sc[kt][r] = e;
            sum += e;
          }
        }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
      f32x16 oh;
#pragma unroll
      for (int r = 0; r < 16; ++r) oh[r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < TPR) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            // slot e of group m is key kt * 32 + fi(8 m + e, h) = 16 m + 4 h + (e & 3) + 8 (e >> 2): two float4 of this lane's feature row of V^T
            const float* vrow = Vl + j * SB_VL_LD + (wave0 + kt) * 32 + 16 * m + 4 * h;
            const float4 v0 = *reinterpret_cast<const float4*>(vrow), v1 = *reinterpret_cast<const float4*>(vrow + 8);
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = sc[kt][8 * m + e] * inv;
            u32x4v ahi, amid, phi, pmid;
            split8(vv, ahi, amid);
            split8(pv, phi, pmid);
            oh = mfma_bf16(amid, phi, oh);
            oh = mfma_bf16(ahi, pmid, oh);
            oh = mfma_bf16(ahi, phi, oh);
          }
        }
      att[hd] = oh;
    }
  }
  {
    f32x16 o[4];
    acc_zero(o);
    DYN_PHASE(3);  // attention heads done
    PTS_LAYER(4, 64)(ring, o, [&](int s) { return att[s / 16][s % 16]; });
    // residual + LayerNorm(eps = 1e-6) over the 128 features (64 here, 64 in the other half's lane)
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[t][r] += g[t][r];
        s1 += o[t][r];
      }
    s1 += __shfl_xor(s1, 32);
    const float mu = s1 * (1.0f / 128.0f);
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = o[t][r] - mu;
        s2 += d * d;
      }
    s2 += __shfl_xor(s2, 32);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / 128.0f) + 1e-6f);
    const float* gam = ctab + h * 64;
    const float* bet = ctab + 128 + h * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = (o[t][r] - mu) * rstd * gam[t * 16 + r] + bet[t * 16 + r];
  }
  DYN_PHASE(4);  // fc + LayerNorm done
  float pf[SB_GIN_RECS];  // (persistent form) one dword of every 16 bytes of the next pass's geometry_fc inputs: pulled into the L2 ~14 k cycles ahead
  if (PERSIST) {
    bool gv;
    const float* nsrc = reinterpret_cast<const float*>(gin_src(more ? wgi + gridDim.x : wgi, gv));
#pragma unroll
    for (int i = 0; i < SB_GIN_RECS; ++i) pf[i] = (more && gv) ? nsrc[i * 256] : 0.f;
  }
  if (!DYN) {
    f32x16 a[4];
    acc_zero(a);
    PTS_LAYER(4, 65)(ring, a, [&](int s) { return s < 64 ? g[s / 16][s % 16] : one_h0; });
    acc_elu_s(a);
    float sigma = row_dot<4>(a, ctab + 256) + ctab[384];
    if (nvalid < 1.0f) sigma = -1e9f;
    if (valid && h == 0) p.raw[point * 4 + 3] = sigma;
    acc_zero(a);
    PTS_LAYER(4, 65)(ring, a, [&](int s) { return s < 64 ? g[s / 16][s % 16] : one_h0; });
    if (valid) {
      float4* dst = reinterpret_cast<float4*>(p.ws + p.o.off_hg) + tile * SB_HG_RECS * 64 + lane;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(t * 4 + q) * 64] = make_float4(a[t][q * 4], a[t][q * 4 + 1], a[t][q * 4 + 2], a[t][q * 4 + 3]);
    }
  } else {
    // ref_pts_fc([globalfeat, PE(pts)])   (mlp_network.py:289-290)
    f32x16 g2[4];
    {
      float pe[17];  // 15 cos|sin pairs (3 coords x 5 octaves), then (x, y), (z, 1)
      float c3[3] = {0.f, 0.f, 0.f};
      if (valid) { c3[0] = p.pts[point * 3]; c3[1] = p.pts[point * 3 + 1]; c3[2] = p.pts[point * 3 + 2]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) octave_embed<5>(c3[c], h, pe + c * 5);
      pe[15] = h == 0 ? c3[0] : c3[1];
      pe[16] = h == 0 ? c3[2] : 1.0f;
      f32x16 a8[8];
      acc_zero(a8);
      PTS_LAYER(8, 81)(ring, a8, [&](int s) { return s < 64 ? g[s / 16][s % 16] : pe[s - 64]; });
      acc_zero(g2);
      PTS_LAYER(4, 129)(ring, g2, [&](int s) { return s < 128 ? elu_s(a8[s / 16][s % 16]) : one_h0; });
    }
    f32x16 a[4];
    acc_zero(a);
    PTS_LAYER(4, 65)(ring, a, [&](int s) {
      if (s >= 64) return one_h0;
      const float r = elu1(g2[s / 16][s % 16]);  // g2 = ELU(out_geometry_fc.2) is kept: rgb_fc reads it again
      g2[s / 16][s % 16] = r;
      return r;
    });
    acc_elu_s(a);
    float sigma = row_dot<4>(a, ctab + 256) + ctab[384] - p.shift;
    if (nvalid < 1.0f) sigma = -1e9f;
    // rgb_fc([globalfeat, PE(view dir)])   (mlp_network.py:299-313)
    float pd[14];  // 12 cos|sin pairs (3 coords x 4 octaves), then (dx, dy), (dz, 1)
    {
      float d3[3] = {0.f, 0.f, 1.f};
      if (valid) unit3(p.ray_d[ray * 3], p.ray_d[ray * 3 + 1], p.ray_d[ray * 3 + 2], d3[0], d3[1], d3[2]);
#pragma unroll
      for (int c = 0; c < 3; ++c) octave_embed<4>(d3[c], h, pd + c * 4);
      pd[12] = h == 0 ? d3[0] : d3[1];
      pd[13] = h == 0 ? d3[2] : 1.0f;
    }
    acc_zero(a);
    PTS_LAYER(4, 78)(ring, a, [&](int s) { return s < 64 ? g2[s / 16][s % 16] : pd[s - 64]; });
    f32x16 b2[2];
    acc_zero(b2);
    PTS_LAYER(2, 65)(ring, b2, [&](int s) { return s < 64 ? elu_s(a[s / 16][s % 16]) : one_h0; });
    acc_elu_s(b2);
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rgb[c] = sigmoid1(row_dot<2>(b2, ctab + 400 + c * 64) + ctab[385 + c]);
      if (nvalid == 0.f) rgb[c] = 0.f;  // masked_fill(sum(mask) == 0, 0)
    }
    if (valid && h == 0) reinterpret_cast<float4*>(p.raw)[point] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
  }
  DYN_PHASE(20);
#if DYN_POINTS_DUO
  if (PERSIST) {
    // the prefetched dwords are summed HERE, behind everything the pass computes (the sum starts from a value the last stores produced, so hipcc cannot
    // raise the additions -- and their wait for the loads -- into the tail layers)
    float sink = nvalid;
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(sink));
#endif
#pragma unroll
    for (int i = 0; i < SB_GIN_RECS; ++i) sink += pf[i];
#if defined(__AMDGCN__)
    asm volatile("" ::"v"(sink));
#endif
    ring3_next_pass(ring);
  }
#endif
  wgi += gridDim.x;
  } while (PERSIST && wgi < n_wg);  // row tiles of this workgroup
}

// ===================================================================================================================
// C: rgb_fc over [globalfeat | x | vis | ray_diff], masked softmax over the views, colour blend
// ===================================================================================================================
// The blend kernel is a stream over the parked features (512 B per row) with a short MFMA chain per row.  A CU streams at ~10 B/cycle
// and a load stalls its wave once the memory queue is full, so a workgroup's loads and layers do not overlap by themselves.  Small
// workgroups (4 waves) with a single 48 KiB weight buffer fit three to a CU (LDS 146 KiB, 168 registers): independent barriers, so
// one workgroup's loads run under another's layers, and the exposed DMA of the single buffer hides the same way.
#define DYN_BLEND_THREADS 256
// THREADS: 256 (lane segments: three workgroups per CU) or 512 (dense rows: the same 256-row partition as the view kernel, since a point's
// rows must sit in one workgroup)
template <int VSEG, int THREADS>
__device__ __forceinline__ void static_blend_body(StaticArgs p) {
  constexpr int PHASE_KID = 2;
  (void)PHASE_KID;
  DYN_PHASE(0);
  float* lds = reinterpret_cast<float*>(dyn_smem);
  float* ctab = lds + NET_CHUNK;  // [SC_CT]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  for (int i = tid; i < SC_CT; i += THREADS) ctab[i] = p.blob[ST_OFF_CTC + i];
  NetRing ring;
  net_ring_init_1(ring, p.blob + ST_OFF_C, SC_CHUNKS, lds, THREADS);
  DYN_PHASE_RING_KID(ring, 2);

  const int V = p.V;
  const long tile = (long)blockIdx.x * (THREADS / 64) + wave;
  const bool tile_ok = tile < p.n_tiles_a;
  // row -> (point, view) exactly as in the view kernel that parked x: lane segments (VSEG > 0) or dense rows (VSEG == 0)
  const DenseRows dr = dense_rows(V, p.PT, ctab + SC_CT);
  const int p_local = VSEG == 0 ? dr.p_local : j / (VSEG == 0 ? 1 : VSEG);
  const int view = VSEG == 0 ? dr.view : (j & (VSEG - 1));
  const long point = VSEG == 0 ? (long)blockIdx.x * p.PT + p_local : tile * p.PT + p_local;
  const bool valid = (VSEG == 0 ? p_local < p.PT : view < V) && (point < p.n_pts);
  const long pv = valid ? point * V + view : 0;

  float msk = valid ? p.mask[pv] : 0.f;
  const float4 rd = valid ? reinterpret_cast<const float4*>(p.ray_diff)[pv] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 vrec = tile_ok ? reinterpret_cast<const float4*>(p.ws + p.o.off_vis)[tile * 32 + j] : make_float4(0.f, 0.f, 0.f, 0.f);  // {vis2, r, g, b} of the row
  const float vis2 = vrec.x;
  const float rgb_in[3] = {valid ? vrec.y : 0.f, valid ? vrec.z : 0.f, valid ? vrec.w : 0.f};
  if (p.mask_rgb && !((rgb_in[0] + rgb_in[1]) + rgb_in[2] > 1e-3f)) msk = 0.f;  // mask = mask * rgb_mask also feeds the masked_fill (mlp_network.py:458-460, 523)
  f32x16 a[4];
  {
    f32x16 x[4];
    const float4* xw = reinterpret_cast<const float4*>(p.ws + p.o.off_x) + (tile_ok ? tile : 0) * 16 * 64 + lane;
    const float4* hg = reinterpret_cast<const float4*>(p.ws + p.o.off_hg) + (valid ? point_rec(p, point, h, SB_HG_RECS) : 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = tile_ok ? nt_load4<8>(xw + (t * 4 + q) * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
        x[t][q * 4] = v.x; x[t][q * 4 + 1] = v.y; x[t][q * 4 + 2] = v.z; x[t][q * 4 + 3] = v.w;
        const float4 b = valid ? hg[(t * 4 + q) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        a[t][q * 4] = b.x; a[t][q * 4 + 1] = b.y; a[t][q * 4 + 2] = b.z; a[t][q * 4 + 3] = b.w;
      }
    const float extra[3] = {h == 0 ? vis2 : rd.x, h == 0 ? rd.y : rd.z, h == 0 ? rd.w : 0.f};
    net_layer<4, SC_L11_STEPS>(ring, a, [&](int s) { return s < 64 ? x[s / 16][s % 16] : extra[s - 64]; });
  }
  f32x16 b2[2];
  acc_init_bias<2>(b2, ctab + 80);
  net_layer<2, 64>(ring, b2, [&](int s) { return elu_s(a[s / 16][s % 16]); });
  acc_elu_s(b2);
  float logit = row_dot<2>(b2, ctab) + ctab[64];
  if (msk == 0.f) logit = -1e9f;
  if (VSEG == 0 ? p_local >= p.PT : view >= V) logit = -3.0e38f;  // padding rows take no share even when every real view is masked (uniform 1/V then)
  const float mx = views_max<VSEG>(dr, logit);
  const float e = (VSEG == 0 && p_local >= p.PT) ? 0.f : __expf(logit - mx);
  const float bw = e / views_sum<VSEG>(dr, e);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = views_sum<VSEG>(dr, rgb_in[c] * bw);
    if (valid && (VSEG == 0 ? dr.rw == dr.base : view == 0) && h == 0) p.raw[point * 4 + c] = v;
  }
  DYN_PHASE(20);
}
// ---- round 4: the blend with its weights RESIDENT in LDS, as persistent workgroups ------------------------------------------------------------------
// rgb_fc.0's per-view part and rgb_fc.2 are 52 packed pairs = 104 KiB: they fit the LDS.  The streaming form above moves three 48 KiB chunks through a
// one-slot ring per 128 rows -- 2.25 x the bytes of the parked x the kernel is there to stream, through the same L2 -> CU path, with the chunk waits
// (8 k of a workgroup's 28-35 k cycles, tools/phasebench.py) covered only by whatever else is resident.  Here one workgroup per CU copies the 52 pairs
// once and walks row tiles; the layers read their A fragments from the resident image (mlp_layer_b6_lds: no ring, no barriers), so in the lane-segment
// flavour the twelve waves of a CU run free of each other, and in the dense flavour only the cross-view reductions still synchronise.
#define SC_L11_PAIRS (((SC_L11_STEPS + 7) / 8) * 4)
#define SC_L12_PAIRS ((64 / 8) * 2)
#define SC_WS_FLOATS ((SC_L11_PAIRS + SC_L12_PAIRS) * B6_PAIR_FLOATS)
// 8 waves (two per SIMD: 256 registers, no spill) since round 6: with the next tile's x prefetched (DYN_BLEND_PREFETCH) the second wave of a SIMD hides what the third
// wave of the 12-wave form (168 registers, 48 B / lane of scratch) was there to hide: 346-349 -> 334 us at the bench shape; the 12-wave form WITH the prefetch spills 108 B / lane: 436 us
#ifndef DYN_BLEND_WS_THREADS
#define DYN_BLEND_WS_THREADS 512
#endif
#if DYN_BLEND_WS
__device__ __forceinline__ long n_units_of(const StaticArgs& p, bool rag, int vseg, int nw) {
  return rag ? p.rg_wg[0] : (vseg == 0 ? (p.n_tiles_a + nw - 1) / nw : p.n_tiles_a);
}
template <int VSEG, int THREADS, bool RAG = false>
__device__ __forceinline__ void static_blend_ws_body(StaticArgs p) {
  float* lds = reinterpret_cast<float*>(dyn_smem);
  float* ctab = lds + SC_WS_FLOATS;  // [SC_CT] (+ the dense flavour's scalar tables behind it)
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  for (int i = tid; i < SC_CT; i += THREADS) ctab[i] = p.blob[ST_OFF_CTC + i];
  {
    // the used pairs of stream C (chunks of B6_CHUNK_PAIRS pairs; the last chunk of a layer is partly filled) -> one compact image
    constexpr int CP = B6_CHUNK_PAIRS, L11_CH = net_layer_chunks(4, SC_L11_STEPS);
    const float4* src = reinterpret_cast<const float4*>(p.blob + ST_OFF_C);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < SC_WS_FLOATS / 4; i += THREADS) {
      const int pair = i / (B6_PAIR_FLOATS / 4), within = i - pair * (B6_PAIR_FLOATS / 4);
      const int sp = pair < SC_L11_PAIRS ? pair : L11_CH * CP + (pair - SC_L11_PAIRS);  // pair index in the chunked stream
      dst[i] = src[(long)sp * (B6_PAIR_FLOATS / 4) + within];
    }
  }
  __syncthreads();
  const float* w11 = lds;
  const float* w12 = lds + SC_L11_PAIRS * B6_PAIR_FLOATS;
  const int V = p.V;
  constexpr int NW = THREADS / 64;
  DenseRows dr = dense_rows(V, p.PT, ctab + SC_CT, 0);
#ifndef DYN_BLEND_PREFETCH
#define DYN_BLEND_PREFETCH 1
#endif
  // Round 6: the parked x of the NEXT tile is requested while this tile is still being worked on -- without a second register set.  x is dead once rgb_fc.0
  // has consumed it, and rgb_fc.2 consumes its input `a` tile by tile: x tile t of the next unit is loaded into the registers a tile of `a` has just left
  // (slots 16 t + 16 of rgb_fc.2's feed), the last one behind the layer.  The loads then have the rest of rgb_fc.2, the softmax and the blend (and, for the
  // later tiles, the first k-groups of the next rgb_fc.0) to land, instead of standing in front of the first MFMA of their own tile.
  f32x16 x[4];
  auto tile_of = [&](long u_) { return VSEG == 0 ? u_ * NW + wave : u_; };
  auto load_x = [&](long u_, int t) DYN_INLINE_LAMBDA {
    const long tl = tile_of(u_);
    const bool ok = u_ < n_units_of(p, RAG, VSEG, NW) && tl < p.n_tiles_a;
    const float4* xw = reinterpret_cast<const float4*>(p.ws + p.o.off_x) + (ok ? tl : 0) * 16 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = ok ? nt_load4<8>(xw + (t * 4 + q) * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
      x[t][q * 4] = v.x; x[t][q * 4 + 1] = v.y; x[t][q * 4 + 2] = v.z; x[t][q * 4 + 3] = v.w;
    }
  };
  // lane-segment flavour: every wave walks its own tiles; dense flavour: the workgroup walks blocks of NW tiles together (its reductions use barriers);
  // ragged dense flavour: the blocks are the plan's workgroups (the row tables of the view kernel that parked x)
  const long n_units = n_units_of(p, RAG, VSEG, NW);
  const long first = VSEG == 0 ? blockIdx.x : (long)blockIdx.x * NW + wave, step = VSEG == 0 ? gridDim.x : (long)gridDim.x * NW;
  if (DYN_BLEND_PREFETCH) {
#pragma unroll
    for (int t = 0; t < 4; ++t) load_x(first, t);
  }
  RaggedHead head_next = {0, 0, 0, 0};  // (ragged rows: this thread's table entry and the scalars of the NEXT unit, requested one unit ahead)
  if (RAG && first < n_units) head_next = ragged_head(p.rg_rowtab, p.rg_ptab, first, dr.rw);
  for (long u = first; u < n_units; u += step) {
    if (RAG) {
      const RaggedHead head = head_next;
      if (u + step < n_units) head_next = ragged_head(p.rg_rowtab, p.rg_ptab, u + step, dr.rw);
      dr = ragged_rows(V, p.rg_rowtab, p.rg_ptab, u, ctab + SC_CT, nullptr, &head);  // (no task loops here: the row offsets stay in registers)
    } else if (VSEG == 0) {
      dr.point0 = u * p.PT;
    }
    const int PT = RAG ? dr.PTW : p.PT;
    const int p_local = VSEG == 0 ? dr.p_local : j / (VSEG == 0 ? 1 : VSEG);
    const int view = VSEG == 0 ? dr.view : (j & (VSEG - 1));
    const long tile = VSEG == 0 ? u * NW + wave : u;
    const bool tile_ok = tile < p.n_tiles_a;
    const long point = VSEG == 0 ? dr.point0 + p_local : tile * PT + p_local;
    const bool valid = (VSEG == 0 ? p_local < PT : view < V) && (point < p.n_pts);
    const long pv = valid ? point * V + view : 0;
    float msk = valid ? p.mask[pv] : 0.f;
    const float4 rd = valid ? reinterpret_cast<const float4*>(p.ray_diff)[pv] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 vrec = tile_ok ? reinterpret_cast<const float4*>(p.ws + p.o.off_vis)[tile * 32 + j] : make_float4(0.f, 0.f, 0.f, 0.f);  // {vis2, r, g, b} of the row
    const float vis2 = vrec.x;
    const float rgb_in[3] = {valid ? vrec.y : 0.f, valid ? vrec.z : 0.f, valid ? vrec.w : 0.f};
    if (p.mask_rgb && !((rgb_in[0] + rgb_in[1]) + rgb_in[2] > 1e-3f)) msk = 0.f;  // mask = mask * rgb_mask also feeds the masked_fill (mlp_network.py:458-460, 523)
    f32x16 a[4];
    {
      const float4* hg = reinterpret_cast<const float4*>(p.ws + p.o.off_hg) + (valid ? point_rec(p, point, h, SB_HG_RECS) : 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (!DYN_BLEND_PREFETCH) load_x(u, t);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = valid ? hg[(t * 4 + q) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
          a[t][q * 4] = b.x; a[t][q * 4 + 1] = b.y; a[t][q * 4 + 2] = b.z; a[t][q * 4 + 3] = b.w;
        }
      }
      const float extra[3] = {h == 0 ? vis2 : rd.x, h == 0 ? rd.y : rd.z, h == 0 ? rd.w : 0.f};
      mlp_layer_b6_lds<4, SC_L11_STEPS>(w11, a, [&](int s) { return s < 64 ? x[s / 16][s % 16] : extra[s - 64]; });
    }
    f32x16 b2[2];
    acc_init_bias<2>(b2, ctab + 80);
    mlp_layer_b6_lds<2, 64>(w12, b2, [&](int s) {
      const float r = elu_s(a[s / 16][s % 16]);
      if (DYN_BLEND_PREFETCH && (s & 15) == 15) load_x(u + step, s / 16);  // tile s / 16 of `a` has just been read for the last time: its registers take the next x
      return r;
    });
    acc_elu_s(b2);
    float logit = row_dot<2>(b2, ctab) + ctab[64];
    if (msk == 0.f) logit = -1e9f;
    if (VSEG == 0 ? p_local >= PT : view >= V) logit = -3.0e38f;  // padding rows take no share even when every real view is masked (uniform 1/V then)
    const float mx = views_max<VSEG>(dr, logit);
    const float e = (VSEG == 0 && p_local >= PT) ? 0.f : __expf(logit - mx);
    const float bw = e / views_sum<VSEG>(dr, e);
    const bool writer = valid && (VSEG == 0 ? dr.rw == dr.base : view == 0) && h == 0;  // the point's first row
    // Ragged rows: a point whose views are ALL masked by the projection keeps one placeholder row; the reference's softmax over V equal logits of -1e9 gives
    // every view 1 / V, its colour is the plain mean of the V gathered colours (mlp_network.py:523-526): taken from rgb_feat here, the rows that held them are gone
    // (the same holds when mask_rgb has masked every remaining row: `nvalid` is the view kernel's sum of the final masks)
    const bool all_masked = RAG && writer && p.ws[p.o.off_nvalid + point] == 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = views_sum<VSEG>(dr, rgb_in[c] * bw);
      if (all_masked) {
        const float wv = 1.0f / (float)V;
        v = 0.f;
        for (int k = 0; k < V; ++k) v += p.rgb_feat[(point * V + k) * 35 + c] * wv;
      }
      if (writer) p.raw[point * 4 + c] = v;
    }
  }
}
template <int VSEG>
__global__ void __launch_bounds__(DYN_BLEND_WS_THREADS, 1) k_static_blend_ws(StaticArgs p) { static_blend_ws_body<VSEG, DYN_BLEND_WS_THREADS>(p); }
template <bool RAG>
__global__ void __launch_bounds__(DYN_VIEW_THREADS, 1) k_static_blend_dense_ws(StaticArgs p) { static_blend_ws_body<0, DYN_VIEW_THREADS, RAG>(p); }
#endif

template <int VSEG>
__global__ void __launch_bounds__(DYN_BLEND_THREADS, 3) k_static_blend(StaticArgs p) { static_blend_body<VSEG, DYN_BLEND_THREADS>(p); }
__global__ void __launch_bounds__(DYN_VIEW_THREADS, 1) k_static_blend_dense(StaticArgs p) { static_blend_body<0, DYN_VIEW_THREADS>(p); }


// -------------------------------------------------------------------------------------------------------------------
// the plan of a ragged launch (see StaticWs): fills a.rg_* from the workspace and runs the four planning kernels on `stream`
static int ragged_plan(StaticArgs& a, const float* s_abs, hipStream_t stream) {
  unsigned* bits = reinterpret_cast<unsigned*>(a.ws + a.o.off_bits);
  float* emin = a.ws + a.o.off_emin;
  int* seg_cnt = reinterpret_cast<int*>(a.ws + a.o.off_segcnt);
  int* seg_start = reinterpret_cast<int*>(a.ws + a.o.off_segstart);
  int* wg = reinterpret_cast<int*>(a.ws + a.o.off_wgstart);
  unsigned short* rowtab = reinterpret_cast<unsigned short*>(a.ws + a.o.off_rowtab);
  int* ptab = reinterpret_cast<int*>(a.ws + a.o.off_ptab);
  a.rg_bits = bits; a.rg_emin = emin; a.rg_wg = wg; a.rg_rowtab = rowtab; a.rg_ptab = ptab;
  DYN_REQUIRE(a.n_pts < (1L << 31), "ragged plan: R * S must stay below 2^31 points");
  DYN_LAUNCH(DYN_K_STATIC_PLAN, "k_ragged_points", k_ragged_points, dim3(dyn_cdiv(a.n_pts, 256)), dim3(256), (size_t)256 * a.V * 5, stream, a.n_pts, a.V, a.mask,
             reinterpret_cast<const float4*>(a.ray_diff), s_abs, bits, emin);
  DYN_LAUNCH(DYN_K_STATIC_PLAN, "k_ragged_segments", k_ragged_segments, dim3((unsigned)a.o.n_seg), dim3(256), (RAG_SEG + 4 + 256) * sizeof(int) + RAG_SEG * sizeof(unsigned short), stream, a.n_pts, a.o.n_seg, bits, seg_cnt, seg_start);
  DYN_LAUNCH(DYN_K_STATIC_PLAN, "k_ragged_scatter", k_ragged_scatter, dim3(1), dim3(1024), 1028 * sizeof(int), stream, a.n_pts, a.o.n_seg, seg_cnt, seg_start, wg);
  DYN_LAUNCH(DYN_K_STATIC_PLAN, "k_ragged_tables", k_ragged_tables, dim3((unsigned)a.o.n_wg_max), dim3(64), 0, stream, bits, wg, rowtab, ptab);
  return 0;
}

extern "C" int dyn_static_net(const DynStaticNetParams* q, void* stream_) {
  DYN_REQUIRE(q, "dyn_static_net: null params");
  DYN_REQUIRE(q->R > 0 && q->S > 0 && q->V > 0, "dyn_static_net: empty problem");
  DYN_REQUIRE(q->V <= 32, "dyn_static_net: at most 32 source views");
  DYN_REQUIRE(q->S <= DYN_MAX_SAMPLES, "dyn_static_net: at most %d samples per ray", DYN_MAX_SAMPLES);
  DYN_REQUIRE(q->blob && q->ray_o && q->ray_d && q->pts && q->rgb_feat && q->ray_diff && q->mask && q->centers && q->raw && q->workspace,
              "dyn_static_net: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  StaticArgs a;
  a.R = q->R; a.S = q->S; a.V = q->V;
  a.o = static_ws(q->R, q->S, q->V);
  DYN_REQUIRE(q->workspace_bytes >= a.o.total * sizeof(float), "dyn_static_net: workspace too small (%zu < %zu bytes)", q->workspace_bytes,
              a.o.total * sizeof(float));
  a.PT = a.o.PT; a.TPR = a.o.TPR;
  a.anti_alias = q->anti_alias_pooling; a.mask_rgb = q->mask_rgb;
  a.n_pts = a.o.n_pts; a.n_tiles_a = a.o.n_tiles_a; a.n_tiles_b = a.o.n_tiles_b;
  a.blob = q->blob; a.pts = q->pts; a.rgb_feat = q->rgb_feat; a.ray_diff = q->ray_diff; a.mask = q->mask; a.centers = q->centers;
  a.raw = q->raw; a.ws = (float*)q->workspace;
  a.rg_bits = nullptr; a.rg_emin = nullptr; a.rg_wg = nullptr; a.rg_rowtab = nullptr; a.rg_ptab = nullptr;

  DYN_LAUNCH(DYN_K_STATIC_REF, "k_static_ref_feat", k_static_ref_feat, dim3(dyn_cdiv(q->R, REF_RAYS)), dim3(256), REF_RAYS * 66 * sizeof(float), stream, q->ray_o,
             q->ray_d, q->blob + ST_OFF_REF, q->R, a.ws + a.o.off_ref);
  const dim3 grid_a(dyn_cdiv(a.n_tiles_a, DYN_VIEW_THREADS / 64)), grid_b(dyn_cdiv(a.n_tiles_b, 4)), blk(DYN_NET_THREADS), blk_v(DYN_VIEW_THREADS);
  const size_t lds_a = (2 * NET_CHUNK + SA_CT + POOL_FLOATS(SA_NX) + RES_FLOATS) * sizeof(float);
  const size_t lds_b = (PTS_RING_SLOTS * PTS_CHUNK + SB_CT + SB_KL_FLOATS + SB_VL_FLOATS) * sizeof(float);
  const size_t lds_c = (NET_CHUNK + SC_CT) * sizeof(float);
  const dim3 grid_c(dyn_cdiv(a.n_tiles_a, DYN_BLEND_THREADS / 64)), blk_c(DYN_BLEND_THREADS);
  const size_t lds_rag = (DENSE_EXTRA + RAG_PTAB) * sizeof(float);
  if (a.o.ragged) {
    const int rc = ragged_plan(a, a.anti_alias ? a.blob + ST_OFF_CTA + 258 : nullptr, stream);
    if (rc != 0) return rc;
  }
  if (a.o.ragged && a.V == 11) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", (k_static_views<0, 11, true>), grid_a, blk_v, lds_a + lds_rag, stream, a);
  else if (a.o.ragged) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", (k_static_views<0, 0, true>), grid_a, blk_v, lds_a + lds_rag, stream, a);
  else if (a.o.dense && a.V == 11) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", (k_static_views<0, 11>), grid_a, blk_v, lds_a + DENSE_EXTRA * sizeof(float), stream, a);
  else if (a.o.dense) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", k_static_views<0>, grid_a, blk_v, lds_a + DENSE_EXTRA * sizeof(float), stream, a);
  else if (q->V <= 4) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", k_static_views<4>, grid_a, blk_v, lds_a, stream, a);
  else if (q->V <= 8) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", k_static_views<8>, grid_a, blk_v, lds_a, stream, a);
  else if (q->V <= 16) DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", k_static_views<16>, grid_a, blk_v, lds_a, stream, a);
  else DYN_LAUNCH(DYN_K_STATIC_VIEWS, "k_static_views", k_static_views<32>, grid_a, blk_v, lds_a, stream, a);
  if (a.TPR <= 4) {
    DYN_LAUNCH(DYN_K_STATIC_POINTS, "k_static_points", (k_net_points<false, 0>), points_grid(grid_b), blk, lds_b, stream, a);
  } else {
    DYN_LAUNCH(DYN_K_STATIC_POINTS_QKV, "k_static_points_qkv", (k_net_points<false, 1>), grid_b, blk, lds_b, stream, a);
    DYN_LAUNCH(DYN_K_STATIC_POINTS, "k_static_points", (k_net_points<false, 2>), grid_b, blk, lds_b, stream, a);
  }
#if DYN_BLEND_WS
  static const int blend_stream = getenv("DYN_BLEND_STREAM") != nullptr;  // developer A/B: the streaming (round-3) form
  if (!blend_stream) {
    const size_t lds_w = (SC_WS_FLOATS + SC_CT + DENSE_SCALARS + RAG_PTAB) * sizeof(float);  // (+ padding: the fixed-trip reductions read up to V - 1 entries past a point's rows)
    const unsigned n_cu = (unsigned)dyn_cu_count();
    if (a.o.dense) {
      const unsigned nb = (unsigned)dyn_cdiv(a.n_tiles_a, DYN_VIEW_THREADS / 64);
      if (a.o.ragged) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_dense_ws<true>, dim3(nb < n_cu ? nb : n_cu), blk_v, lds_w, stream, a);
      else DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_dense_ws<false>, dim3(nb < n_cu ? nb : n_cu), blk_v, lds_w, stream, a);
    } else {
      const unsigned nb = (unsigned)dyn_cdiv(a.n_tiles_a, DYN_BLEND_WS_THREADS / 64);
      const dim3 gw(nb < n_cu ? nb : n_cu), bw_(DYN_BLEND_WS_THREADS);
      if (q->V <= 4) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_ws<4>, gw, bw_, lds_w, stream, a);
      else if (q->V <= 8) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_ws<8>, gw, bw_, lds_w, stream, a);
      else if (q->V <= 16) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_ws<16>, gw, bw_, lds_w, stream, a);
      else DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_ws<32>, gw, bw_, lds_w, stream, a);
    }
    return 0;
  }
#endif
  if (a.o.dense) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend_dense, grid_a, blk_v, lds_c + DENSE_SCALARS * sizeof(float), stream, a);
  else if (q->V <= 4) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend<4>, grid_c, blk_c, lds_c, stream, a);
  else if (q->V <= 8) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend<8>, grid_c, blk_c, lds_c, stream, a);
  else if (q->V <= 16) DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend<16>, grid_c, blk_c, lds_c, stream, a);
  else DYN_LAUNCH(DYN_K_STATIC_BLEND, "k_static_blend", k_static_blend<32>, grid_c, blk_c, lds_c, stream, a);
  return 0;
}

// ===================================================================================================================
// DynibarDynamic: packing
// ===================================================================================================================
extern "C" size_t dyn_dynamic_net_blob_floats(void) { return DY_BLOB_FLOATS; }

extern "C" int dyn_dynamic_net_pack(const float* const* T, int F, float* blob, size_t blob_floats) {
  DYN_REQUIRE(T && blob, "dyn_dynamic_net_pack: null pointer");
  DYN_REQUIRE(F == 32, "dyn_dynamic_net_pack: the kernels are specialised for 32 feature channels");
  DYN_REQUIRE(blob_floats >= DY_BLOB_FLOATS, "dyn_dynamic_net_pack: blob too small");
  for (int i = 0; i < DT_NUM_TENSORS; ++i) DYN_REQUIRE(T[i] != nullptr, "dyn_dynamic_net_pack: tensor %d is NULL", i);
  std::vector<float> o;
  g_pack_range_error = false;
  o.reserve(DY_BLOB_FLOATS);
  {
    const float *W = T[DT_BASE0_W], *b = T[DT_BASE0_B];  // input [mean | var | x]  (mlp_network.py:262-266)
    pack_net_layer(o, 8, DA_L3P_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int c = da_c35(s % DA_NX, h);
      return c < 0 ? 0.f : W[(32 * t + i) * 105 + (s / DA_NX) * 35 + c];
    }, DYN_ELU_PRE));
    pack_net_layer(o, 8, DA_L3V_STEPS, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s == DA_NX) return h == 0 ? b[n] : 0.f;
      const int c = da_c35(s, h);
      return c < 0 ? 0.f : W[n * 105 + 70 + c];
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 4, SA_L4_STEPS, scaled(chained(T[DT_BASE2_W], nullptr, 128, 256, 256), DYN_ELU_POST));
  pack_net_layer(o, 4, SA_L5_STEPS, scaled(chained(T[DT_VIS0_W], nullptr, 128, 128, 128), DYN_ELU_PRE));
  pack_net_layer(o, 4, SA_L5_STEPS, scaled(chained(T[DT_VIS2_W], nullptr, 128, 128, 128), DYN_ELU_POST));
  pack_net_layer(o, 4, SA_L5_STEPS, scaled(chained(T[DT_VISB0_W], nullptr, 128, 128, 128), DYN_ELU_PRE));
  DYN_REQUIRE(o.size() == DY_OFF_B, "dynamic pack: A stream size mismatch");
#if DYN_POINTS_DUO
  g_pack_chunk_pairs = PTS_CP;
#endif
  {
    const float *W = T[DT_GEO0_W], *b = T[DT_GEO0_B];
    pack_net_layer(o, 8, 129, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < 128) return W[n * 257 + (s < 64 ? 0 : 128) + chain_feature(s % 64, h)];
      return h == 0 ? W[n * 257 + 256] : b[n];
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 4, 129, scaled_wb(chained(T[DT_GEO2_W], T[DT_GEO2_B], 128, 256, 256), 128, DYN_ELU_POST, 1.0));
  pack_net_layer(o, 4, 64, chained(T[DT_WQ], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 64, chained(T[DT_WK], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 64, chained(T[DT_WV], nullptr, 128, 128, 128));
  pack_net_layer(o, 4, 64, chained(T[DT_FC], nullptr, 128, 128, 128));
  {
    const float *W = T[DT_REFPTS0_W], *b = T[DT_REFPTS0_B];  // [256, 128 + 33]
    pack_net_layer(o, 8, 81, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < 64) return W[n * 161 + chain_feature(s, h)];
      if (s < 79) { const int c = (s - 64) / 5, f = (s - 64) % 5; return W[n * 161 + 128 + 3 + (h * 5 + f) * 3 + c]; }
      if (s == 79) return W[n * 161 + 128 + h];
      return h == 0 ? W[n * 161 + 128 + 2] : b[n];
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 4, 129, scaled_wb(chained(T[DT_REFPTS2_W], T[DT_REFPTS2_B], 128, 256, 256), 128, DYN_ELU_POST, 1.0));
  pack_net_layer(o, 4, 65, scaled(chained(T[DT_OG0_W], T[DT_OG0_B], 128, 128, 128), DYN_ELU_PRE));
  {
    const float *W = T[DT_RGB0_W], *b = T[DT_RGB0_B];  // [128, 128 + 27]
    pack_net_layer(o, 4, 78, scaled([=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < 64) return W[n * 155 + chain_feature(s, h)];
      if (s < 76) { const int c = (s - 64) / 4, f = (s - 64) % 4; return W[n * 155 + 128 + 3 + (h * 4 + f) * 3 + c]; }
      if (s == 76) return W[n * 155 + 128 + h];
      return h == 0 ? W[n * 155 + 128 + 2] : b[n];
    }, DYN_ELU_PRE));
  }
  pack_net_layer(o, 2, 65, scaled_wb(chained(T[DT_RGB2_W], T[DT_RGB2_B], 64, 128, 128), 64, DYN_ELU_POST * DYN_ELU_PRE, DYN_ELU_PRE));
  g_pack_chunk_pairs = B6_CHUNK_PAIRS;
  DYN_REQUIRE(o.size() == DY_OFF_CTA, "dynamic pack: B stream size mismatch");
  pack_rowtab(o, T[DT_VIS2_W] + 128 * 128, 128, DYN_ELU_POST);
  pack_rowtab(o, T[DT_VISB2_W], 128, DYN_ELU_POST);
  o.push_back(T[DT_VIS2_B][128]);
  o.push_back(T[DT_VISB2_B][0]);
  o.resize(DY_OFF_CTA + 272, 0.f);
  pack_rowtab(o, T[DT_BASE2_B], 128);
  pack_rowtab(o, T[DT_VIS0_B], 128, DYN_ELU_PRE);
  pack_rowtab(o, T[DT_VIS2_B], 128);
  pack_rowtab(o, T[DT_VISB0_B], 128, DYN_ELU_PRE);
  o.resize(DY_OFF_CTB, 0.f);
  pack_rowtab(o, T[DT_LN_G], 128);
  pack_rowtab(o, T[DT_LN_B], 128);
  pack_rowtab(o, T[DT_OG2_W], 128, DYN_ELU_POST);
  o.push_back(T[DT_OG2_B][0]);
  for (int c = 0; c < 3; ++c) o.push_back(T[DT_RGB4_B][c]);
  o.resize(DY_OFF_CTB + 400, 0.f);
  for (int c = 0; c < 3; ++c) pack_rowtab(o, T[DT_RGB4_W] + c * 64, 64, DYN_ELU_POST);
  o.resize(DY_OFF_POSENC, 0.f);
  // sinusoid table (mlp_network.py:218-234), evaluated in double like numpy, stored in D-layout order [pos][half][64]
  for (int pos = 0; pos < DYN_MAX_SAMPLES; ++pos)
    for (int h = 0; h < 2; ++h)
      for (int k = 0; k < 64; ++k) {
        const int f = chain_feature(k, h);
        const double ang = (double)pos / pow(10000.0, 2.0 * (f / 2) / 128.0);
        o.push_back((float)((f % 2 == 0) ? sin(ang) : cos(ang)));
      }
  for (int i = 0; i < 256 * 21; ++i) o.push_back(T[DT_RAYDIR0_W][i]);
  for (int i = 0; i < 256; ++i) o.push_back(T[DT_RAYDIR0_B][i]);
  for (int i = 0; i < 35 * 256; ++i) o.push_back(T[DT_RAYDIR2_W][i]);
  for (int i = 0; i < 35; ++i) o.push_back(T[DT_RAYDIR2_B][i]);
  o.resize(DY_BLOB_FLOATS, 0.f);
  DYN_REQUIRE(!g_pack_range_error, "dyn_dynamic_net_pack: a weight is outside the half-float range of the split engine (|w| >= 65504 or not finite)");
  for (size_t i = 0; i < DY_BLOB_FLOATS; ++i) blob[i] = o[i];
  return 0;
}

extern "C" size_t dyn_dynamic_net_workspace_bytes(int R, int S, int V) {
  if (R <= 0 || S <= 0 || V <= 0 || V > 32) return 0;
  return static_ws(R, S, V, true).total * sizeof(float);
}

// direction_feat = ray_dir_fc(PE(time))  (mlp_network.py:240-247): the same 35-vector for every ray, sample and view
__global__ void __launch_bounds__(256) k_dynamic_time_feat(const float* __restrict__ Wt, const float* __restrict__ time, float* __restrict__ out) {
  float* hid = reinterpret_cast<float*>(dyn_smem);  // [256]
  const int tid = threadIdx.x;
  const float t = time[0];
  float pe[21];
  pe[0] = t;
  for (int f = 0; f < 10; ++f) {
    const float a = (float)(1 << f) * t;
    pe[1 + f] = cosf(a);
    pe[11 + f] = sinf(a);
  }
  {
    const float* w = Wt + tid * 21;
    float acc = Wt[256 * 21 + tid];
    for (int k = 0; k < 21; ++k) acc = fmaf(w[k], pe[k], acc);
    hid[tid] = acc > 0.f ? acc : expm1f(acc);
  }
  __syncthreads();
  if (tid < 35) {
    const float* w = Wt + 256 * 21 + 256 + tid * 256;
    float acc = Wt[256 * 21 + 256 + 35 * 256 + tid];
    for (int k = 0; k < 256; ++k) acc = fmaf(w[k], hid[k], acc);
    out[tid] = acc > 0.f ? acc : expm1f(acc);
  } else if (tid < 36) {
    out[tid] = 0.f;
  }
}

// per point-view chain of the dynamic net: (rgb_feat + direction_feat) -> mean/var (mask weights) -> base_fc -> shared tail
template <int VSEG, bool RAG = false>
__global__ void __launch_bounds__(DYN_VIEW_THREADS, 2) k_dynamic_views(StaticArgs p) {
  static_assert(!RAG || VSEG == 0, "ragged rows are a dense-rows flavour");
  if (RAG && (int)blockIdx.x >= p.rg_wg[0]) return;  // the grid is the plan's upper bound
  float* lds = reinterpret_cast<float*>(dyn_smem);
  float* ctab = lds + 2 * NET_CHUNK;  // [SA_CT]
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  for (int i = tid; i < SA_CT; i += DYN_VIEW_THREADS) ctab[i] = p.blob[DY_OFF_CTA + i];
  NetRing ring;
  constexpr int LDS_FLOATS = 2 * NET_CHUNK + SA_CT + POOL_FLOATS(SA_NX) + RES_FLOATS + (VSEG == 0 ? DENSE_EXTRA : 0);
  net_ring_init(ring, p.blob + DY_OFF_A, DA_CHUNKS, lds, 0, (VSEG >= 8 || VSEG == 0) ? net_layer_chunks(8, DA_L3P_STEPS) : 0, DYN_VIEW_THREADS);

  const int V = p.V;
  const long tile = (long)blockIdx.x * (DYN_VIEW_THREADS / 64) + wave;
  // row -> (point, view): power-of-two lane segments (VSEG > 0), dense rows (VSEG == 0) or ragged dense rows (RAG), as in k_static_views
  const DenseRows dr = RAG ? ragged_rows(V, p.rg_rowtab, p.rg_ptab, blockIdx.x, lds + LDS_FLOATS - DENSE_SCALARS, reinterpret_cast<int*>(lds + LDS_FLOATS))
                           : dense_rows(V, p.PT, lds + LDS_FLOATS - DENSE_SCALARS);
  const int PT = RAG ? dr.PTW : p.PT;
  const int p_local = VSEG == 0 ? dr.p_local : j / (VSEG == 0 ? 1 : VSEG);
  const int view = VSEG == 0 ? dr.view : (j & (VSEG - 1));
  const long point = VSEG == 0 ? dr.point0 + p_local : tile * PT + p_local;
  const bool valid = (VSEG == 0 ? p_local < PT : view < V) && (point < p.n_pts);
  const int seg_base = 0;
  const long pv = valid ? point * V + view : 0;
  const float msk = valid ? p.mask[pv] : 0.f;
  const float* tf = p.ws + p.o.off_ref;
  float xin[DA_NX];
#pragma unroll
  for (int q = 0; q < DA_NX; ++q) {
    const int ch = h == 0 ? q : 18 + q;
    xin[q] = (valid && ch < 35) ? nt_load1<2>(p.rgb_feat + pv * 35 + ch) + tf[ch] : 0.f;
  }
  const float wraw = views_sum<VSEG>(dr, msk);
  const float wgt = msk / (wraw + 1e-8f);
  f32x16 a1[8];
  base_fc0<VSEG, DA_NX>(ring, p.blob + DY_OFF_A, xin, wgt, V, view, p_local, ctab + SA_CT, a1, &dr, wraw / (wraw + 1e-8f));
  views_tail<VSEG, false>(ring, a1, p, ctab, wgt, msk, tile, point, valid, view, seg_base, &dr, lds);
}

extern "C" int dyn_dynamic_net(const DynDynamicNetParams* q, void* stream_) {
  DYN_REQUIRE(q, "dyn_dynamic_net: null params");
  DYN_REQUIRE(q->R > 0 && q->S > 0 && q->V > 0, "dyn_dynamic_net: empty problem");
  DYN_REQUIRE(q->V <= 32, "dyn_dynamic_net: at most 32 source views");
  DYN_REQUIRE(q->S <= DYN_MAX_SAMPLES, "dyn_dynamic_net: at most %d samples per ray", DYN_MAX_SAMPLES);
  DYN_REQUIRE(q->blob && q->ray_d && q->pts && q->rgb_feat && q->mask && q->time && q->raw && q->workspace, "dyn_dynamic_net: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  StaticArgs a;
  a.R = q->R; a.S = q->S; a.V = q->V;
  a.o = static_ws(q->R, q->S, q->V, true);
  DYN_REQUIRE(q->workspace_bytes >= a.o.total * sizeof(float), "dyn_dynamic_net: workspace too small (%zu < %zu bytes)", q->workspace_bytes,
              a.o.total * sizeof(float));
  a.PT = a.o.PT; a.TPR = a.o.TPR;
  a.anti_alias = 0; a.mask_rgb = 0;
  a.n_pts = a.o.n_pts; a.n_tiles_a = a.o.n_tiles_a; a.n_tiles_b = a.o.n_tiles_b;
  a.shift = q->shift;
  a.blob = q->blob; a.ray_d = q->ray_d; a.pts = q->pts; a.rgb_feat = q->rgb_feat; a.ray_diff = nullptr; a.mask = q->mask; a.centers = nullptr;
  a.raw = q->raw; a.ws = (float*)q->workspace;
  a.rg_bits = nullptr; a.rg_emin = nullptr; a.rg_wg = nullptr; a.rg_rowtab = nullptr; a.rg_ptab = nullptr;
  DYN_LAUNCH(DYN_K_DYNAMIC_TIME, "k_dynamic_time_feat", k_dynamic_time_feat, dim3(1), dim3(256), 256 * sizeof(float), stream,
             q->blob + DY_OFF_TIME, q->time, a.ws + a.o.off_ref);
  const dim3 grid_a(dyn_cdiv(a.n_tiles_a, DYN_VIEW_THREADS / 64)), grid_b(dyn_cdiv(a.n_tiles_b, 4)), blk(DYN_NET_THREADS), blk_v(DYN_VIEW_THREADS);
  const size_t lds_a = (2 * NET_CHUNK + SA_CT + POOL_FLOATS(SA_NX) + RES_FLOATS) * sizeof(float);
  const size_t lds_b = (PTS_RING_SLOTS * PTS_CHUNK + DB_CT + SB_KL_FLOATS + SB_VL_FLOATS) * sizeof(float);
  if (a.o.ragged) {
    const int rc = ragged_plan(a, nullptr, stream);
    if (rc != 0) return rc;
    DYN_LAUNCH(DYN_K_DYNAMIC_VIEWS, "k_dynamic_views", (k_dynamic_views<0, true>), grid_a, blk_v, lds_a + (DENSE_EXTRA + RAG_PTAB) * sizeof(float), stream, a);
  } else if (a.o.dense) DYN_LAUNCH(DYN_K_DYNAMIC_VIEWS, "k_dynamic_views", k_dynamic_views<0>, grid_a, blk_v, lds_a + DENSE_EXTRA * sizeof(float), stream, a);
  else if (q->V <= 4) DYN_LAUNCH(DYN_K_DYNAMIC_VIEWS, "k_dynamic_views", k_dynamic_views<4>, grid_a, blk_v, lds_a, stream, a);
  else if (q->V <= 8) DYN_LAUNCH(DYN_K_DYNAMIC_VIEWS, "k_dynamic_views", k_dynamic_views<8>, grid_a, blk_v, lds_a, stream, a);
  else if (q->V <= 16) DYN_LAUNCH(DYN_K_DYNAMIC_VIEWS, "k_dynamic_views", k_dynamic_views<16>, grid_a, blk_v, lds_a, stream, a);
  else DYN_LAUNCH(DYN_K_DYNAMIC_VIEWS, "k_dynamic_views", k_dynamic_views<32>, grid_a, blk_v, lds_a, stream, a);
  if (a.TPR <= 4) {
    DYN_LAUNCH(DYN_K_DYNAMIC_POINTS, "k_dynamic_points", (k_net_points<true, 0>), points_grid(grid_b), blk, lds_b, stream, a);
  } else {
    DYN_LAUNCH(DYN_K_DYNAMIC_POINTS_QKV, "k_dynamic_points_qkv", (k_net_points<true, 1>), grid_b, blk, lds_b, stream, a);
    DYN_LAUNCH(DYN_K_DYNAMIC_POINTS, "k_dynamic_points", (k_net_points<true, 2>), grid_b, blk, lds_b, stream, a);
  }
  return 0;
}

// ===================================================================================================================
// MotionMLP (mlp_network.py:558-618): 8 x 256 ReLU MLP over PE(x, y, z, t) with a skip into layer 5, 3B DCT coefficients out
// one wave = 32 sample points; the 132 Fourier features are recomputed for the skip instead of being kept in registers
// ===================================================================================================================
enum { MT_L0_W, MT_L0_B, MT_L1_W, MT_L1_B, MT_L2_W, MT_L2_B, MT_L3_W, MT_L3_B, MT_L4_W, MT_L4_B, MT_L5_W, MT_L5_B, MT_L6_W, MT_L6_B, MT_L7_W,
       MT_L7_B, MT_COEFF_W, MT_COEFF_B, MT_NUM_TENSORS };
#define MO_PE_STEPS 66  /* 64 cos|sin pairs (4 coords x 16 frequencies), (x, y), (z, t) */
constexpr int MO_CHUNKS = net_layer_chunks(8, MO_PE_STEPS + 1) + 4 * net_layer_chunks(8, 129) + net_layer_chunks(8, MO_PE_STEPS + 129) +
                          2 * net_layer_chunks(8, 129) + net_layer_chunks(1, 129);
constexpr size_t MO_OFF_FREQ = (size_t)MO_CHUNKS * NET_CHUNK;  // the 16 frequencies of torch.linspace(1, 17, 16)
constexpr size_t MO_BLOB_FLOATS = MO_OFF_FREQ + 16;

extern "C" size_t dyn_motion_mlp_blob_floats(void) { return MO_BLOB_FLOATS; }

// reference column of the 132-wide embedding for PE k-step s (< MO_PE_STEPS), half h
static int mo_pe_col(int s, int h) {
  if (s < 64) { const int c = s / 16, f = s % 16; return 4 + (h * 16 + f) * 4 + c; }
  return (s - 64) * 2 + h;
}

extern "C" int dyn_motion_mlp_pack(const float* const* T, int num_basis, float* blob, size_t blob_floats) {
  DYN_REQUIRE(T && blob, "dyn_motion_mlp_pack: null pointer");
  DYN_REQUIRE(num_basis >= 1 && 3 * num_basis <= 32, "dyn_motion_mlp_pack: 3 * num_basis must be at most 32");
  DYN_REQUIRE(blob_floats >= MO_BLOB_FLOATS, "dyn_motion_mlp_pack: blob too small");
  for (int i = 0; i < MT_NUM_TENSORS; ++i) DYN_REQUIRE(T[i] != nullptr, "dyn_motion_mlp_pack: tensor %d is NULL", i);
  std::vector<float> o;
  g_pack_range_error = false;
  o.reserve(MO_BLOB_FLOATS);
  {
    const float *W = T[MT_L0_W], *b = T[MT_L0_B];
    pack_net_layer(o, 8, MO_PE_STEPS + 1, [=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < MO_PE_STEPS) return W[n * 132 + mo_pe_col(s, h)];
      return h == 0 ? b[n] : 0.f;
    });
  }
  for (int l = 1; l <= 4; ++l) pack_net_layer(o, 8, 129, chained(T[MT_L0_W + 2 * l], T[MT_L0_B + 2 * l], 256, 256, 256));
  {
    const float *W = T[MT_L5_W], *b = T[MT_L5_B];  // input = cat([embedding(132), h(256)])
    pack_net_layer(o, 8, MO_PE_STEPS + 129, [=](int t, int i, int s, int h) -> float {
      const int n = 32 * t + i;
      if (s < MO_PE_STEPS) return W[n * 388 + mo_pe_col(s, h)];
      const int s2 = s - MO_PE_STEPS;
      if (s2 < 128) return W[n * 388 + 132 + chain_feature(s2, h)];
      return h == 0 ? b[n] : 0.f;
    });
  }
  pack_net_layer(o, 8, 129, chained(T[MT_L6_W], T[MT_L6_B], 256, 256, 256));
  pack_net_layer(o, 8, 129, chained(T[MT_L7_W], T[MT_L7_B], 256, 256, 256));
  pack_net_layer(o, 1, 129, chained(T[MT_COEFF_W], T[MT_COEFF_B], 3 * num_basis, 256, 256));
  DYN_REQUIRE(o.size() == MO_OFF_FREQ, "motion pack: stream size mismatch");
  {
    // torch.linspace(1, 17, 16) in fp32: start + i * step below the midpoint, end - (n - 1 - i) * step above it
    const float step = (17.0f - 1.0f) / 15.0f;
    for (int i = 0; i < 16; ++i) o.push_back(i < 8 ? 1.0f + (float)i * step : 17.0f - (float)(15 - i) * step);
  }
  DYN_REQUIRE(!g_pack_range_error, "dyn_motion_mlp_pack: a weight is outside the half-float range of the split engine (|w| >= 65504 or not finite)");
  for (size_t i = 0; i < MO_BLOB_FLOATS; ++i) blob[i] = o[i];
  return 0;
}

// sin | cos of an argument beyond the two-constant reduction's reach (|x| > 1e6: never a scene coordinate times a frequency <= 17, but the
// reference's torch.sin / torch.cos take any float).  By VALUE: the library's sincosf(x, &s, &c) writes through pointers, and with it in the
// loop the 66-entry embedding lived in scratch memory -- every read of it a scratch load behind `s_waitcnt vmcnt(0)`, which also drained
// the weight ring's DMA queue (round 4).
__device__ __attribute__((noinline)) float sin_or_cos_huge(float x, int want_sin) { return want_sin ? sinf(x) : cosf(x); }

__device__ __forceinline__ void motion_embed(const float (&c4)[4], const float* __restrict__ freq, int h, float (&pe)[MO_PE_STEPS]) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      float sn, cs;
      const float arg = freq[f] * c4[c];
      sincos_small(arg, sn, cs);
      float v = h == 0 ? cs : sn;
      if (__builtin_expect(fabsf(arg) > 1.0e6f, 0)) v = sin_or_cos_huge(arg, h);
      pe[c * 16 + f] = v;
    }
  pe[64] = h == 0 ? c4[0] : c4[1];
  pe[65] = h == 0 ? c4[2] : c4[3];
}

// DYN_MOTION_DUO = 1 (default): the interleaved layer loop on the three-slot ring (dyn_mlp.h, round 4); 0: the round-3 form (A/B builds)
#ifndef DYN_MOTION_DUO
#define DYN_MOTION_DUO 1
#endif
#if DYN_MOTION_DUO
typedef WeightRing3 MotionRing;
#define MOTION_RING_SLOTS B6D_SLOTS
#define motion_ring_init(R, stream, total, lds) ring3_init(R, stream, total, lds, DYN_NET_THREADS)
#define motion_layer mlp_layer_b6_duo
#else
typedef NetRing MotionRing;
#define MOTION_RING_SLOTS 2
#define motion_ring_init(R, stream, total, lds) net_ring_init_t(R, stream, total, lds, DYN_NET_THREADS)
#define motion_layer net_layer
#endif

// The last n_zero_last samples of every ray carry no motion: raw_coeff[:, -n_last:, :] *= 0 (render_ray.py:684).  Rounds 1-4 evaluated the chain for them
// and multiplied by zero; since round 5 the launch covers only the S - n_zero_last samples per ray that keep their coefficients (a tenth fewer rows at the
// reference's n_last = round(0.1 S)) and k_motion_zero_tail writes the zeros.  (+0 where c x 0 gave -0 for a negative c: every consumer adds or multiplies
// them into sums that are equal either way; a NON-FINITE chain output no longer turns into NaN there.)
__global__ void __launch_bounds__(256) k_motion_zero_tail(long R, int S, int n_zero_last, int n_out, float* __restrict__ coeff) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_ray = (long)n_zero_last * n_out;
  if (i >= R * per_ray) return;
  const long ray = i / per_ray;
  coeff[(ray * S + (S - n_zero_last)) * n_out + (i - ray * per_ray)] = 0.f;
}

__global__ void __launch_bounds__(DYN_NET_THREADS, 1)
k_motion_mlp(const float* __restrict__ blob, const float* __restrict__ pts, const float* __restrict__ time, long n_kept, int S, int n_zero_last,
             int n_out, float inv_div, float* __restrict__ coeff) {
  DYN_CLAIM_REGISTER_FILE();
  float* lds = reinterpret_cast<float*>(dyn_smem);
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
  MotionRing ring;
  motion_ring_init(ring, blob, MO_CHUNKS, lds);
  // row -> (ray, sample) over the samples that keep their coefficients
  const long row = ((long)blockIdx.x * 4 + wave) * 32 + j;
  const bool valid = row < n_kept;
  const int Sk = S - n_zero_last;
  const long ray = (valid ? row : 0) / Sk;
  const long point = ray * S + ((valid ? row : 0) - ray * Sk);
  float c4[4] = {0.f, 0.f, 0.f, time[0]};
  if (valid) { c4[0] = pts[point * 3]; c4[1] = pts[point * 3 + 1]; c4[2] = pts[point * 3 + 2]; }
  const float* freq = blob + MO_OFF_FREQ;
  const float one_h0 = h == 0 ? 1.0f : 0.0f;
  f32x16 a[8], b[8];
  // the embedding feeds layer 0 and the skip layer: evaluated once and kept (one wave per SIMD: 512 registers per lane to spend)
  float pe[MO_PE_STEPS];
  motion_embed(c4, freq, h, pe);
  // the ReLU of a layer rides in its consumer's feed (two instructions per value in the gaps between MFMAs instead of a 256-instruction
  // sweep over the accumulators between the layers)
  acc_zero(a);
  motion_layer<8, MO_PE_STEPS + 1>(ring, a, [&](int s) { return s < MO_PE_STEPS ? pe[s] : one_h0; });
  acc_zero(b);
  motion_layer<8, 129>(ring, b, [&](int s) { return s < 128 ? relu1(a[s / 16][s % 16]) : one_h0; });
  acc_zero(a);
  motion_layer<8, 129>(ring, a, [&](int s) { return s < 128 ? relu1(b[s / 16][s % 16]) : one_h0; });
  acc_zero(b);
  motion_layer<8, 129>(ring, b, [&](int s) { return s < 128 ? relu1(a[s / 16][s % 16]) : one_h0; });
  acc_zero(a);
  motion_layer<8, 129>(ring, a, [&](int s) { return s < 128 ? relu1(b[s / 16][s % 16]) : one_h0; });
  acc_zero(b);
  motion_layer<8, MO_PE_STEPS + 129>(ring, b, [&](int s) {
    if (s < MO_PE_STEPS) return pe[s];
    return s - MO_PE_STEPS < 128 ? relu1(a[(s - MO_PE_STEPS) / 16][(s - MO_PE_STEPS) % 16]) : one_h0;
  });
  acc_zero(a);
  motion_layer<8, 129>(ring, a, [&](int s) { return s < 128 ? relu1(b[s / 16][s % 16]) : one_h0; });
  acc_zero(b);
  motion_layer<8, 129>(ring, b, [&](int s) { return s < 128 ? relu1(a[s / 16][s % 16]) : one_h0; });
  f32x16 c1[1];
  acc_zero(c1);
  motion_layer<1, 129>(ring, c1, [&](int s) { return s < 128 ? relu1(b[s / 16][s % 16]) : one_h0; });
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = dyn_fi(r, h);
      if (f < n_out) coeff[point * n_out + f] = c1[0][r] * inv_div;
    }
  }
}

extern "C" int dyn_motion_mlp(const float* blob, const float* pts, const float* time, int R, int S, int num_basis, int n_zero_last,
                              float sf_mag_div, float* coeff, void* stream) {
  DYN_REQUIRE(blob && pts && time && coeff, "dyn_motion_mlp: null pointer");
  DYN_REQUIRE(R > 0 && S > 0 && num_basis >= 1 && 3 * num_basis <= 32 && n_zero_last >= 0 && sf_mag_div != 0.f, "dyn_motion_mlp: bad argument");
  const int nz = n_zero_last < S ? n_zero_last : S;
  const long n_kept = (long)R * (S - nz);
  if (n_kept > 0)
    DYN_LAUNCH(DYN_K_MOTION_MLP, "k_motion_mlp", k_motion_mlp, dim3(dyn_cdiv(n_kept, 128)), dim3(DYN_NET_THREADS), MOTION_RING_SLOTS * NET_CHUNK * sizeof(float),
               (hipStream_t)stream, blob, pts, time, n_kept, S, nz, 3 * num_basis, 1.0f / sf_mag_div, coeff);
  if (nz > 0)
    DYN_LAUNCH(DYN_K_MOTION_TAIL, "k_motion_zero_tail", k_motion_zero_tail, dim3(dyn_cdiv((long)R * nz * 3 * num_basis, 256)), dim3(256), 0, (hipStream_t)stream, (long)R, S, nz,
               3 * num_basis, coeff);
  return 0;
}

extern "C" int dyn_mlp_split_terms(void) {
  return DYN_SPLIT_TERMS;
}
extern "C" int dyn_mlp_split_kind(void) {  // 0: native fp32 MFMA, 1: bf16 parts, 2: half-float parts
#if DYN_SPLIT_F16
  return 2;
#else
  return 1;
#endif
}

// ===================================================================================================================
// engine self-test: y = elu(W x + b) for one Linear 64 -> 64, through pack_layer / the weight ring / mlp_layer
// ===================================================================================================================
__global__ void __launch_bounds__(DYN_NET_THREADS, 2) k_selftest(const float* __restrict__ stream, const float* __restrict__ x, float* __restrict__ y,
                                                                  int rows) {
  float* lds = reinterpret_cast<float*>(dyn_smem);
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  NetRing ring;
  net_ring_init_t(ring, stream, 2 * net_layer_chunks(2, 33), lds, DYN_NET_THREADS);
  const int row = (blockIdx.x * 4 + wave) * 32 + j;
  f32x16 in[2];
#pragma unroll
  for (int s = 0; s < 32; ++s) in[s / 16][s % 16] = row < rows ? x[row * 64 + 32 * (s / 16) + dyn_fi(s % 16, h)] : 0.f;
  const float one_h0 = h == 0 ? 1.0f : 0.0f;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {  // two chained applications of the same layer
    f32x16 acc[2];
    acc_zero(acc);
    net_layer<2, 33>(ring, acc, [&](int s) { return s < 32 ? in[s / 16][s % 16] : one_h0; });
    acc_elu(acc);
    in[0] = acc[0];
    in[1] = acc[1];
  }
  if (row < rows)
#pragma unroll
    for (int s = 0; s < 32; ++s) y[row * 64 + 32 * (s / 16) + dyn_fi(s % 16, h)] = in[s / 16][s % 16];
}

extern "C" int dyn_mlp_selftest(const float* W, const float* b, const float* x, float* y, int rows, float* stream_buf, void* stream) {
  // W [64,64], b [64]: HOST; x [rows,64], y [rows,64], stream_buf [2 * chunks * 4096]: DEVICE (stream_buf is filled here via hipMemcpy)
  DYN_REQUIRE(W && b && x && y && stream_buf && rows > 0, "dyn_mlp_selftest: bad argument");
  std::vector<float> o;
  pack_net_layer(o, 2, 33, chained(W, b, 64, 64, 64));
  pack_net_layer(o, 2, 33, chained(W, b, 64, 64, 64));
  if (hipMemcpy(stream_buf, o.data(), o.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    dyn_set_error("dyn_mlp_selftest: hipMemcpy failed");
    return DYN_E_LAUNCH;
  }
  DYN_LAUNCH(DYN_K_SELFTEST, "k_selftest", k_selftest, dim3(dyn_cdiv(rows, 128)), dim3(DYN_NET_THREADS), 2 * NET_CHUNK * sizeof(float),
             (hipStream_t)stream, stream_buf, x, y, rows);
  return 0;
}
