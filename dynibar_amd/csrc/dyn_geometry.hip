// Geometry / sampling / compositing kernels (HBM-bound byte & index work).  Built with -ffp-contract=off so that
// the elementwise fp32 arithmetic rounds exactly like the reference's un-fused ATen ops: depth samples, sample
// points, normalised pixel locations and the inverse-CDF indices are then bit-identical to the CPU oracle wherever the
// inputs are.
#include <stdarg.h>
#include <stdlib.h>

#include "dyn_device.h"
#include "dyn_host.h"

static thread_local char g_err[512] = "";
void dyn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dyn_last_error(void) { return g_err; }
extern "C" int dyn_abi_version(void) { return DYN_ABI_VERSION; }


// ---------------------------------------------------------------------------------------------------------------
// per-kernel timing
// ---------------------------------------------------------------------------------------------------------------
#define DYN_PROF_RING 512
static struct {
  int on;
  hipEvent_t ev[DYN_K_COUNT][DYN_PROF_RING][2];
  int made[DYN_K_COUNT];   // events created so far per slot
  int used[DYN_K_COUNT];   // records pending since the last read
  double ms[DYN_K_COUNT];  // carried over when the ring wraps
  long n[DYN_K_COUNT];
} g_prof;
static const char* const g_prof_names[DYN_K_COUNT] = {
    "k_prepare_cameras", "k_nchw_to_nhwc", "k_sample_along_ray", "k_points_from_z", "k_project_gather", "k_sample_mask", "k_composite",
    "k_fine_samples", "k_static_ref_feat", "k_static_views", "k_static_points", "k_static_blend", "k_selftest", "k_dynamic_time_feat",
    "k_dynamic_views", "k_dynamic_points", "k_motion_mlp", "k_trajectory_points", "k_render_flows", "k_expected_scene_flow", "k_image_rays",
    "k_static_points_qkv", "k_dynamic_points_qkv", "k_enc_conv7", "k_enc_conv3", "k_enc_conv1", "k_enc_block_out",
    "k_train_gemm", "k_train_rows", "k_train_attn", "k_gather_bwd", "k_motion_zero_tail", "k_ragged_plan"};

static void prof_flush(int slot) {
  for (int i = 0; i < g_prof.used[slot]; ++i) {
    float ms = 0.f;
    (void)hipEventSynchronize(g_prof.ev[slot][i][1]);
    if (hipEventElapsedTime(&ms, g_prof.ev[slot][i][0], g_prof.ev[slot][i][1]) == hipSuccess) g_prof.ms[slot] += ms;
    g_prof.n[slot] += 1;
  }
  g_prof.used[slot] = 0;
}
void dyn_prof_begin(int slot, hipStream_t stream) {
  if (!g_prof.on) return;
  if (g_prof.used[slot] == DYN_PROF_RING) prof_flush(slot);
  const int i = g_prof.used[slot];
  if (i >= g_prof.made[slot]) {
    (void)hipEventCreate(&g_prof.ev[slot][i][0]);
    (void)hipEventCreate(&g_prof.ev[slot][i][1]);
    g_prof.made[slot] = i + 1;
  }
  (void)hipEventRecord(g_prof.ev[slot][i][0], stream);
}
void dyn_prof_end(int slot, hipStream_t stream) {
  if (!g_prof.on) return;
  (void)hipEventRecord(g_prof.ev[slot][g_prof.used[slot]][1], stream);
  g_prof.used[slot] += 1;
}
extern "C" int dyn_profile_enable(int on) {
  g_prof.on = on ? 1 : 0;
  return 0;
}
extern "C" int dyn_profile_count(void) { return DYN_K_COUNT; }
extern "C" const char* dyn_profile_name(int slot) { return (slot >= 0 && slot < DYN_K_COUNT) ? g_prof_names[slot] : ""; }
extern "C" int dyn_profile_read(float* total_ms, int* launches) {
  DYN_REQUIRE(total_ms && launches, "dyn_profile_read: null pointer");
  for (int s = 0; s < DYN_K_COUNT; ++s) {
    prof_flush(s);
    total_ms[s] = (float)g_prof.ms[s];
    launches[s] = (int)g_prof.n[s];
    g_prof.ms[s] = 0.0;
    g_prof.n[s] = 0;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// camera preparation: P = K . inv(c2w)   (projection.py:42-47 does torch.inverse + bmm on [V,4,4])
// ---------------------------------------------------------------------------------------------------------------
__device__ void invert4x4(const double* m, double* inv) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = m[i * 4 + j];
      a[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    double best = fabs(a[c][c]);
    for (int r = c + 1; r < 4; ++r)
      if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
    if (piv != c)
      for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    double d = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= d;
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        double f = a[r][c];
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
      }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
}

__global__ void k_prepare_cameras(const float* __restrict__ cams, int V, const float* __restrict__ query_cam,
                                  float* __restrict__ proj, float* __restrict__ query_center) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < V) {
    const float* c = cams + (long)v * 34;
    double K[16], P[16], W2C[16];
    for (int i = 0; i < 16; ++i) { K[i] = c[2 + i]; P[i] = c[18 + i]; }
    invert4x4(P, W2C);
    float* o = proj + (long)v * 16;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += K[i * 4 + k] * W2C[k * 4 + j];
        o[i * 4 + j] = (float)s;
      }
    o[12] = c[18 + 3];
    o[13] = c[18 + 7];
    o[14] = c[18 + 11];
    o[15] = 0.f;
  } else if (v == V && query_cam != nullptr) {
    query_center[0] = query_cam[18 + 3];
    query_center[1] = query_cam[18 + 7];
    query_center[2] = query_cam[18 + 11];
    query_center[3] = 0.f;
  }
}

extern "C" int dyn_prepare_cameras(const float* cams, int V, const float* query_cam, float* proj, float* query_center,
                                   void* stream) {
  DYN_REQUIRE(cams && proj && V > 0, "dyn_prepare_cameras: null pointer or V<=0");
  DYN_REQUIRE(query_cam == nullptr || query_center != nullptr, "dyn_prepare_cameras: query_center missing");
  DYN_LAUNCH(DYN_K_PREPARE_CAMERAS, "dyn_prepare_cameras", k_prepare_cameras, dim3(dyn_cdiv(V + 1, 64)), dim3(64), 0, (hipStream_t)stream, cams, V, query_cam, proj,
                     query_center);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// NCHW -> NHWC repack of the feature maps (once per target view; 1.2 MB per view at 72x128x32)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int F, int HW) {
  // block = 64 pixels x all channels through an LDS tile (pad 1) so both sides are coalesced
  float* tile = reinterpret_cast<float*>(dyn_smem);  // [F][65]
  const int v = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const float* s = src + (long)v * F * HW;
  float* d = dst + (long)v * F * HW;
  for (int i = tid; i < F * 64; i += blockDim.x) {
    int c = i >> 6, p = i & 63;
    tile[c * 65 + p] = (p0 + p < HW) ? s[(long)c * HW + p0 + p] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < F * 64; i += blockDim.x) {
    int p = i / F, c = i % F;
    if (p0 + p < HW) d[(long)(p0 + p) * F + c] = tile[c * 65 + p];
  }
}

extern "C" int dyn_nchw_to_nhwc(const float* src, float* dst, int V, int F, int Hf, int Wf, void* stream) {
  DYN_REQUIRE(src && dst && V > 0 && F > 0 && Hf > 0 && Wf > 0, "dyn_nchw_to_nhwc: bad argument");
  DYN_REQUIRE((size_t)F * 65 * 4 <= 150 * 1024, "dyn_nchw_to_nhwc: F too large");
  int HW = Hf * Wf;
  DYN_LAUNCH(DYN_K_NCHW_TO_NHWC, "dyn_nchw_to_nhwc", k_nchw_to_nhwc, dim3(dyn_cdiv(HW, 64), V), dim3(256), (size_t)F * 65 * 4, (hipStream_t)stream, src, dst, F, HW);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// a5 depth sampling  (render_ray.py:67-131)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float coarse_depth(int i, int S, float near, float far, int inv_uniform) {
  if (inv_uniform) {
    float start = 1.0f / near;
    float step = (1.0f / far - start) / (float)(S - 1);
    return 1.0f / (start + (float)i * step);
  }
  float step = (far - near) / (float)(S - 1);
  return near + (float)i * step;
}

__global__ void k_sample_along_ray(DynSampleParams p) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.R * p.S) return;
  int r = (int)(idx / p.S), i = (int)(idx % p.S);
  float near = p.depth_range[0], far = p.depth_range[1];
  float z = coarse_depth(i, p.S, near, far, p.inv_uniform);
  if (p.t_rand != nullptr) {
    float zm = coarse_depth(i > 0 ? i - 1 : 0, p.S, near, far, p.inv_uniform);
    float zp = coarse_depth(i < p.S - 1 ? i + 1 : i, p.S, near, far, p.inv_uniform);
    float lower = (i == 0) ? z : 0.5f * (z + zm);
    float upper = (i == p.S - 1) ? z : 0.5f * (zp + z);
    z = lower + (upper - lower) * p.t_rand[idx];
  }
  p.z_vals[idx] = z;
  if (p.s_vals != nullptr) p.s_vals[idx] = ((1.0f / z) - (1.0f / near)) / (1.0f / far - 1.0f / near);
  if (p.pts != nullptr) {
    for (int c = 0; c < 3; ++c) p.pts[idx * 3 + c] = z * p.ray_d[r * 3 + c] + p.ray_o[r * 3 + c];
  }
}

extern "C" int dyn_sample_along_ray(const DynSampleParams* p, void* stream) {
  DYN_REQUIRE(p && p->ray_o && p->ray_d && p->depth_range && p->z_vals, "dyn_sample_along_ray: null pointer");
  DYN_REQUIRE(p->R > 0 && p->S > 1, "dyn_sample_along_ray: need R>0, S>1");
  long n = (long)p->R * p->S;
  DYN_LAUNCH(DYN_K_SAMPLE, "dyn_sample_along_ray", k_sample_along_ray, dim3(dyn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, *p);
  return 0;
}

__global__ void k_points_from_z(const float* __restrict__ ray_o, const float* __restrict__ ray_d, const float* __restrict__ z_vals,
                                const float* __restrict__ depth_range, long n, int S, float* __restrict__ pts, float* __restrict__ s_vals) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  int r = (int)(idx / S);
  float z = z_vals[idx];
  if (pts != nullptr)
    for (int c = 0; c < 3; ++c) pts[idx * 3 + c] = z * ray_d[r * 3 + c] + ray_o[r * 3 + c];
  if (s_vals != nullptr) {
    float near = depth_range[0], far = depth_range[1];
    s_vals[idx] = ((1.0f / z) - (1.0f / near)) / (1.0f / far - 1.0f / near);
  }
}

extern "C" int dyn_points_from_z(const float* ray_o, const float* ray_d, const float* z_vals, const float* depth_range, int R, int S,
                                 float* pts, float* s_vals, void* stream) {
  DYN_REQUIRE(ray_o && ray_d && z_vals && R > 0 && S > 0, "dyn_points_from_z: bad argument");
  DYN_REQUIRE(s_vals == nullptr || depth_range != nullptr, "dyn_points_from_z: depth_range missing");
  long n = (long)R * S;
  DYN_LAUNCH(DYN_K_POINTS_FROM_Z, "dyn_points_from_z", k_points_from_z, dim3(dyn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ray_o, ray_d, z_vals, depth_range, n, S, pts,
                     s_vals);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// K1: fused projection + bilinear gather + ray_diff + mask   (projection.py:103-176)
//
// HBM-bound: 160 B written per point-view, every source map read once.  No LDS and no barriers, so the
// occupancy is set by registers alone and each wave keeps many taps in flight:
// one wavefront = 64 consecutive point-view rows g = (r*S + s)*V + v.
//   phase 1: lane = row.  Project the point, write ray_diff (one dwordx4 per lane, 1 KiB contiguous per wave) and the mask,
//            blend and write the three RGB taps, keep the feature-tap origin/fractions in registers.
//   phase 2: lane = (row-in-group, 16-byte channel group).  F/4 lanes cover one 128-byte channels-last tap line; the tap
//            descriptors come from the owning lane by cross-lane reads.  All four taps of PG_UNROLL row-groups are issued
//            before the first blend.  Each lane stores its 16 bytes straight into the row (rows are 140 B, so the store is only
//            4-byte aligned: the wave still covers whole contiguous spans).
// Workgroup -> rows mapping is XCD-aware: workgroup b runs on XCD b % 8, and XCD x is given the x-th contiguous eighth of the
// rows, so neighbouring rays (overlapping epipolar footprints) share one L2.
// ---------------------------------------------------------------------------------------------------------------
#ifndef PG_THREADS
#define PG_THREADS 64
#endif
#ifndef PG_UNROLL
#define PG_UNROLL 4
#endif
#ifndef PG_OCC
#define PG_OCC 4
#endif
#ifndef PG_STAGE
#define PG_STAGE 1  /* assemble each wave's 64 output rows in LDS and store them as aligned, fully coalesced dwordx4 */
#endif
#ifdef DYN_PHASE_TIMING  /* developer instrumentation: cycle stamps of one wave of two workgroups (tools/phasebench.py) */
__device__ unsigned long long g_pg_phase[2][8];
#define PG_PHASE(i)                                                                                          \
  do {                                                                                                       \
    if (threadIdx.x == 0 && (blockIdx.x == 8 || blockIdx.x == gridDim.x / 2)) g_pg_phase[blockIdx.x != 8][i] = __builtin_readcyclecounter(); \
  } while (0)
extern "C" int dyn_debug_pg_phases(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pg_phase), sizeof(g_pg_phase)) == hipSuccess ? 0 : 1;
}
#else
#define PG_PHASE(i)
#endif
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));

__device__ __forceinline__ float safe_floor_coord(float x, float size) {
  // taps further than one pixel outside contribute zero either way; the clamp keeps the int conversion defined
  return fminf(fmaxf(x, -2.0f), size + 1.0f);
}

// DYN_NORMALIZE_IEEE = 1 (measured, round 3): the gather kernel 88 -> 94.5 us (0.51 -> 0.475 of 8 TB/s) for three square roots and nine divisions per
// point-view, and the result still is not bitwise ATen's (its norm reduces in another order); ray_diff is checked to 1e-6 either way.  The perspective
// divide and normalize(), which decide the bilinear taps and make the gathered values bit-exact, are IEEE at no measurable cost.
#ifndef DYN_NORMALIZE_IEEE
#define DYN_NORMALIZE_IEEE 0
#endif
__device__ __forceinline__ void normalize3(float x, float y, float z, float& ox, float& oy, float& oz) {
#if DYN_NORMALIZE_IEEE
  // F.normalize(eps=1e-12) as written (projection.py:83-97): v / max(sqrt(x^2 + y^2 + z^2), eps) with a correctly rounded square root and IEEE
  // divisions (this unit is built without contraction, so the sum of squares rounds like ATen's)
  const float d = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
  ox = x / d;
  oy = y / d;
  oz = z / d;
#else
  const float inv = fminf(rsqrtf(x * x + y * y + z * z), 1e12f);  // one reciprocal square root (1 ulp) and three multiplies
  ox = x * inv;
  oy = y * inv;
  oz = z * inv;
#endif
}

// exact n / d for 32-bit n with the host-made multiplier m = ceil(2^32 / d): the estimate is q or q + 1
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned m) {
  if (d == 1) return n;
  unsigned q = __umulhi(n, m);
  return q - ((unsigned long long)q * d > n ? 1u : 0u);
}

struct PGShape {
  int R, S, V, H, W, Hf, Wf, F;
  float img_h, img_w, inv_wm1, inv_hm1;
  unsigned mV, mS;  // fast_div multipliers
  long N, ntask, tasks_per_xcd;
};

// bilinear tap set of one row on a [Hm, Wm] map: clamped (always loadable) corner coordinates and the four weights, zeroed for
// corners outside the map (zeros padding).  ix, iy: un-normalised align_corners=True coordinates.
__device__ __forceinline__ int iclamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
struct Taps {
  int x0, x1, y0, y1;
  float w_nw, w_ne, w_sw, w_se;
};
__device__ __forceinline__ Taps make_taps(float nx, float ny, int Wm, int Hm) {
  const float ix = safe_floor_coord((nx + 1.0f) * ((float)(Wm - 1) / 2.0f), (float)Wm);
  const float iy = safe_floor_coord((ny + 1.0f) * ((float)(Hm - 1) / 2.0f), (float)Hm);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float fx = ix - fx0, fy = iy - fy0;
  const float e = 1.0f - fx, s = 1.0f - fy;
  const bool x0ok = (x0 >= 0) && (x0 < Wm), x1ok = (x0 + 1 >= 0) && (x0 + 1 < Wm);
  const bool y0ok = (y0 >= 0) && (y0 < Hm), y1ok = (y0 + 1 >= 0) && (y0 + 1 < Hm);
  Taps t;
  t.x0 = iclamp(x0, Wm - 1); t.x1 = iclamp(x0 + 1, Wm - 1);
  t.y0 = iclamp(y0, Hm - 1); t.y1 = iclamp(y0 + 1, Hm - 1);
  t.w_nw = (x0ok && y0ok) ? s * e : 0.f;
  t.w_ne = (x1ok && y0ok) ? s * fx : 0.f;
  t.w_sw = (x0ok && y1ok) ? fy * e : 0.f;
  t.w_se = (x1ok && y1ok) ? fy * fx : 0.f;
  return t;
}

// ---- compute_traj_pts fused into the consumers of the displaced points (render_ray.py:361-369, :691-725) ----------------------------------------------
// A point seen at the time of basis row `row` is p + (traj(row) - traj(ref)), traj(row)[a] = sum_b c[a B + b] basis[row, b] (row < 0: the undisplaced point).
// ONE function for k_trajectory_points (which materialises [V,R,S,3] for callers of the reference's helper and for training) and for the gather kernels'
// fused form (the array never exists): the same sums in the same order, so the fused gather equals the gather on the materialised points bit for bit.
// (The flows of the fused path use the linearity of the expected point instead -- k_render_flows_traj.)
struct PGTraj {
  const float* coeff;  // [R,S,3 B] or nullptr
  const float* basis;  // [frames, B]
  const int* rows;     // device [V]
  int B, ref;
};
__device__ __forceinline__ void traj_displace(const float* __restrict__ c, const float* __restrict__ basis, int B, int row, int ref, float& x, float& y, float& z) {
  if (row < 0) return;
  float d[3];
  for (int a = 0; a < 3; ++a) {
    float t0 = 0.f, s = 0.f;
    for (int b = 0; b < B; ++b) t0 += c[a * B + b] * basis[(long)ref * B + b];
    for (int b = 0; b < B; ++b) s += c[a * B + b] * basis[(long)row * B + b];
    d[a] = s - t0;
  }
  x = x + d[0]; y = y + d[1]; z = z + d[2];
}

__global__ void __launch_bounds__(PG_THREADS, PG_OCC)
k_project_gather(PGShape q, const float* __restrict__ ray_o, const float* __restrict__ ray_d, const float* __restrict__ z_vals,
                 const float* __restrict__ pts_st, const float* __restrict__ xyz, const float4* __restrict__ proj4,
                 const float* __restrict__ query_center, const float* __restrict__ src_rgb, const float4* __restrict__ feat4,
                 float* __restrict__ rgb_feat, float4* __restrict__ ray_diff, float* __restrict__ mask, PGTraj tj) {
  const int lane = dyn_lane();
  // XCD-aware task id: the (b / 8)-th workgroup of XCD (b % 8) takes tasks from that XCD's contiguous range
  const long wg = blockIdx.x;
  const long local = (wg >> 3) * (PG_THREADS / 64) + dyn_wave();
  const long task = (wg & 7) * q.tasks_per_xcd + local;
  const int C = 3 + q.F;
#if PG_STAGE
  float* tile = reinterpret_cast<float*>(dyn_smem) + dyn_wave() * (64 * C);  // this wave's [64][C] output rows
  const bool live = local < q.tasks_per_xcd && task < q.ntask;
  if (!live) {  // keep the workgroup barrier count uniform
    __syncthreads();
    return;
  }
#else
  if (local >= q.tasks_per_xcd || task >= q.ntask) return;  // whole wave leaves; no barriers below
#endif
  PG_PHASE(0);
  const long g0 = task * 64;
  const bool has = g0 + lane < q.N;
  const unsigned g = (unsigned)(has ? g0 + lane : q.N - 1);  // idle lanes shadow the last row (loads stay in bounds, stores are skipped)

  PG_PHASE(1);
  // ---- phase 1 ----
  const unsigned rs = fast_div(g, (unsigned)q.V, q.mV);
  const int v = (int)(g - rs * q.V);
  const unsigned r = fast_div(rs, (unsigned)q.S, q.mS);
  float sx, sy, sz;  // reference-time point (xyz_st)
  if (pts_st != nullptr) {
    sx = pts_st[rs * 3 + 0]; sy = pts_st[rs * 3 + 1]; sz = pts_st[rs * 3 + 2];
  } else {
    const float z = z_vals[rs];
    sx = z * ray_d[r * 3 + 0] + ray_o[r * 3 + 0];
    sy = z * ray_d[r * 3 + 1] + ray_o[r * 3 + 1];
    sz = z * ray_d[r * 3 + 2] + ray_o[r * 3 + 2];
  }
  float x = sx, y = sy, z3 = sz;  // per-view (motion displaced) point
  if (xyz != nullptr) {
    const long o = ((long)v * q.R * q.S + rs) * 3;
    x = xyz[o]; y = xyz[o + 1]; z3 = xyz[o + 2];
  } else if (tj.coeff != nullptr) {
    traj_displace(tj.coeff + (long)rs * 3 * tj.B, tj.basis, tj.B, tj.rows[v], tj.ref, x, y, z3);
  }
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2], P3 = proj4[v * 4 + 3];
  const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
  const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
  const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
  const float zc = fmaxf(hz, 1e-8f);
  float px = hx / zc, py = hy / zc;  // IEEE division, like the reference's tensor division (projection.py:53-55)
  px = fminf(fmaxf(px, -1e6f), 1e6f);
  py = fminf(fmaxf(py, -1e6f), 1e6f);
  const float wm1 = q.img_w - 1.0f, hm1 = q.img_h - 1.0f;
  const bool inb = (px <= wm1) && (px >= 0.f) && (py <= hm1) && (py >= 0.f);
  // normalize() then grid_sample's align_corners=True un-normalisation (ATen CPU: (x + 1) * ((size - 1) / 2))
  const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;  // normalize(): 2 * pixel / [w - 1, h - 1] - 1 (projection.py:22-30), divisions as written there
  const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
  const Taps tf = make_taps(nx, ny, q.Wf, q.Hf);
  {
    // RGB taps of this row: four unconditional 12-byte loads
    const Taps t = make_taps(nx, ny, q.W, q.H);
    const float* img = src_rgb + (long)v * q.H * q.W * 3;
    const f32x3u a = *reinterpret_cast<const f32x3u*>(img + (t.y0 * q.W + t.x0) * 3);
    const f32x3u b = *reinterpret_cast<const f32x3u*>(img + (t.y0 * q.W + t.x1) * 3);
    const f32x3u c = *reinterpret_cast<const f32x3u*>(img + (t.y1 * q.W + t.x0) * 3);
    const f32x3u d = *reinterpret_cast<const f32x3u*>(img + (t.y1 * q.W + t.x1) * 3);
    // compute_angle (projection.py:61-101), overlapping the tap latency
    float ax, ay, az, bx, by, bz, dx, dy, dz;
    normalize3(query_center[0] - sx, query_center[1] - sy, query_center[2] - sz, ax, ay, az);
    normalize3(P3.x - x, P3.y - y, P3.z - z3, bx, by, bz);
    normalize3(ax - bx, ay - by, az - bz, dx, dy, dz);
    if (has) {
      if (ray_diff != nullptr) nt_store4<1>(ray_diff + g, make_float4(dx, dy, dz, ax * bx + ay * by + az * bz));
      nt_store1<1>(mask + g, (inb && (hz > 0.f)) ? 1.0f : 0.0f);
      f32x3u o3;
      o3.x = fmaf(d.x, t.w_se, fmaf(c.x, t.w_sw, fmaf(b.x, t.w_ne, a.x * t.w_nw)));
      o3.y = fmaf(d.y, t.w_se, fmaf(c.y, t.w_sw, fmaf(b.y, t.w_ne, a.y * t.w_nw)));
      o3.z = fmaf(d.z, t.w_se, fmaf(c.z, t.w_sw, fmaf(b.z, t.w_ne, a.z * t.w_nw)));
#if PG_STAGE
      tile[lane * C] = o3.x; tile[lane * C + 1] = o3.y; tile[lane * C + 2] = o3.z;
#else
      *reinterpret_cast<f32x3u*>(rgb_feat + (long)g * C) = o3;
#endif
    }
  }
  PG_PHASE(2);
  // feature-tap descriptors of this row as float4-element offsets into feat4 (one 128-byte line = F4 elements)
  const int F4 = q.F >> 2;
  const int vbase = v * q.Hf * q.Wf;
  const int o_nw = (vbase + tf.y0 * q.Wf + tf.x0) * F4, o_ne = (vbase + tf.y0 * q.Wf + tf.x1) * F4;
  const int o_sw = (vbase + tf.y1 * q.Wf + tf.x0) * F4, o_se = (vbase + tf.y1 * q.Wf + tf.x1) * F4;

  // ---- phase 2: F4 lanes per row, RPI rows per pass ----
  const int RPI = 64 / F4;
  const int rsub = lane / F4, c4 = lane - rsub * F4;
  const bool lane_on = rsub < RPI;
  for (int it0 = 0; it0 * RPI < 64; it0 += PG_UNROLL) {
    float4 ta[PG_UNROLL], tb[PG_UNROLL], tc[PG_UNROLL], td[PG_UNROLL];
    float w0[PG_UNROLL], w1[PG_UNROLL], w2[PG_UNROLL], w3[PG_UNROLL];
#pragma unroll
    for (int u = 0; u < PG_UNROLL; ++u) {
      const int src = (it0 + u) * RPI + rsub < 63 ? (it0 + u) * RPI + rsub : 63;  // owning lane of the row (cross-lane reads are executed by every lane)
      ta[u] = feat4[__shfl(o_nw, src) + c4];
      tb[u] = feat4[__shfl(o_ne, src) + c4];
      tc[u] = feat4[__shfl(o_sw, src) + c4];
      td[u] = feat4[__shfl(o_se, src) + c4];
      w0[u] = __shfl(tf.w_nw, src); w1[u] = __shfl(tf.w_ne, src); w2[u] = __shfl(tf.w_sw, src); w3[u] = __shfl(tf.w_se, src);
    }
#pragma unroll
    for (int u = 0; u < PG_UNROLL; ++u) {
      f32x4u o;
      o.x = fmaf(td[u].x, w3[u], fmaf(tc[u].x, w2[u], fmaf(tb[u].x, w1[u], ta[u].x * w0[u])));
      o.y = fmaf(td[u].y, w3[u], fmaf(tc[u].y, w2[u], fmaf(tb[u].y, w1[u], ta[u].y * w0[u])));
      o.z = fmaf(td[u].z, w3[u], fmaf(tc[u].z, w2[u], fmaf(tb[u].z, w1[u], ta[u].z * w0[u])));
      o.w = fmaf(td[u].w, w3[u], fmaf(tc[u].w, w2[u], fmaf(tb[u].w, w1[u], ta[u].w * w0[u])));
      const int src = (it0 + u) * RPI + rsub;
#if PG_STAGE
      if (lane_on && src < 64) {
        float* t = tile + src * C + 3 + c4 * 4;
        t[0] = o.x; t[1] = o.y; t[2] = o.z; t[3] = o.w;
      }
#else
      if (lane_on && src < 64 && g0 + src < q.N) *reinterpret_cast<f32x4u*>(rgb_feat + (g0 + src) * C + 3 + c4 * 4) = o;
#endif
    }
  }
  PG_PHASE(3);
#if PG_STAGE
  __syncthreads();  // the wave's LDS writes are complete and visible (one barrier per workgroup; waves only share the barrier)
  {
    const long nrow = (q.N - g0 < 64) ? (q.N - g0) : 64;
    const int nflt = (int)nrow * C;
    float* dst = rgb_feat + g0 * C;  // g0 * C * 4 bytes = task * 64 * C * 4: 16-byte aligned for any C
    const float4* src4 = reinterpret_cast<const float4*>(tile);
    float4* dst4 = reinterpret_cast<float4*>(dst);
    // streaming output, never read again by this kernel: non-temporal stores keep the 4 MiB L2 slices for the source maps
    for (int i = lane; i < (nflt >> 2); i += 64) nt_store4<1>(dst4 + i, src4[i]);
    for (int i = (nflt & ~3) + lane; i < nflt; i += 64) nt_store1<1>(dst + i, tile[i]);
  }
#endif
#ifdef DYN_PHASE_TIMING
  __builtin_amdgcn_s_waitcnt(0x0F70);  // the stores have left
#endif
  PG_PHASE(4);
}


// ---------------------------------------------------------------------------------------------------------------
// K1, tile form (the shipped one): one workgroup = P consecutive sample points x all V views, one wavefront = 64 / P views.
//
// Why: in row order (point-major, view-minor) the 64 rows of a wave are 8 points x 8 views, i.e. eight different source maps per
// tap instruction and a working set of ~48 cache lines per wave: with 16 waves on a CU the 32 KiB vector L1 thrashes (PMC: the
// texture addresser is busy 84 % of the time, half of it waiting on pending misses).  Here a wave walks ONE epipolar line: its
// lanes are consecutive samples of a ray seen from one source view, so the eight rows of a tap instruction fall into one or two
// neighbouring bilinear cells (a few cache lines, mostly L1 hits) and the wave's working set is ~1 KiB.
// The output tile [P][V][3+F] is assembled in LDS in its final memory order and leaves as ONE contiguous, 16-byte aligned burst
// (P = 64, V = 8: 70 KiB); ray_diff and mask reuse the same LDS afterwards and leave the same way.  With all views of a point in
// one workgroup the per-point observation count (render_ray.py:736-741, the former k_sample_mask launch, which re-read the mask
// this kernel had just written) is a by-product.
// ---------------------------------------------------------------------------------------------------------------
#ifndef PGT_UNROLL
#define PGT_UNROLL 2  /* tap instructions in flight per pass (x 4 taps).  With P = 16, inside the pipeline: 1: 83.5 / 120.1 us at 8 / 11 views, 2: 82.2-83.1 / 115.2-115.6, 3: 85.3 / 119.6, 4 (rounds 2-4): 83.5-85.0 / 119.7-121.6, 8: spills */
#endif
#ifndef PGT_DEFAULT_P
#define PGT_DEFAULT_P 16
#endif
struct PGTile {
  int R, S, V, H, W, Hf, Wf, F;
  float img_h, img_w, inv_wm1, inv_hm1, mask_thresh;
  unsigned mS;          // fast_div multiplier for S
  long n_pts;           // R * S
  long ntile, tiles_per_xcd;
  int pref_wgs;         // workgroups that stream the source maps into the memory-side cache before their tile
};

template <int P>
__global__ void k_project_gather_tile(PGTile q, const float* __restrict__ ray_o, const float* __restrict__ ray_d, const float* __restrict__ z_vals,
                                      const float* __restrict__ pts_st, const float* __restrict__ xyz, const float4* __restrict__ proj4,
                                      const float* __restrict__ query_center, const float* __restrict__ src_rgb, const float4* __restrict__ feat4,
                                      float* __restrict__ rgb_feat, float4* __restrict__ ray_diff, float* __restrict__ mask,
                                      float* __restrict__ pix_mask, PGTraj tj) {
  constexpr int VPW = 64 / P;  // views per wave
  float* tile = reinterpret_cast<float*>(dyn_smem);
  const int lane = dyn_lane(), wave = dyn_wave();
  const int V = q.V, C = 3 + q.F;
#ifndef PGT_PREFETCH
#define PGT_PREFETCH 1
#endif
#if PGT_PREFETCH
  // In the rendering pipeline the source maps (24 MB at 8 views) have long left the memory-side cache when this kernel starts (the
  // network kernels stream gigabytes in between), and a gather whose every first touch of a cell goes to HBM runs 30 % slower
  // (measured: 122 us cold, 92 us warm).  The first workgroups therefore stream the maps once, coalesced, at full HBM rate, into the
  // Infinity Cache (~5 us chip-wide, overlapped with their own first taps); the values are not used.
  if (blockIdx.x < (unsigned)q.pref_wgs) {
    const long n4a = (long)V * q.Hf * q.Wf * (q.F >> 2);            // float4 elements of the feature maps
    const long n4b = ((long)V * q.H * q.W * 3) >> 2;                 // ... of the colour images (16-byte aligned base)
    const long per = (n4a + n4b + q.pref_wgs - 1) / q.pref_wgs;
    const long lo = (long)blockIdx.x * per, hi = lo + per < n4a + n4b ? lo + per : n4a + n4b;
    const float4* rgb4 = reinterpret_cast<const float4*>(src_rgb);
    float sink = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      const float4 t = i < n4a ? feat4[i] : rgb4[i - n4a];
      sink += t.x;
    }
    asm volatile("" ::"v"(sink));
  }
#endif
  // XCD-aware tile id: workgroup b runs on XCD b % 8; XCD x owns the x-th contiguous eighth of the tiles (neighbouring rays share an L2)
  const long wg = blockIdx.x;
  const long local = wg >> 3;
  const long t = (wg & 7) * q.tiles_per_xcd + local;
  const bool tile_live = local < q.tiles_per_xcd && t < q.ntile;
  const long p0 = t * P;                       // first point of the tile
  const int pl = lane & (P - 1);               // point within the tile
  const int v = wave * VPW + (lane / P);       // this lane's view
  const long pt = p0 + pl;
  const int vv = v < V ? v : V - 1;            // idle lanes shadow a real row: their loads stay in bounds, their results are dropped
  const unsigned rs = (unsigned)(tile_live ? (pt < q.n_pts ? pt : q.n_pts - 1) : 0);

  // the fused trajectory form: the tile's P x 3 B motion coefficients are one contiguous run -- read once, coalesced, and parked in LDS behind the tile's own
  // regions at an odd point stride (a lane reading its point's 18 values straight from global memory touches ten cache lines per load instruction: the gather of
  // the 7 dynamic views ran 11 % slower that way than on the materialised [V,R,S,3] array)
  const float* tj_c = nullptr;
  if (tj.coeff != nullptr) {
    // (every wave parks its own copy: its lanes are all P points x its views, and a wave needs no workgroup barrier for its own LDS words)
    const int C3 = 3 * tj.B, CS = C3 | 1;
    float* cst = tile + P * V * (C + 5) + wave * (P * CS);
    if (tile_live) {
      const long n_here = (q.n_pts - p0 < P ? q.n_pts - p0 : P) * C3;
      for (int e = lane; e < n_here; e += 64) cst[(e / C3) * CS + (e % C3)] = tj.coeff[p0 * C3 + e];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's LDS writes have landed (a wave runs in lock step)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    tj_c = cst + (pt < q.n_pts ? pl : 0) * CS;
  }
  // ---- phase 1: lane = (point, view) ----
  const unsigned r = fast_div(rs, (unsigned)q.S, q.mS);
  float sx, sy, sz;
  if (pts_st != nullptr) {
    sx = pts_st[rs * 3 + 0]; sy = pts_st[rs * 3 + 1]; sz = pts_st[rs * 3 + 2];
  } else {
    const float z = z_vals[rs];
    sx = z * ray_d[r * 3 + 0] + ray_o[r * 3 + 0];
    sy = z * ray_d[r * 3 + 1] + ray_o[r * 3 + 1];
    sz = z * ray_d[r * 3 + 2] + ray_o[r * 3 + 2];
  }
  float x = sx, y = sy, z3 = sz;
  if (xyz != nullptr) {
    const long o = ((long)vv * q.n_pts + rs) * 3;
    x = xyz[o]; y = xyz[o + 1]; z3 = xyz[o + 2];
  } else if (tj.coeff != nullptr) {
    // the displaced point is formed here from the point's 3 B motion coefficients (LDS: the lanes of a point's views read the same words) and two basis rows
    if (tile_live) traj_displace(tj_c, tj.basis, tj.B, tj.rows[vv], tj.ref, x, y, z3);
  }
  const float4 P0 = proj4[vv * 4], P1 = proj4[vv * 4 + 1], P2 = proj4[vv * 4 + 2], P3 = proj4[vv * 4 + 3];
  const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
  const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
  const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
  const float zc = fmaxf(hz, 1e-8f);
  float px = hx / zc, py = hy / zc;  // IEEE division, like the reference's tensor division (projection.py:53-55)
  px = fminf(fmaxf(px, -1e6f), 1e6f);
  py = fminf(fmaxf(py, -1e6f), 1e6f);
  const float wm1 = q.img_w - 1.0f, hm1 = q.img_h - 1.0f;
  const bool inb = (px <= wm1) && (px >= 0.f) && (py <= hm1) && (py >= 0.f);
  const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;  // normalize(): 2 * pixel / [w - 1, h - 1] - 1 (projection.py:22-30), divisions as written there
  const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
  const Taps tf = make_taps(nx, ny, q.Wf, q.Hf);
  const int row = pl * V + vv;  // row of the tile in its final order
  float4 rd;
  float mk;
  {
    const Taps tt = make_taps(nx, ny, q.W, q.H);
    const float* img = src_rgb + (long)vv * q.H * q.W * 3;
    const f32x3u a = *reinterpret_cast<const f32x3u*>(img + (tt.y0 * q.W + tt.x0) * 3);
    const f32x3u b = *reinterpret_cast<const f32x3u*>(img + (tt.y0 * q.W + tt.x1) * 3);
    const f32x3u c = *reinterpret_cast<const f32x3u*>(img + (tt.y1 * q.W + tt.x0) * 3);
    const f32x3u d = *reinterpret_cast<const f32x3u*>(img + (tt.y1 * q.W + tt.x1) * 3);
    float ax, ay, az, bx, by, bz, dx, dy, dz;
    normalize3(query_center[0] - sx, query_center[1] - sy, query_center[2] - sz, ax, ay, az);
    normalize3(P3.x - x, P3.y - y, P3.z - z3, bx, by, bz);
    normalize3(ax - bx, ay - by, az - bz, dx, dy, dz);
    rd = make_float4(dx, dy, dz, ax * bx + ay * by + az * bz);
    mk = (inb && (hz > 0.f)) ? 1.0f : 0.0f;
    if (v < V) {  // ray_diff / mask have their own LDS behind the feature tile (P V 20 bytes): written here, they leave with the tile behind the ONE barrier
      reinterpret_cast<float4*>(tile + P * V * C)[row] = rd;
      (tile + P * V * C + P * V * 4)[row] = (pt < q.n_pts) ? mk : 0.0f;
    }
    if (v < V) {
      tile[row * C + 0] = fmaf(d.x, tt.w_se, fmaf(c.x, tt.w_sw, fmaf(b.x, tt.w_ne, a.x * tt.w_nw)));
      tile[row * C + 1] = fmaf(d.y, tt.w_se, fmaf(c.y, tt.w_sw, fmaf(b.y, tt.w_ne, a.y * tt.w_nw)));
      tile[row * C + 2] = fmaf(d.z, tt.w_se, fmaf(c.z, tt.w_sw, fmaf(b.z, tt.w_ne, a.z * tt.w_nw)));
    }
  }
  const int F4 = q.F >> 2;
  const int vbase = vv * q.Hf * q.Wf;
  const int o_nw = (vbase + tf.y0 * q.Wf + tf.x0) * F4, o_ne = (vbase + tf.y0 * q.Wf + tf.x1) * F4;
  const int o_sw = (vbase + tf.y1 * q.Wf + tf.x0) * F4, o_se = (vbase + tf.y1 * q.Wf + tf.x1) * F4;

  // ---- phase 2: F4 lanes per row, RPI consecutive rows (= consecutive samples of one view) per tap instruction ----
  const int RPI = 64 / F4;
  const int rsub = lane / F4, c4 = lane - rsub * F4;
  const bool lane_on = rsub < RPI;
  for (int it0 = 0; it0 * RPI < 64; it0 += PGT_UNROLL) {
    float4 ta[PGT_UNROLL], tb[PGT_UNROLL], tc[PGT_UNROLL], td[PGT_UNROLL];
    float w0[PGT_UNROLL], w1[PGT_UNROLL], w2[PGT_UNROLL], w3[PGT_UNROLL];
#pragma unroll
    for (int u = 0; u < PGT_UNROLL; ++u) {
      const int src = (it0 + u) * RPI + rsub < 63 ? (it0 + u) * RPI + rsub : 63;
      ta[u] = feat4[__shfl(o_nw, src) + c4];
      tb[u] = feat4[__shfl(o_ne, src) + c4];
      tc[u] = feat4[__shfl(o_sw, src) + c4];
      td[u] = feat4[__shfl(o_se, src) + c4];
      w0[u] = __shfl(tf.w_nw, src); w1[u] = __shfl(tf.w_ne, src); w2[u] = __shfl(tf.w_sw, src); w3[u] = __shfl(tf.w_se, src);
    }
#pragma unroll
    for (int u = 0; u < PGT_UNROLL; ++u) {
      const int src = (it0 + u) * RPI + rsub;  // owning lane: (point src % P, view wave * VPW + src / P)
      const int sv = wave * VPW + src / P;
      if (lane_on && src < 64 && sv < V) {
        float* o = tile + ((src & (P - 1)) * V + sv) * C + 3 + c4 * 4;
        o[0] = fmaf(td[u].x, w3[u], fmaf(tc[u].x, w2[u], fmaf(tb[u].x, w1[u], ta[u].x * w0[u])));
        o[1] = fmaf(td[u].y, w3[u], fmaf(tc[u].y, w2[u], fmaf(tb[u].y, w1[u], ta[u].y * w0[u])));
        o[2] = fmaf(td[u].z, w3[u], fmaf(tc[u].z, w2[u], fmaf(tb[u].z, w1[u], ta[u].z * w0[u])));
        o[3] = fmaf(td[u].w, w3[u], fmaf(tc[u].w, w2[u], fmaf(tb[u].w, w1[u], ta[u].w * w0[u])));
      }
    }
  }
  __syncthreads();
  // ---- the tile leaves as one contiguous burst (p0 * V * C * 4 bytes: 16-byte aligned since P is a multiple of 4) ----
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long npt = !tile_live ? 0 : ((q.n_pts - p0 < P) ? (q.n_pts - p0) : P);
  {
    const int nflt = (int)npt * V * C;
    float* dst = rgb_feat + p0 * V * C;
    const float4* src4 = reinterpret_cast<const float4*>(tile);
    float4* dst4 = reinterpret_cast<float4*>(dst);
    for (int i = tid; i < (nflt >> 2); i += nthr) nt_store4<1>(dst4 + i, src4[i]);
    for (int i = (nflt & ~3) + tid; i < nflt; i += nthr) nt_store1<1>(dst + i, tile[i]);
  }
  // ---- ray_diff [P][V] float4 and mask [P][V]: their own LDS behind the tile (round 5: no second and third barrier, no LDS phase between the bursts) ----
  float4* rdt = reinterpret_cast<float4*>(tile + P * V * C);
  float* mkt = tile + P * V * C + P * V * 4;
  {
    const int nrow = (int)npt * V;
    float4* dst4 = ray_diff + p0 * V;
    if (ray_diff != nullptr)  // (the dynamic branch never reads it: DynibarDynamic takes no ray_diff -- 16 of its 160 output bytes per point-view stay home)
      for (int i = tid; i < nrow; i += nthr) nt_store4<1>(dst4 + i, rdt[i]);
    float* dm = mask + p0 * V;  // p0 * V * 4 bytes: 16-byte aligned
    const float4* m4 = reinterpret_cast<const float4*>(mkt);
    for (int i = tid; i < (nrow >> 2); i += nthr) nt_store4<1>(reinterpret_cast<float4*>(dm) + i, m4[i]);
    for (int i = (nrow & ~3) + tid; i < nrow; i += nthr) nt_store1<1>(dm + i, mkt[i]);
    if (pix_mask != nullptr && tid < npt) {
      float cnt = 0.f;
      for (int k = 0; k < V; ++k) cnt += mkt[tid * V + k];  // 0/1 addends: exact in any order
      pix_mask[p0 + tid] = cnt > q.mask_thresh ? 1.0f : 0.0f;
    }
  }
}

static PGTraj pg_traj_of(const DynProjectGatherParams* p) {
  PGTraj tj;
  tj.coeff = p->traj_coeff; tj.basis = p->traj_basis; tj.rows = p->traj_rows; tj.B = p->traj_B; tj.ref = p->traj_ref;
  return tj;
}
static int project_gather_rows(const DynProjectGatherParams* p, void* stream) {
  PGShape q;
  q.R = p->R; q.S = p->S; q.V = p->V; q.H = p->H; q.W = p->W; q.Hf = p->Hf; q.Wf = p->Wf; q.F = p->F;
  q.img_h = p->img_h; q.img_w = p->img_w;
  q.inv_wm1 = 1.0f / (p->img_w - 1.0f); q.inv_hm1 = 1.0f / (p->img_h - 1.0f);
  q.N = (long)p->R * p->S * p->V;
  q.mV = (unsigned)(((1ULL << 32) + p->V - 1) / p->V);
  q.mS = (unsigned)(((1ULL << 32) + p->S - 1) / p->S);
  q.ntask = (q.N + 63) / 64;
  const int wpb = PG_THREADS / 64;
  q.tasks_per_xcd = ((q.ntask + 7) / 8 + wpb - 1) / wpb * wpb;  // whole workgroups per XCD range
  const long nblocks = 8 * (q.tasks_per_xcd / wpb);
  DYN_LAUNCH(DYN_K_PROJECT_GATHER, "dyn_project_gather", k_project_gather, dim3((unsigned)nblocks), dim3(PG_THREADS),
             PG_STAGE ? (size_t)(PG_THREADS / 64) * 64 * (3 + p->F) * sizeof(float) : 0, (hipStream_t)stream, q,
             p->ray_o, p->ray_d, p->z_vals, p->pts_st, p->xyz, reinterpret_cast<const float4*>(p->proj), p->query_center, p->src_rgb,
             reinterpret_cast<const float4*>(p->feat_cl), p->rgb_feat, reinterpret_cast<float4*>(p->ray_diff), p->mask, pg_traj_of(p));
  if (p->pix_mask != nullptr) return dyn_sample_mask(p->mask, p->R * p->S, p->V, p->pix_mask_thresh, p->pix_mask, stream);
  return 0;
}

extern "C" int dyn_project_gather(const DynProjectGatherParams* p, void* stream) {
  DYN_REQUIRE(p, "dyn_project_gather: null params");
  DYN_REQUIRE(p->R > 0 && p->S > 0 && p->V > 0, "dyn_project_gather: empty problem");
  DYN_REQUIRE(p->F > 0 && (p->F % 4) == 0 && p->F <= 256, "dyn_project_gather: F must be a multiple of 4 (<=256)");
  DYN_REQUIRE(p->proj && p->query_center && p->src_rgb && p->feat_cl && p->rgb_feat && p->mask,
              "dyn_project_gather: null pointer");
  DYN_REQUIRE(p->pts_st != nullptr || (p->ray_o && p->ray_d && p->z_vals), "dyn_project_gather: need pts_st or (ray_o, ray_d, z_vals)");
  DYN_REQUIRE(p->traj_coeff == nullptr || (p->xyz == nullptr && p->pts_st != nullptr && p->traj_basis != nullptr && p->traj_rows != nullptr && p->traj_B > 0 && p->traj_ref >= 0),
              "dyn_project_gather: the fused trajectory form needs pts_st, traj_basis, traj_rows (device), traj_B > 0, traj_ref >= 0 and no xyz");
  DYN_REQUIRE(p->H > 1 && p->W > 1 && p->Hf > 1 && p->Wf > 1, "dyn_project_gather: maps must be at least 2x2");
  const long N = (long)p->R * p->S * p->V;
  DYN_REQUIRE(N < (1L << 31) && (long)p->V * p->H * p->W * 3 < (1L << 31) && (long)p->V * p->Hf * p->Wf * p->F < (1L << 31),
              "dyn_project_gather: R*S*V and the map sizes must stay below 2^31 elements (split the ray batch)");
  static const int legacy = getenv("DYN_PG_ROWS") != nullptr;  // developer A/B: the row-order kernel of round 1
  const int C = 3 + p->F;
  // points per tile: 64 (one view per wave) while the [P][V][C] tile leaves room for two workgroups per CU (V <= 8) -- and for 9 ... 11 views, where one
  // 9- to 11-wave workgroup per CU measured 3-5 % faster than three 6-wave workgroups with an idle view slot (tools/k1sweep.py, round 4: 11 views 172 vs
  // 177-181 us; at 15 views the two-views-per-wave form wins, 201 vs 221 us) --, else 32 (two views per wave)
  static const int force_p = getenv("DYN_PG_P") ? atoi(getenv("DYN_PG_P")) : 0;  // developer A/B
  // Points per tile.  Rounds 2-4 chose the tallest tile that left two workgroups per CU (P = 64, one view per wave; P = 32 beyond 8 views) from sweeps of the kernel
  // ALONE with the maps evicted between launches (tools/k1sweep.py).  Round 5 measured it INSIDE the pipeline (bench step, DYN_PG_P forced; gpurun_out/r5c9_k1_v11.txt,
  // r5c10_k1.txt): P = 16 -- four views per wave, 2-4 waves and 20-40 KiB per workgroup, five to eight independent workgroups per CU whose load and store phases
  // overlap -- is the fastest at every view count: 7 views 88.0 -> 85.1 us, 8 views 91.7 -> 89.4 (0.504 of 8 TB/s), 11 views 143.0 -> 124.3 (0.42 -> 0.50),
  // 15 views 165.1 -> 163.9; frame 24.1 -> 22.9 ms.  (The stand-alone sweep had P = 16 at +12 % for 8 views: there the cold maps dominate and taller tiles re-use taps.)
  DYN_REQUIRE(force_p == 0 || force_p == 8 || force_p == 16 || force_p == 32 || force_p == 64, "dyn_project_gather: DYN_PG_P must be 8, 16, 32 or 64 (got %d)", force_p);
  const int P = force_p ? force_p : PGT_DEFAULT_P;
  const int waves = (p->V * P + 63) / 64;
  const size_t lds = ((size_t)P * p->V * (C + 5) + (p->traj_coeff != nullptr ? (size_t)waves * P * ((3 * p->traj_B) | 1) : 0)) * sizeof(float);
  if (legacy || waves > 16 || lds > 160 * 1024 || (64 % (p->F / 4)) != 0) return project_gather_rows(p, stream);
  const PGTraj tj = pg_traj_of(p);
  PGTile q;
  q.R = p->R; q.S = p->S; q.V = p->V; q.H = p->H; q.W = p->W; q.Hf = p->Hf; q.Wf = p->Wf; q.F = p->F;
  q.img_h = p->img_h; q.img_w = p->img_w;
  q.inv_wm1 = 1.0f / (p->img_w - 1.0f); q.inv_hm1 = 1.0f / (p->img_h - 1.0f);
  q.mask_thresh = p->pix_mask_thresh;
  q.mS = (unsigned)(((1ULL << 32) + p->S - 1) / p->S);
  q.n_pts = (long)p->R * p->S;
  q.ntile = (q.n_pts + P - 1) / P;
  q.tiles_per_xcd = (q.ntile + 7) / 8;
  const long nblocks = 8 * q.tiles_per_xcd;
  static const int no_pref = getenv("DYN_PG_NOPREF") != nullptr;  // developer A/B
  // the first 1024 workgroups (four per CU) share the streaming of the maps (round 5, inside the pipeline, P = 16, us at 8 / 11 views: none 120.7 / 195.2,
  // 256 workgroups 101.7 / 132.9, 512: 87.8 / 124.8, 1024: 81.9 / 117.7, 1280-2048: 83.7-84.7 / 118.9-120.7, 4096: 86.5 / 126.9)
  static const int pref_env = getenv("DYN_PG_PREF") ? atoi(getenv("DYN_PG_PREF")) : 0;  // developer A/B
  const long pref_n = pref_env > 0 ? pref_env : 1024;
  q.pref_wgs = no_pref ? 0 : (int)(nblocks < pref_n ? nblocks : pref_n);
  if (P == 8)
    DYN_LAUNCH(DYN_K_PROJECT_GATHER, "dyn_project_gather", k_project_gather_tile<8>, dim3((unsigned)nblocks), dim3(waves * 64), lds, (hipStream_t)stream, q,
               p->ray_o, p->ray_d, p->z_vals, p->pts_st, p->xyz, reinterpret_cast<const float4*>(p->proj), p->query_center, p->src_rgb,
               reinterpret_cast<const float4*>(p->feat_cl), p->rgb_feat, reinterpret_cast<float4*>(p->ray_diff), p->mask, p->pix_mask, tj);
  else if (P == 16)
    DYN_LAUNCH(DYN_K_PROJECT_GATHER, "dyn_project_gather", k_project_gather_tile<16>, dim3((unsigned)nblocks), dim3(waves * 64), lds, (hipStream_t)stream, q,
               p->ray_o, p->ray_d, p->z_vals, p->pts_st, p->xyz, reinterpret_cast<const float4*>(p->proj), p->query_center, p->src_rgb,
               reinterpret_cast<const float4*>(p->feat_cl), p->rgb_feat, reinterpret_cast<float4*>(p->ray_diff), p->mask, p->pix_mask, tj);
  else if (P == 64)
    DYN_LAUNCH(DYN_K_PROJECT_GATHER, "dyn_project_gather", k_project_gather_tile<64>, dim3((unsigned)nblocks), dim3(waves * 64), lds, (hipStream_t)stream, q,
               p->ray_o, p->ray_d, p->z_vals, p->pts_st, p->xyz, reinterpret_cast<const float4*>(p->proj), p->query_center, p->src_rgb,
               reinterpret_cast<const float4*>(p->feat_cl), p->rgb_feat, reinterpret_cast<float4*>(p->ray_diff), p->mask, p->pix_mask, tj);
  else
    DYN_LAUNCH(DYN_K_PROJECT_GATHER, "dyn_project_gather", k_project_gather_tile<32>, dim3((unsigned)nblocks), dim3(waves * 64), lds, (hipStream_t)stream, q,
               p->ray_o, p->ray_d, p->z_vals, p->pts_st, p->xyz, reinterpret_cast<const float4*>(p->proj), p->query_center, p->src_rgb,
               reinterpret_cast<const float4*>(p->feat_cl), p->rgb_feat, reinterpret_cast<float4*>(p->ray_diff), p->mask, p->pix_mask, tj);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the feature gather (training; the autograd of F.grid_sample w.r.t. its input, projection.py:160-167):
// d featmaps[v, y, x, :] += w_tap * d rgb_feat[row, 3:3+F] at the four taps of every (point, view) row.  The pixel location is
// recomputed with the forward kernel's own arithmetic (same projection, same make_taps), so the taps and weights are the forward's.
// One thread per (row, group of four channels); fp32 hardware atomics into the channels-last gradient map (zeroed by the caller).
// xyz [V,R,S,3] (the dynamic branch's motion-displaced points) or NULL (static branch: every view sees pts_st).  The gradient w.r.t.
// the locations themselves (the motion path) is not part of this slice.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_bwd(PGShape q, const float* __restrict__ pts_st, const float* __restrict__ xyz, const float4* __restrict__ proj4,
                                                    const float* __restrict__ drgb_feat, long ld_d, int col0, float* __restrict__ dfeat) {
  const int G = q.F / 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx / G;
  const int cg = (int)(idx - row * G);
  if (row >= q.N) return;
  const long rs = row / q.V;
  const int v = (int)(row - rs * q.V);
  const float* pt = xyz != nullptr ? xyz + ((long)v * q.R * q.S + rs) * 3 : pts_st + rs * 3;  // per-view (motion displaced) point
  const float x = pt[0], y = pt[1], z3 = pt[2];
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2];
  const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
  const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
  const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
  const float zc = fmaxf(hz, 1e-8f);
  float px = hx / zc, py = hy / zc;  // IEEE division, like the reference's tensor division (projection.py:53-55)
  px = fminf(fmaxf(px, -1e6f), 1e6f);
  py = fminf(fmaxf(py, -1e6f), 1e6f);
  const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;  // normalize(): 2 * pixel / [w - 1, h - 1] - 1 (projection.py:22-30), divisions as written there
  const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
  const Taps t = make_taps(nx, ny, q.Wf, q.Hf);
  const float* d = drgb_feat + row * ld_d + col0 + cg * 4;
  const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
  float* base = dfeat + (long)v * q.Hf * q.Wf * q.F + cg * 4;
  const float wts[4] = {t.w_nw, t.w_ne, t.w_sw, t.w_se};
  const int offs[4] = {t.y0 * q.Wf + t.x0, t.y0 * q.Wf + t.x1, t.y1 * q.Wf + t.x0, t.y1 * q.Wf + t.x1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (wts[k] == 0.f) continue;
    float* o = base + (long)offs[k] * q.F;
    atomicAdd(o + 0, wts[k] * d0);
    atomicAdd(o + 1, wts[k] * d1);
    atomicAdd(o + 2, wts[k] * d2);
    atomicAdd(o + 3, wts[k] * d3);
  }
}

// F == 32 (the only width the encoder produces): lane = channel, so one atomic instruction adds two complete 128-byte pixel records
// (32 lanes each) instead of 4-byte pieces of eight records -- the L2 atomic units work per line.
__global__ void __launch_bounds__(256) k_gather_bwd32(PGShape q, const float* __restrict__ pts_st, const float* __restrict__ xyz,
                                                      const float4* __restrict__ proj4, const float* __restrict__ drgb_feat, long ld_d, int col0,
                                                      float* __restrict__ dfeat) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row = idx >> 5;
  const int c = (int)(idx & 31);
  if (row >= q.N) return;
  const long rs = row / q.V;
  const int v = (int)(row - rs * q.V);
  const float* pt = xyz != nullptr ? xyz + ((long)v * q.R * q.S + rs) * 3 : pts_st + rs * 3;
  const float x = pt[0], y = pt[1], z3 = pt[2];
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2];
  const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
  const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
  const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
  const float zc = fmaxf(hz, 1e-8f);
  float px = hx / zc, py = hy / zc;  // IEEE division, like the reference's tensor division (projection.py:53-55)
  px = fminf(fmaxf(px, -1e6f), 1e6f);
  py = fminf(fmaxf(py, -1e6f), 1e6f);
  const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;  // normalize(): 2 * pixel / [w - 1, h - 1] - 1 (projection.py:22-30), divisions as written there
  const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
  const Taps t = make_taps(nx, ny, q.Wf, q.Hf);
  const float d = drgb_feat[row * ld_d + col0 + c];
  float* base = dfeat + (long)v * q.Hf * q.Wf * 32 + c;
  const float wts[4] = {t.w_nw, t.w_ne, t.w_sw, t.w_se};
  const int offs[4] = {t.y0 * q.Wf + t.x0, t.y0 * q.Wf + t.x1, t.y1 * q.Wf + t.x0, t.y1 * q.Wf + t.x1};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (wts[k] != 0.f) atomicAdd(base + (long)offs[k] * 32, wts[k] * d);
}

// The same scatter walked along the ray: a thread owns (ray, view, channel) and visits the S samples in order.  Neighbouring samples of a
// ray land in the same bilinear cell of a source map again and again (an epipolar line of a 72 x 128 map is a few dozen cells long for 64
// samples), and every atomic on one 128-byte pixel record serialises in L2 behind the others on it -- so contributions to an unchanged cell
// are summed in registers and leave as ONE set of four atomics when the cell changes.  Next sample's point and gradient are requested before
// the current one is used.
__global__ void __launch_bounds__(256) k_gather_bwd32_ray(PGShape q, const float* __restrict__ pts_st, const float* __restrict__ xyz,
                                                          const float4* __restrict__ proj4, const float* __restrict__ drgb_feat, long ld_d, int col0,
                                                          float* __restrict__ dfeat) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long rv = idx >> 5;
  const int c = (int)(idx & 31);
  if (rv >= (long)q.R * q.V) return;
  const long r = rv / q.V;
  const int v = (int)(rv - r * q.V);
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2];
  float* base = dfeat + (long)v * q.Hf * q.Wf * 32 + c;
  const long rs0 = r * q.S;
  const float* pt0 = xyz != nullptr ? xyz + ((long)v * q.R * q.S + rs0) * 3 : pts_st + rs0 * 3;
  const float* d0 = drgb_feat + (rs0 * q.V + v) * ld_d + col0 + c;
  const long d_step = (long)q.V * ld_d;
  int cur[4] = {-1, -1, -1, -1};
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float x = pt0[0], y = pt0[1], z3 = pt0[2], d = d0[0];
  for (int s = 0; s < q.S; ++s) {
    const int sn = s + 1 < q.S ? s + 1 : s;
    const float xn = pt0[sn * 3], yn = pt0[sn * 3 + 1], zn = pt0[sn * 3 + 2], dn = d0[sn * d_step];
    const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
    const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
    const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
    const float zc = fmaxf(hz, 1e-8f);
    float px = hx / zc, py = hy / zc;  // the forward pass's arithmetic (k_gather_bwd32 above)
    px = fminf(fmaxf(px, -1e6f), 1e6f);
    py = fminf(fmaxf(py, -1e6f), 1e6f);
    const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;
    const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
    const Taps t = make_taps(nx, ny, q.Wf, q.Hf);
    const int offs[4] = {t.y0 * q.Wf + t.x0, t.y0 * q.Wf + t.x1, t.y1 * q.Wf + t.x0, t.y1 * q.Wf + t.x1};
    const float wts[4] = {t.w_nw, t.w_ne, t.w_sw, t.w_se};
    if (offs[0] != cur[0] || offs[1] != cur[1] || offs[2] != cur[2] || offs[3] != cur[3]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (acc[k] != 0.f) atomicAdd(base + (long)cur[k] * 32, acc[k]);
        cur[k] = offs[k];
        acc[k] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = fmaf(wts[k], d, acc[k]);
    x = xn; y = yn; z3 = zn; d = dn;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (acc[k] != 0.f) atomicAdd(base + (long)cur[k] * 32, acc[k]);
}

extern "C" int dyn_gather_bwd(const float* pts_st, const float* xyz, const float* proj, int R, int S, int V, int Hf, int Wf, int F, float img_h, float img_w,
                              const float* drgb_feat, long ld_d, int col0, float* dfeat_cl, void* stream) {
  DYN_REQUIRE((pts_st || xyz) && proj && drgb_feat && dfeat_cl, "dyn_gather_bwd: null pointer");
  DYN_REQUIRE(R > 0 && S > 0 && V > 0 && Hf > 1 && Wf > 1 && F > 0 && (F % 4) == 0, "dyn_gather_bwd: bad shape");
  PGShape q;
  q.R = R; q.S = S; q.V = V; q.H = 0; q.W = 0; q.Hf = Hf; q.Wf = Wf; q.F = F;
  q.img_h = img_h; q.img_w = img_w;
  q.inv_wm1 = 1.0f / (img_w - 1.0f); q.inv_hm1 = 1.0f / (img_h - 1.0f);
  q.N = (long)R * S * V;
  q.mV = q.mS = 0; q.ntask = 0; q.tasks_per_xcd = 0;
  if (F == 32) {
    if (S >= 8) {  // along the rays, contributions to an unchanged bilinear cell merged in registers
      DYN_LAUNCH(DYN_K_TRAIN_GATHER_BWD, "dyn_gather_bwd", k_gather_bwd32_ray, dim3((unsigned)(((long)R * V * 32 + 255) / 256)), dim3(256), 0,
                 (hipStream_t)stream, q, pts_st, xyz, reinterpret_cast<const float4*>(proj), drgb_feat, ld_d, col0, dfeat_cl);
      return 0;
    }
    DYN_LAUNCH(DYN_K_TRAIN_GATHER_BWD, "dyn_gather_bwd", k_gather_bwd32, dim3((unsigned)((q.N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q,
               pts_st, xyz, reinterpret_cast<const float4*>(proj), drgb_feat, ld_d, col0, dfeat_cl);
    return 0;
  }
  const long n = q.N * (F / 4);
  DYN_LAUNCH(DYN_K_TRAIN_GATHER_BWD, "dyn_gather_bwd", k_gather_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, pts_st, xyz,
             reinterpret_cast<const float4*>(proj), drgb_feat, ld_d, col0, dfeat_cl);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the gather w.r.t. the sample LOCATIONS (training, third slice: the motion path; the autograd of F.grid_sample w.r.t. its
// grid, of normalize() and of compute_projections, projection.py:32-59,:134-167): d rgb_feat[row, 0:3+F] -> d xyz[v, point, 0:3].
// For one map, val_c = (1-fy)[(1-fx) NW_c + fx NE_c] + fy [(1-fx) SW_c + fx SE_c] with zero-valued corners outside the map, so
//   d val_c / d ix = (1-fy)(NE_c - NW_c) + fy (SE_c - SW_c),   d val_c / d iy = (1-fx)(SW_c - NW_c) + fx (SE_c - NE_c);
// ix = (nx + 1)(W_m - 1)/2, nx = 2 px / (w - 1) - 1, px = clamp(hx / max(hz, 1e-8)), h = P [x y z 1]^T.  One thread per (point, view).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tap_grad(const float* __restrict__ map, int Wm, int Hm, int C, float nx, float ny, const float* __restrict__ d,
                                         float& gnx, float& gny) {
  const float ix = safe_floor_coord((nx + 1.0f) * ((float)(Wm - 1) / 2.0f), (float)Wm);
  const float iy = safe_floor_coord((ny + 1.0f) * ((float)(Hm - 1) / 2.0f), (float)Hm);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float fx = ix - fx0, fy = iy - fy0;
  const bool x0ok = (x0 >= 0) && (x0 < Wm), x1ok = (x0 + 1 >= 0) && (x0 + 1 < Wm);
  const bool y0ok = (y0 >= 0) && (y0 < Hm), y1ok = (y0 + 1 >= 0) && (y0 + 1 < Hm);
  const float* nw = map + ((long)iclamp(y0, Hm - 1) * Wm + iclamp(x0, Wm - 1)) * C;
  const float* ne = map + ((long)iclamp(y0, Hm - 1) * Wm + iclamp(x0 + 1, Wm - 1)) * C;
  const float* sw = map + ((long)iclamp(y0 + 1, Hm - 1) * Wm + iclamp(x0, Wm - 1)) * C;
  const float* se = map + ((long)iclamp(y0 + 1, Hm - 1) * Wm + iclamp(x0 + 1, Wm - 1)) * C;
  const float m_nw = (x0ok && y0ok) ? 1.f : 0.f, m_ne = (x1ok && y0ok) ? 1.f : 0.f, m_sw = (x0ok && y1ok) ? 1.f : 0.f, m_se = (x1ok && y1ok) ? 1.f : 0.f;
  float gix = 0.f, giy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float a = nw[c] * m_nw, b = ne[c] * m_ne, e = sw[c] * m_sw, f = se[c] * m_se;
    gix += d[c] * ((1.0f - fy) * (b - a) + fy * (f - e));
    giy += d[c] * ((1.0f - fx) * (e - a) + fx * (f - b));
  }
  gnx += gix * ((float)(Wm - 1) / 2.0f);
  gny += giy * ((float)(Hm - 1) / 2.0f);
}

__global__ void __launch_bounds__(256) k_gather_bwd_pts(PGShape q, const float* __restrict__ pts_st, const float* __restrict__ xyz,
                                                        const float4* __restrict__ proj4, const float* __restrict__ src_rgb,
                                                        const float* __restrict__ feat_cl, const float* __restrict__ drgb_feat, long ld_d,
                                                        float* __restrict__ dxyz) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= q.N) return;
  const long rs = row / q.V;
  const int v = (int)(row - rs * q.V);
  const float* pt = xyz != nullptr ? xyz + ((long)v * q.R * q.S + rs) * 3 : pts_st + rs * 3;
  const float x = pt[0], y = pt[1], z3 = pt[2];
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2];
  const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
  const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
  const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
  const float zc = fmaxf(hz, 1e-8f);
  const float izc = 1.0f / zc;
  const float pxu = hx / zc, pyu = hy / zc;  // the forward's pixel (same taps); izc only scales the gradient
  const float px = fminf(fmaxf(pxu, -1e6f), 1e6f), py = fminf(fmaxf(pyu, -1e6f), 1e6f);
  const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;  // normalize(): 2 * pixel / [w - 1, h - 1] - 1 (projection.py:22-30), divisions as written there
  const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
  const float* d = drgb_feat + row * ld_d;
  float gnx = 0.f, gny = 0.f;
  tap_grad(src_rgb + (long)v * q.H * q.W * 3, q.W, q.H, 3, nx, ny, d, gnx, gny);
  tap_grad(feat_cl + (long)v * q.Hf * q.Wf * q.F, q.Wf, q.Hf, q.F, nx, ny, d + 3, gnx, gny);
  const float gpx = (pxu >= -1e6f && pxu <= 1e6f) ? gnx * 2.0f * q.inv_wm1 : 0.f;
  const float gpy = (pyu >= -1e6f && pyu <= 1e6f) ? gny * 2.0f * q.inv_hm1 : 0.f;
  const float ghx = gpx * izc, ghy = gpy * izc;
  const float ghz = hz > 1e-8f ? -(gpx * hx + gpy * hy) * izc * izc : 0.f;
  float* o = dxyz + ((long)v * q.R * q.S + rs) * 3;
  o[0] = P0.x * ghx + P1.x * ghy + P2.x * ghz;
  o[1] = P0.y * ghx + P1.y * ghy + P2.y * ghz;
  o[2] = P0.z * ghx + P1.z * ghy + P2.z * ghz;
}

// The same for 32-channel maps with one lane per channel (32 lanes per (point, view) row): the four pixel records of a tap and the row's
// gradient are each ONE coalesced 128-byte read instead of 32 four-byte reads per thread (lanes 0..2 also take the colour taps), and the
// per-channel products meet by a 32-lane butterfly; lane 0 applies the chain rule.
__device__ __forceinline__ void tap_grad_lane(const float* __restrict__ map, int Wm, int Hm, int C, float nx, float ny, float dc, int c, float& gnx,
                                              float& gny) {
  const float ix = safe_floor_coord((nx + 1.0f) * ((float)(Wm - 1) / 2.0f), (float)Wm);
  const float iy = safe_floor_coord((ny + 1.0f) * ((float)(Hm - 1) / 2.0f), (float)Hm);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float fx = ix - fx0, fy = iy - fy0;
  const bool x0ok = (x0 >= 0) && (x0 < Wm), x1ok = (x0 + 1 >= 0) && (x0 + 1 < Wm);
  const bool y0ok = (y0 >= 0) && (y0 < Hm), y1ok = (y0 + 1 >= 0) && (y0 + 1 < Hm);
  const float a = (x0ok && y0ok) ? map[((long)iclamp(y0, Hm - 1) * Wm + iclamp(x0, Wm - 1)) * C + c] : 0.f;
  const float b = (x1ok && y0ok) ? map[((long)iclamp(y0, Hm - 1) * Wm + iclamp(x0 + 1, Wm - 1)) * C + c] : 0.f;
  const float e = (x0ok && y1ok) ? map[((long)iclamp(y0 + 1, Hm - 1) * Wm + iclamp(x0, Wm - 1)) * C + c] : 0.f;
  const float f = (x1ok && y1ok) ? map[((long)iclamp(y0 + 1, Hm - 1) * Wm + iclamp(x0 + 1, Wm - 1)) * C + c] : 0.f;
  gnx += dc * ((1.0f - fy) * (b - a) + fy * (f - e)) * ((float)(Wm - 1) / 2.0f);
  gny += dc * ((1.0f - fx) * (e - a) + fx * (f - b)) * ((float)(Hm - 1) / 2.0f);
}
__global__ void __launch_bounds__(256) k_gather_bwd_pts32(PGShape q, const float* __restrict__ pts_st, const float* __restrict__ xyz,
                                                          const float4* __restrict__ proj4, const float* __restrict__ src_rgb,
                                                          const float* __restrict__ feat_cl, const float* __restrict__ drgb_feat, long ld_d,
                                                          float* __restrict__ dxyz) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long row_raw = idx >> 5;
  const int c = (int)(idx & 31);
  const bool live = row_raw < q.N;
  const long row = live ? row_raw : q.N - 1;  // (every lane stays for the butterfly)
  const long rs = row / q.V;
  const int v = (int)(row - rs * q.V);
  const float* pt = xyz != nullptr ? xyz + ((long)v * q.R * q.S + rs) * 3 : pts_st + rs * 3;
  const float x = pt[0], y = pt[1], z3 = pt[2];
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2];
  const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
  const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
  const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
  const float zc = fmaxf(hz, 1e-8f);
  const float izc = 1.0f / zc;
  const float pxu = hx / zc, pyu = hy / zc;  // the forward's pixel (same taps); izc only scales the gradient
  const float px = fminf(fmaxf(pxu, -1e6f), 1e6f), py = fminf(fmaxf(pyu, -1e6f), 1e6f);
  const float nx = 2.0f * px / (q.img_w - 1.0f) - 1.0f;
  const float ny = 2.0f * py / (q.img_h - 1.0f) - 1.0f;
  const float* d = drgb_feat + row * ld_d;
  float gnx = 0.f, gny = 0.f;
  if (c < 3) tap_grad_lane(src_rgb + (long)v * q.H * q.W * 3, q.W, q.H, 3, nx, ny, d[c], c, gnx, gny);
  tap_grad_lane(feat_cl + (long)v * q.Hf * q.Wf * 32, q.Wf, q.Hf, 32, nx, ny, d[3 + c], c, gnx, gny);
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) {
    gnx += __shfl_xor(gnx, o);
    gny += __shfl_xor(gny, o);
  }
  if (c != 0 || !live) return;
  const float gpx = (pxu >= -1e6f && pxu <= 1e6f) ? gnx * 2.0f * q.inv_wm1 : 0.f;
  const float gpy = (pyu >= -1e6f && pyu <= 1e6f) ? gny * 2.0f * q.inv_hm1 : 0.f;
  const float ghx = gpx * izc, ghy = gpy * izc;
  const float ghz = hz > 1e-8f ? -(gpx * hx + gpy * hy) * izc * izc : 0.f;
  float* o = dxyz + ((long)v * q.R * q.S + rs) * 3;
  o[0] = P0.x * ghx + P1.x * ghy + P2.x * ghz;
  o[1] = P0.y * ghx + P1.y * ghy + P2.y * ghz;
  o[2] = P0.z * ghx + P1.z * ghy + P2.z * ghz;
}

extern "C" int dyn_gather_bwd_pts(const float* pts_st, const float* xyz, const float* proj, const float* src_rgb, const float* feat_cl, int R, int S,
                                  int V, int H, int W, int Hf, int Wf, int F, float img_h, float img_w, const float* drgb_feat, long ld_d,
                                  float* dxyz, void* stream) {
  DYN_REQUIRE((pts_st || xyz) && proj && src_rgb && feat_cl && drgb_feat && dxyz, "dyn_gather_bwd_pts: null pointer");
  DYN_REQUIRE(R > 0 && S > 0 && V > 0 && H > 1 && W > 1 && Hf > 1 && Wf > 1 && F > 0, "dyn_gather_bwd_pts: bad shape");
  PGShape q;
  q.R = R; q.S = S; q.V = V; q.H = H; q.W = W; q.Hf = Hf; q.Wf = Wf; q.F = F;
  q.img_h = img_h; q.img_w = img_w;
  q.inv_wm1 = 1.0f / (img_w - 1.0f); q.inv_hm1 = 1.0f / (img_h - 1.0f);
  q.N = (long)R * S * V;
  q.mV = q.mS = 0; q.ntask = 0; q.tasks_per_xcd = 0;
  if (F == 32) {
    DYN_LAUNCH(DYN_K_TRAIN_GATHER_BWD, "dyn_gather_bwd_pts", k_gather_bwd_pts32, dim3((unsigned)((q.N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
               q, pts_st, xyz, reinterpret_cast<const float4*>(proj), src_rgb, feat_cl, drgb_feat, ld_d, dxyz);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAIN_GATHER_BWD, "dyn_gather_bwd_pts", k_gather_bwd_pts, dim3((unsigned)((q.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q,
             pts_st, xyz, reinterpret_cast<const float4*>(proj), src_rgb, feat_cl, drgb_feat, ld_d, dxyz);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// K5: DCT-basis trajectory points  (render_ray.py:361-369, :686-709)
// ---------------------------------------------------------------------------------------------------------------
struct TrajRows {
  int n, ref;
  int rows[32];
};
// One workgroup = 256 consecutive points.  A point's 3 B coefficients (72 bytes at B = 6) and its n x 3 outputs are both short records at
// odd strides, so both sides go through LDS: the coefficient block of the workgroup is one contiguous run (coalesced in, odd row stride
// in LDS), the outputs of up to 8 trajectory rows are assembled as [row][256 x 3] and leave as contiguous runs.  (Lane-strided record
// accesses ran this kernel at a seventh of the HBM rate: 335 us for 1 M points.)
#define TRAJ_GROUP 8
__global__ void __launch_bounds__(256) k_trajectory_points(const float* __restrict__ coeff, const float* __restrict__ basis, const float* __restrict__ pts,
                                                           long n_pts, int B, TrajRows tr, float* __restrict__ pts_seq) {
  float* traj_lds = reinterpret_cast<float*>(dyn_smem);
  const int tid = threadIdx.x, C = 3 * B, CS = C | 1;  // odd row stride
  float* cs = traj_lds;             // [256][CS]
  float* ps = cs + 256 * CS;        // [256 x 3] the undisplaced points
  float* os = ps + 768;             // [TRAJ_GROUP][256 x 3]
  const long i0 = (long)blockIdx.x * 256;
  const int np = (int)((n_pts - i0 < 256) ? n_pts - i0 : 256);
  for (int e = tid; e < np * C; e += 256) cs[(e / C) * CS + (e % C)] = coeff[i0 * C + e];
  for (int e = tid; e < np * 3; e += 256) ps[e] = pts[i0 * 3 + e];
  __syncthreads();
  const float* c = cs + tid * CS;
  for (int v0 = 0; v0 < tr.n; v0 += TRAJ_GROUP) {
    const int nv = tr.n - v0 < TRAJ_GROUP ? tr.n - v0 : TRAJ_GROUP;
    for (int vv = 0; vv < nv; ++vv) {
      float qx = ps[tid * 3], qy = ps[tid * 3 + 1], qz = ps[tid * 3 + 2];
      traj_displace(c, basis, B, tr.rows[v0 + vv], tr.ref, qx, qy, qz);  // (the sums of the fused consumers: traj_displace)
      os[vv * 768 + tid * 3] = qx; os[vv * 768 + tid * 3 + 1] = qy; os[vv * 768 + tid * 3 + 2] = qz;
    }
    __syncthreads();
    for (int vv = 0; vv < nv; ++vv) {
      float* dst = pts_seq + ((long)(v0 + vv) * n_pts + i0) * 3;
      for (int e = tid; e < np * 3; e += 256) __builtin_nontemporal_store(os[vv * 768 + e], dst + e);
    }
    __syncthreads();
  }
}
extern "C" int dyn_trajectory_points(const float* coeff, const float* basis, const float* pts, long n_pts, int B, const int* rows, int n_rows,
                                     int row_ref, float* pts_seq, void* stream) {
  DYN_REQUIRE(coeff && basis && pts && rows && pts_seq, "dyn_trajectory_points: null pointer");
  DYN_REQUIRE(n_pts > 0 && B > 0 && n_rows > 0 && n_rows <= 32 && row_ref >= 0, "dyn_trajectory_points: bad argument");
  TrajRows tr;
  tr.n = n_rows; tr.ref = row_ref;
  for (int i = 0; i < 32; ++i) tr.rows[i] = i < n_rows ? rows[i] : -1;
  const size_t lds = (size_t)(256 * ((3 * B) | 1) + 768 + TRAJ_GROUP * 768) * sizeof(float);
  DYN_REQUIRE(lds <= 64 * 1024, "dyn_trajectory_points: too many basis functions");
  DYN_LAUNCH(DYN_K_TRAJECTORY, "dyn_trajectory_points", k_trajectory_points, dim3(dyn_cdiv(n_pts, 256)), dim3(256), lds, (hipStream_t)stream, coeff,
             basis, pts, n_pts, B, tr, pts_seq);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// a22 expected optical flow (render_ray.py:333-358) and expected scene flow (:584-595, :1086-1096): one wavefront per ray
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_render_flows(const float* __restrict__ weights, const float* __restrict__ pts_seq, const float* __restrict__ proj,
                                                      const float* __restrict__ uv, int R, int S, int V, float* __restrict__ flows) {
  const int lane = dyn_lane();
  const long w = (long)blockIdx.x * 4 + dyn_wave();  // (view, ray)
  if (w >= (long)V * R) return;
  const int v = (int)(w / R), r = (int)(w % R);
  float ex = 0.f, ey = 0.f, ez = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float wt = weights[(long)r * S + s];
    const float* q = pts_seq + (((long)v * R + r) * S + s) * 3;
    ex += wt * q[0]; ey += wt * q[1]; ez += wt * q[2];
  }
  ex = wave_sum(ex); ey = wave_sum(ey); ez = wave_sum(ez);
  if (lane == 0) {
    const float* P = proj + v * 16;  // rows of K . inv(c2w)
    const float hx = P[0] * ex + P[1] * ey + P[2] * ez + P[3];
    const float hy = P[4] * ex + P[5] * ey + P[6] * ez + P[7];
    const float hz = P[8] * ex + P[9] * ey + P[10] * ez + P[11];
    flows[((long)v * R + r) * 2 + 0] = hx / hz - uv[r * 2 + 0];
    flows[((long)v * R + r) * 2 + 1] = hy / hz - uv[r * 2 + 1];
  }
}
// the same from the motion coefficients (the fused form of the eval path: pts_seq does not exist).  The expected point is linear in the coefficients:
//   sum_s w_s (p_s + traj_row(c_s) - traj_ref(c_s)) = sum_s w_s p_s + sum_b (sum_s w_s c_s[a, b]) (basis[row, b] - basis[ref, b]),
// so ONE pass over a ray's samples (its S x 3 B coefficients are one contiguous run: coalesced into LDS, odd sample stride) serves all V views: 3 + 3 B weighted
// sums per ray, then V x 3 B products.  (Recomputing every displaced point per view cost 12 ms per frame against 0.5 ms for the materialised form; the
// reordering moves the flows by 1e-7 relative -- they are compared with the reference's at 2e-4 px.)  One wavefront per ray.
#define FLOW_MAX_B 16
__global__ void __launch_bounds__(256) k_render_flows_traj(const float* __restrict__ weights, const float* __restrict__ pts, PGTraj tj, const float* __restrict__ proj,
                                                           const float* __restrict__ uv, int R, int S, int V, float* __restrict__ flows) {
  const int lane = dyn_lane(), wave = dyn_wave();
  const int C3 = 3 * tj.B, CS = C3 | 1;
  float* cw = reinterpret_cast<float*>(dyn_smem) + (size_t)wave * (64 * CS + 3 * FLOW_MAX_B + 4);  // [64 samples][CS] staging, then the ray's 3 B + 3 sums
  float* sums = cw + 64 * CS;
  const int r = blockIdx.x * 4 + wave;
  const bool live = r < R;
  float acc[3 * FLOW_MAX_B + 3];
#pragma unroll
  for (int k = 0; k < 3 * FLOW_MAX_B + 3; ++k) acc[k] = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {  // 64 samples at a time through the wave's LDS block (its own: a wave runs in lock step, no workgroup barrier)
    const int ns = S - s0 < 64 ? S - s0 : 64;
    if (live) {
      const float* src = tj.coeff + ((long)r * S + s0) * C3;
      for (int e = lane; e < ns * C3; e += 64) cw[(e / C3) * CS + (e % C3)] = src[e];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (live && lane < ns) {
      const long i = (long)r * S + s0 + lane;
      const float wt = weights[i];
      acc[0] += wt * pts[i * 3]; acc[1] += wt * pts[i * 3 + 1]; acc[2] += wt * pts[i * 3 + 2];
      const float* c = cw + lane * CS;
#pragma unroll
      for (int k = 0; k < 3 * FLOW_MAX_B; ++k)
        if (k < C3) acc[3 + k] += wt * c[k];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int k = 0; k < 3 * FLOW_MAX_B + 3; ++k)
    if (k < C3 + 3) {
      const float t = wave_sum(acc[k]);
      if (lane == 0) sums[k] = t;
    }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  for (int v = lane; live && v < V; v += 64) {
    const int row = tj.rows[v];
    float e3[3] = {sums[0], sums[1], sums[2]};
    if (row >= 0) {
      for (int a = 0; a < 3; ++a) {
        float d = 0.f;
        for (int b = 0; b < tj.B; ++b) d += sums[3 + a * tj.B + b] * (tj.basis[(long)row * tj.B + b] - tj.basis[(long)tj.ref * tj.B + b]);
        e3[a] += d;
      }
    }
    const float* P = proj + v * 16;  // rows of K . inv(c2w)
    const float hx = P[0] * e3[0] + P[1] * e3[1] + P[2] * e3[2] + P[3];
    const float hy = P[4] * e3[0] + P[5] * e3[1] + P[6] * e3[2] + P[7];
    const float hz = P[8] * e3[0] + P[9] * e3[1] + P[10] * e3[2] + P[11];
    flows[((long)v * R + r) * 2 + 0] = hx / hz - uv[r * 2 + 0];
    flows[((long)v * R + r) * 2 + 1] = hy / hz - uv[r * 2 + 1];
  }
}
extern "C" int dyn_render_flows_traj(const float* weights, const float* pts, const float* coeff, const float* basis, int B, const int* rows_dev, int row_ref,
                                     const float* proj, const float* uv, int R, int S, int V, float* flows, void* stream) {
  DYN_REQUIRE(weights && pts && coeff && basis && rows_dev && proj && uv && flows && R > 0 && S > 0 && V > 0 && B > 0 && B <= FLOW_MAX_B && row_ref >= 0,
              "dyn_render_flows_traj: bad argument (1 <= B <= %d)", FLOW_MAX_B);
  PGTraj tj;
  tj.coeff = coeff; tj.basis = basis; tj.rows = rows_dev; tj.B = B; tj.ref = row_ref;
  const size_t lds = (size_t)4 * (64 * ((3 * B) | 1) + 3 * FLOW_MAX_B + 4) * sizeof(float);
  DYN_LAUNCH(DYN_K_RENDER_FLOWS, "dyn_render_flows_traj", k_render_flows_traj, dim3(dyn_cdiv(R, 4)), dim3(256), lds, (hipStream_t)stream, weights, pts, tj,
             proj, uv, R, S, V, flows);
  return 0;
}
extern "C" int dyn_render_flows(const float* weights, const float* pts_seq, const float* proj, const float* uv, int R, int S, int V, float* flows,
                                void* stream) {
  DYN_REQUIRE(weights && pts_seq && proj && uv && flows && R > 0 && S > 0 && V > 0, "dyn_render_flows: bad argument");
  DYN_LAUNCH(DYN_K_RENDER_FLOWS, "dyn_render_flows", k_render_flows, dim3(dyn_cdiv((long)V * R, 4)), dim3(256), 0, (hipStream_t)stream, weights,
             pts_seq, proj, uv, R, S, V, flows);
  return 0;
}

// Projector.compute_projections / compute_angle (projection.py:32-101) as stand-alone exports: the render path fuses them into the gather
// kernel; these serve callers of the reference's helper methods.  One thread per (view, point).
__global__ void __launch_bounds__(256) k_project_points(const float* __restrict__ xyz, const float* __restrict__ xyz_st, const float* __restrict__ proj,
                                                        const float* __restrict__ query_center, int V, long n_pts, float* __restrict__ pix,
                                                        float* __restrict__ in_front, float* __restrict__ ray_diff) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)V * n_pts) return;
  const int v = (int)(i / n_pts);
  const float x = xyz[i * 3], y = xyz[i * 3 + 1], z3 = xyz[i * 3 + 2];
  const float4* proj4 = reinterpret_cast<const float4*>(proj);
  const float4 P0 = proj4[v * 4], P1 = proj4[v * 4 + 1], P2 = proj4[v * 4 + 2], P3 = proj4[v * 4 + 3];
  if (pix != nullptr) {
    const float hx = fmaf(P0.w, 1.0f, fmaf(P0.z, z3, fmaf(P0.y, y, P0.x * x)));
    const float hy = fmaf(P1.w, 1.0f, fmaf(P1.z, z3, fmaf(P1.y, y, P1.x * x)));
    const float hz = fmaf(P2.w, 1.0f, fmaf(P2.z, z3, fmaf(P2.y, y, P2.x * x)));
    const float zc = fmaxf(hz, 1e-8f);
    pix[i * 2] = fminf(fmaxf(hx / zc, -1e6f), 1e6f);
    pix[i * 2 + 1] = fminf(fmaxf(hy / zc, -1e6f), 1e6f);
    in_front[i] = hz > 0.f ? 1.0f : 0.0f;
  }
  if (ray_diff != nullptr) {
    const float sx = xyz_st[i * 3], sy = xyz_st[i * 3 + 1], sz = xyz_st[i * 3 + 2];
    float ax, ay, az, bx, by, bz, dx, dy, dz;
    normalize3(query_center[0] - sx, query_center[1] - sy, query_center[2] - sz, ax, ay, az);
    normalize3(P3.x - x, P3.y - y, P3.z - z3, bx, by, bz);
    normalize3(ax - bx, ay - by, az - bz, dx, dy, dz);
    reinterpret_cast<float4*>(ray_diff)[i] = make_float4(dx, dy, dz, ax * bx + ay * by + az * bz);
  }
}
extern "C" int dyn_project_points(const float* xyz, const float* xyz_st, const float* proj, const float* query_center, int V, long n_pts, float* pix,
                                  float* in_front, float* ray_diff, void* stream) {
  DYN_REQUIRE(xyz && proj && V > 0 && n_pts > 0, "dyn_project_points: bad argument");
  DYN_REQUIRE((pix == nullptr) == (in_front == nullptr), "dyn_project_points: pix and in_front go together");
  DYN_REQUIRE(ray_diff == nullptr || (xyz_st != nullptr && query_center != nullptr), "dyn_project_points: ray_diff needs xyz_st and query_center");
  DYN_LAUNCH(DYN_K_RENDER_FLOWS, "dyn_project_points", k_project_points, dim3(dyn_cdiv((long)V * n_pts, 256)), dim3(256), 0, (hipStream_t)stream, xyz, xyz_st,
             proj, query_center, V, n_pts, pix, in_front, ray_diff);
  return 0;
}

// Backward of the trajectory points (training, third slice): pts_seq[v, p, a] = pts[p, a] + sum_b coeff[p, a B + b] (basis[row_v, b] -
// basis[ref, b]) for row_v >= 0, = pts[p, a] otherwise.  d coeff and d pts are per point; d basis is reduced per workgroup in LDS and added
// to the global table (zeroed by the caller) with atomics.
__global__ void __launch_bounds__(256) k_trajectory_bwd(const float* __restrict__ dseq, const float* __restrict__ coeff, const float* __restrict__ basis,
                                                        long n_pts, int B, TrajRows tr, float* __restrict__ dcoeff, float* __restrict__ dbasis,
                                                        float* __restrict__ dpts) {
  float* acc = reinterpret_cast<float*>(dyn_smem);  // [(n + 1)][B]: per trajectory row, last = the reference row
  const int tid = threadIdx.x;
  for (int e = tid; e < (tr.n + 1) * B; e += 256) acc[e] = 0.f;
  __syncthreads();
  const long p = (long)blockIdx.x * 256 + tid;
  if (p < n_pts) {
    float dp[3] = {0.f, 0.f, 0.f};
    for (int e = 0; e < 3 * B; ++e) dcoeff[p * 3 * B + e] = 0.f;
    for (int v = 0; v < tr.n; ++v) {
      const float* d = dseq + ((long)v * n_pts + p) * 3;
      const int row = tr.rows[v];
      for (int a = 0; a < 3; ++a) {
        dp[a] += d[a];
        if (row >= 0) {
          for (int b = 0; b < B; ++b) {
            dcoeff[p * 3 * B + a * B + b] += d[a] * (basis[(long)row * B + b] - basis[(long)tr.ref * B + b]);
            const float t = d[a] * coeff[p * 3 * B + a * B + b];
            atomicAdd(acc + v * B + b, t);
            atomicAdd(acc + tr.n * B + b, -t);
          }
        }
      }
    }
    if (dpts != nullptr)
      for (int a = 0; a < 3; ++a) dpts[p * 3 + a] = dp[a];
  }
  __syncthreads();
  for (int e = tid; e < (tr.n + 1) * B; e += 256) {
    const int v = e / B, b = e % B;
    const int row = v < tr.n ? tr.rows[v] : tr.ref;
    if (row >= 0 && acc[e] != 0.f) atomicAdd(dbasis + (long)row * B + b, acc[e]);
  }
}
// B <= 8 basis functions (the reference ships 6): a point's coefficients and their gradients stay in registers (the generic kernel above
// read-modify-writes 3 B floats of global memory per trajectory row with an 18-float stride between lanes), and the basis gradient is summed
// over the wave before it touches the workgroup's LDS accumulators (one atomic per wave instead of 64 on the same address).
__global__ void __launch_bounds__(256) k_trajectory_bwd8(const float* __restrict__ dseq, const float* __restrict__ coeff, const float* __restrict__ basis,
                                                         long n_pts, int B, TrajRows tr, float* __restrict__ dcoeff, float* __restrict__ dbasis,
                                                         float* __restrict__ dpts) {
  float* acc = reinterpret_cast<float*>(dyn_smem);  // [(n + 1)][B]: per trajectory row, last = the reference row
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < (tr.n + 1) * B; e += 256) acc[e] = 0.f;
  __syncthreads();
  const long p = (long)blockIdx.x * 256 + tid;
  const bool live = p < n_pts;
  const long pc = live ? p : n_pts - 1;
  float co[3][8], dc[3][8], dp[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      co[a][b] = b < B ? coeff[pc * 3 * B + a * B + b] : 0.f;
      dc[a][b] = 0.f;
    }
  for (int v = 0; v < tr.n; ++v) {
    const float* d = dseq + ((long)v * n_pts + pc) * 3;
    const float d0 = live ? d[0] : 0.f, d1 = live ? d[1] : 0.f, d2 = live ? d[2] : 0.f;
    dp[0] += d0; dp[1] += d1; dp[2] += d2;
    const int row = tr.rows[v];
    if (row >= 0) {  // uniform
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b < B) {
          const float db = basis[(long)row * B + b] - basis[(long)tr.ref * B + b];
          dc[0][b] += d0 * db; dc[1][b] += d1 * db; dc[2][b] += d2 * db;
          const float t = wave_sum(d0 * co[0][b] + d1 * co[1][b] + d2 * co[2][b]);
          if (lane == 0) {
            atomicAdd(acc + v * B + b, t);
            atomicAdd(acc + tr.n * B + b, -t);
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b < B) dcoeff[p * 3 * B + a * B + b] = dc[a][b];
    if (dpts != nullptr)
      for (int a = 0; a < 3; ++a) dpts[p * 3 + a] = dp[a];
  }
  __syncthreads();
  for (int e = tid; e < (tr.n + 1) * B; e += 256) {
    const int v = e / B, b = e % B;
    const int row = v < tr.n ? tr.rows[v] : tr.ref;
    if (row >= 0 && acc[e] != 0.f) atomicAdd(dbasis + (long)row * B + b, acc[e]);
  }
}
extern "C" int dyn_trajectory_bwd(const float* dseq, const float* coeff, const float* basis, long n_pts, int B, const int* rows, int n_rows,
                                  int row_ref, float* dcoeff, float* dbasis, float* dpts, void* stream) {
  DYN_REQUIRE(dseq && coeff && basis && rows && dcoeff && dbasis, "dyn_trajectory_bwd: null pointer");
  DYN_REQUIRE(n_pts > 0 && B > 0 && n_rows > 0 && n_rows <= 32 && row_ref >= 0, "dyn_trajectory_bwd: bad argument");
  TrajRows tr;
  tr.n = n_rows; tr.ref = row_ref;
  for (int i = 0; i < 32; ++i) tr.rows[i] = i < n_rows ? rows[i] : -1;
  if (B <= 8) {
    DYN_LAUNCH(DYN_K_TRAJECTORY, "dyn_trajectory_bwd", k_trajectory_bwd8, dim3(dyn_cdiv(n_pts, 256)), dim3(256),
               (size_t)(n_rows + 1) * B * sizeof(float), (hipStream_t)stream, dseq, coeff, basis, n_pts, B, tr, dcoeff, dbasis, dpts);
    return 0;
  }
  DYN_LAUNCH(DYN_K_TRAJECTORY, "dyn_trajectory_bwd", k_trajectory_bwd, dim3(dyn_cdiv(n_pts, 256)), dim3(256), (size_t)(n_rows + 1) * B * sizeof(float),
             (hipStream_t)stream, dseq, coeff, basis, n_pts, B, tr, dcoeff, dbasis, dpts);
  return 0;
}

// Backward of the expected optical flow (render_ray.py:333-358): flows[v, r] = pi(P_v E) - uv, E = sum_s w[r, s] q[v, r, s].
// dE = J^T dflow; dw[r, s] = sum_v dE_v . q[v, r, s]; dq[v, r, s] = w[r, s] dE_v.  One wavefront per ray, lanes over samples.
__global__ void __launch_bounds__(256) k_render_flows_bwd(const float* __restrict__ dflows, const float* __restrict__ weights,
                                                          const float* __restrict__ pts_seq, const float* __restrict__ proj, int R, int S, int V,
                                                          float* __restrict__ dweights, float* __restrict__ dseq) {
  const int lane = dyn_lane();
  const int r = blockIdx.x * 4 + dyn_wave();
  if (r >= R) return;
  for (int s = lane; s < S; s += 64) dweights[(long)r * S + s] = 0.f;
  for (int v = 0; v < V; ++v) {
    float ex = 0.f, ey = 0.f, ez = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float wt = weights[(long)r * S + s];
      const float* q = pts_seq + (((long)v * R + r) * S + s) * 3;
      ex += wt * q[0]; ey += wt * q[1]; ez += wt * q[2];
    }
    ex = wave_sum(ex); ey = wave_sum(ey); ez = wave_sum(ez);
    const float* P = proj + v * 16;
    const float hx = P[0] * ex + P[1] * ey + P[2] * ez + P[3];
    const float hy = P[4] * ex + P[5] * ey + P[6] * ez + P[7];
    const float hz = P[8] * ex + P[9] * ey + P[10] * ez + P[11];
    const float gx = dflows[((long)v * R + r) * 2], gy = dflows[((long)v * R + r) * 2 + 1];
    const float ghx = gx / hz, ghy = gy / hz, ghz = -(gx * hx + gy * hy) / (hz * hz);
    const float dEx = P[0] * ghx + P[4] * ghy + P[8] * ghz;
    const float dEy = P[1] * ghx + P[5] * ghy + P[9] * ghz;
    const float dEz = P[2] * ghx + P[6] * ghy + P[10] * ghz;
    for (int s = lane; s < S; s += 64) {
      const long o = (((long)v * R + r) * S + s) * 3;
      const float wt = weights[(long)r * S + s];
      dweights[(long)r * S + s] += dEx * pts_seq[o] + dEy * pts_seq[o + 1] + dEz * pts_seq[o + 2];
      dseq[o] = wt * dEx; dseq[o + 1] = wt * dEy; dseq[o + 2] = wt * dEz;
    }
  }
}
extern "C" int dyn_render_flows_bwd(const float* dflows, const float* weights, const float* pts_seq, const float* proj, int R, int S, int V,
                                    float* dweights, float* dseq, void* stream) {
  DYN_REQUIRE(dflows && weights && pts_seq && proj && dweights && dseq && R > 0 && S > 0 && V > 0, "dyn_render_flows_bwd: bad argument");
  DYN_LAUNCH(DYN_K_RENDER_FLOWS, "dyn_render_flows_bwd", k_render_flows_bwd, dim3(dyn_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, dflows, weights,
             pts_seq, proj, R, S, V, dweights, dseq);
  return 0;
}

__global__ void __launch_bounds__(256) k_expected_scene_flow(const float* __restrict__ weights, const float* __restrict__ coeff,
                                                             const float* __restrict__ basis, int R, int S, int B, int row_p, int row_m,
                                                             int row_ref, float* __restrict__ out) {
  // (round 6: a ray's S x 3 B coefficients are one contiguous run -- 64 samples at a time they are read coalesced into the wave's own LDS block at an odd
  // sample stride; a lane reading its sample's 72-byte record straight from global memory ran this kernel at 76 us per 8192 x 128 samples.  Same sums, same order.)
  const int lane = dyn_lane(), wave = dyn_wave();
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;  // (whole waves leave; no workgroup barrier below)
  const int C3 = 3 * B, CS = C3 | 1;
  float* cw = reinterpret_cast<float*>(dyn_smem) + (size_t)wave * 64 * CS;
  float ap[3] = {0.f, 0.f, 0.f}, am[3] = {0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int ns = S - s0 < 64 ? S - s0 : 64;
    const float* src = coeff + ((long)r * S + s0) * C3;
    for (int e = lane; e < ns * C3; e += 64) cw[(e / C3) * CS + (e % C3)] = src[e];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int s = s0 + lane;
    const float wt = lane < ns ? weights[(long)r * S + s] : 0.f;
    const float* c = cw + (lane < ns ? lane : 0) * CS;
    for (int a = 0; a < 3 && lane < ns; ++a) {
      float t0 = 0.f, tp = 0.f, tm = 0.f;
      for (int b = 0; b < B; ++b) {
        const float cb = c[a * B + b];
        t0 += cb * basis[(long)row_ref * B + b];
        tp += cb * basis[(long)row_p * B + b];
        tm += cb * basis[(long)row_m * B + b];
      }
      ap[a] += wt * (tp - t0);
      am[a] += wt * (tm - t0);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  for (int a = 0; a < 3; ++a) {
    ap[a] = wave_sum(ap[a]);
    am[a] = wave_sum(am[a]);
    if (lane == 0) out[r * 3 + a] = fmaxf(ap[a], am[a]);
  }
}
extern "C" int dyn_expected_scene_flow(const float* weights, const float* coeff, const float* basis, int R, int S, int B, int row_p, int row_m,
                                       int row_ref, float* exp_sf, void* stream) {
  DYN_REQUIRE(weights && coeff && basis && exp_sf && R > 0 && S > 0 && B > 0, "dyn_expected_scene_flow: bad argument");
  DYN_REQUIRE(row_p >= 0 && row_m >= 0 && row_ref >= 0, "dyn_expected_scene_flow: basis rows must be non-negative");
  DYN_LAUNCH(DYN_K_SCENE_FLOW, "dyn_expected_scene_flow", k_expected_scene_flow, dim3(dyn_cdiv(R, 4)), dim3(256), (size_t)4 * 64 * ((3 * B) | 1) * sizeof(float), (hipStream_t)stream, weights,
             coeff, basis, R, S, B, row_p, row_m, row_ref, exp_sf);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// a2 all-pixel rays of a target view (sample_ray.py:143-163): d = c2w[:3,:3] . inv(K[:3,:3]) . [u, v, 1], o = c2w[:3,3]
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_image_rays(const float* __restrict__ camera, int Ws, int n, int stride, float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // M = R . inv(K3) in double (the reference inverts K with LAPACK in fp32; agreement is to ~1e-7 relative, not bitwise)
  double K[9], Rm[9], Ki[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) { K[a * 3 + b] = camera[2 + a * 4 + b]; Rm[a * 3 + b] = camera[18 + a * 4 + b]; }
  const double det = K[0] * (K[4] * K[8] - K[5] * K[7]) - K[1] * (K[3] * K[8] - K[5] * K[6]) + K[2] * (K[3] * K[7] - K[4] * K[6]);
  Ki[0] = (K[4] * K[8] - K[5] * K[7]) / det; Ki[1] = (K[2] * K[7] - K[1] * K[8]) / det; Ki[2] = (K[1] * K[5] - K[2] * K[4]) / det;
  Ki[3] = (K[5] * K[6] - K[3] * K[8]) / det; Ki[4] = (K[0] * K[8] - K[2] * K[6]) / det; Ki[5] = (K[2] * K[3] - K[0] * K[5]) / det;
  Ki[6] = (K[3] * K[7] - K[4] * K[6]) / det; Ki[7] = (K[1] * K[6] - K[0] * K[7]) / det; Ki[8] = (K[0] * K[4] - K[1] * K[3]) / det;
  const float u = (float)((i % Ws) * stride), v = (float)((i / Ws) * stride);
  for (int a = 0; a < 3; ++a) {
    const float m0 = (float)(Rm[a * 3] * Ki[0] + Rm[a * 3 + 1] * Ki[3] + Rm[a * 3 + 2] * Ki[6]);
    const float m1 = (float)(Rm[a * 3] * Ki[1] + Rm[a * 3 + 1] * Ki[4] + Rm[a * 3 + 2] * Ki[7]);
    const float m2 = (float)(Rm[a * 3] * Ki[2] + Rm[a * 3 + 1] * Ki[5] + Rm[a * 3 + 2] * Ki[8]);
    rays_d[i * 3 + a] = fmaf(m0, u, fmaf(m1, v, m2));
    rays_o[i * 3 + a] = camera[18 + a * 4 + 3];
  }
}
extern "C" int dyn_image_rays(const float* camera, int H, int W, int render_stride, float* rays_o, float* rays_d, void* stream) {
  DYN_REQUIRE(camera && rays_o && rays_d && H > 0 && W > 0 && render_stride > 0, "dyn_image_rays: bad argument");
  const int Hs = (H + render_stride - 1) / render_stride, Ws = (W + render_stride - 1) / render_stride;
  DYN_LAUNCH(DYN_K_IMAGE_RAYS, "dyn_image_rays", k_image_rays, dim3(dyn_cdiv((long)Hs * Ws, 256)), dim3(256), 0, (hipStream_t)stream, camera, Ws,
             Hs * Ws, render_stride, rays_o, rays_d);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// sample masks
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_sample_mask(const float* __restrict__ mask, long RS, int V, float thresh, float* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RS) return;
  float s = 0.f;
  for (int v = 0; v < V; ++v) s += mask[i * V + v];
  out[i] = (s > thresh) ? 1.0f : 0.0f;
}
extern "C" int dyn_sample_mask(const float* mask, int RS, int V, float thresh, float* pix_mask, void* stream) {
  DYN_REQUIRE(mask && pix_mask && RS > 0 && V > 0, "dyn_sample_mask: bad argument");
  DYN_LAUNCH(DYN_K_SAMPLE_MASK, "dyn_sample_mask", k_sample_mask, dim3(dyn_cdiv(RS, 256)), dim3(256), 0, (hipStream_t)stream, mask, (long)RS, V, thresh, pix_mask);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// K2: alpha compositing, one wavefront per ray   (render_ray.py:134-330)
// lanes = samples: sigma->alpha elementwise, transmittance = wave-level exclusive product scan, sums = wave reductions.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigma_to_alpha(float sigma, bool last) {
  // nn.Softplus(beta=1, threshold=20); dists = 1 except the last sample = 1e10 (render_ray.py:154-182)
  float sp = (sigma > 20.0f) ? sigma : log1pf(expf(sigma));
  float dist = last ? 1e10f : 1.0f;
  return 1.0f - expf(-sp * dist);
}

__global__ void __launch_bounds__(256) k_composite(DynCompositeParams p) {
  const int lane = dyn_lane();
  const int r = blockIdx.x * 4 + dyn_wave();
  const bool two = p.raw_static != nullptr;
  // every lane of a wave shares r, so the whole wave leaves together (no barrier is used in this kernel)
  if (r >= p.R) return;
  float carry = 1.0f;
  float acc_rgb[3] = {0.f, 0.f, 0.f}, acc_st[3] = {0.f, 0.f, 0.f}, acc_dy[3] = {0.f, 0.f, 0.f};
  float acc_depth = 0.f, cnt_dy = 0.f, cnt_st = 0.f;
  for (int s0 = 0; s0 < p.S; s0 += 64) {
    const int s = s0 + lane;
    const bool ok = s < p.S;
    const long i = (long)r * p.S + s;
    float4 rd = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(0.f, 0.f, 0.f, 0.f);
    float z = 0.f, a_dy = 0.f, a_st = 0.f, alpha = 0.f;
    if (ok) {
      rd = reinterpret_cast<const float4*>(p.raw_dy)[i];
      z = p.z_vals[i];
      a_dy = sigma_to_alpha(rd.w, s == p.S - 1);
      cnt_dy += p.pix_mask_dy[i];
      if (two) {
        rs = reinterpret_cast<const float4*>(p.raw_static)[i];
        a_st = sigma_to_alpha(rs.w, s == p.S - 1);
        alpha = 1.0f - (1.0f - a_st) * (1.0f - a_dy);
        cnt_st += p.pix_mask_st[i];
      } else {
        alpha = a_dy;
      }
    }
    const float f = ok ? ((1.0f - alpha) + 1e-10f) : 1.0f;
    const float incl = wave_inclusive_prod(f, lane);
    float excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0f;
    const float T = carry * excl;
    carry = carry * __shfl(incl, 63);
    if (ok) {
      const float w = alpha * T;
      p.weights[i] = w;
      if (p.alpha) p.alpha[i] = alpha;
      acc_depth += w * z;
      if (two) {
        const float wd = a_dy * T, ws = a_st * T;
        if (p.alpha_dy) p.alpha_dy[i] = a_dy;
        if (p.weights_dy) p.weights_dy[i] = wd;
        if (p.weights_st) p.weights_st[i] = ws;
        acc_dy[0] += wd * rd.x; acc_dy[1] += wd * rd.y; acc_dy[2] += wd * rd.z;
        acc_st[0] += ws * rs.x; acc_st[1] += ws * rs.y; acc_st[2] += ws * rs.z;
      } else {
        acc_rgb[0] += w * rd.x; acc_rgb[1] += w * rd.y; acc_rgb[2] += w * rd.z;
      }
    }
  }
  acc_depth = wave_sum(acc_depth);
  cnt_dy = wave_sum(cnt_dy);
  cnt_st = wave_sum(cnt_st);
  for (int c = 0; c < 3; ++c) {
    acc_rgb[c] = wave_sum(acc_rgb[c]);
    acc_dy[c] = wave_sum(acc_dy[c]);
    acc_st[c] = wave_sum(acc_st[c]);
  }
  if (lane == 0) {
    for (int c = 0; c < 3; ++c) {
      p.rgb[r * 3 + c] = two ? (acc_dy[c] + acc_st[c]) : acc_rgb[c];
      if (two && p.rgb_static) p.rgb_static[r * 3 + c] = acc_st[c];
      if (two && p.rgb_dy) p.rgb_dy[r * 3 + c] = acc_dy[c];
    }
    p.depth[r] = acc_depth;
    p.ray_mask[r] = ((cnt_dy > 8.0f) || (two && cnt_st > 8.0f)) ? 1.0f : 0.0f;
  }
}

extern "C" int dyn_composite(const DynCompositeParams* p, void* stream) {
  DYN_REQUIRE(p && p->raw_dy && p->z_vals && p->pix_mask_dy && p->rgb && p->depth && p->ray_mask && p->weights,
              "dyn_composite: null pointer");
  DYN_REQUIRE(p->raw_static == nullptr || p->pix_mask_st != nullptr, "dyn_composite: pix_mask_st missing");
  DYN_REQUIRE(p->R > 0 && p->S > 0, "dyn_composite: empty problem");
  DYN_LAUNCH(DYN_K_COMPOSITE, "dyn_composite", k_composite, dim3(dyn_cdiv(p->R, 4)), dim3(256), 0, (hipStream_t)stream, *p);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// K4: importance sampling + sorted merge, one lane per ray   (render_ray.py:19-64, :790-821)
// The cdf is a sequential per-ray prefix sum accumulated in double and rounded to fp32 per element, which is what torch.cumsum
// does on the CPU, so the inverse-CDF indices reproduce the reference's exactly given the same weights.  Per-ray scratch lives in LDS, interleaved by lane.
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_fine_samples(DynFineSampleParams p, int T) {
  float* lds = reinterpret_cast<float*>(dyn_smem);
  const int t = threadIdx.x;
  const int r = blockIdx.x * T + t;
  const int S = p.S, N = p.N, M = S - 2, NB = S - 1;
  float* cdf = lds;                 // [NB][T]   cdf_0..cdf_M
  float* bins = lds + (long)NB * T; // [NB][T]
  float* news = bins + (long)NB * T;  // [N][T] the new depths, kept ascending in z
  if (r >= p.R) return;  // no barrier in this kernel
  const float* z = p.z_vals + (long)r * S;
  const float* w = p.weights + (long)r * S;
  // normaliser: the fp32 addends summed in double and rounded once, i.e. the correctly rounded sum that torch.sum's
  // multi-accumulator vector reduction approximates to within an ulp (a plain fp32 running sum is ~1e-6 off when the
  // weights span many orders of magnitude, which an empty bin of mass 1e-5 amplifies to several percent of its width)
  double total_d = 0.0;
  for (int j = 0; j < M; ++j) {
    float wj = (p.inv_uniform ? w[S - 2 - j] : w[1 + j]) + 1e-5f;
    total_d += (double)wj;
  }
  const float total = (float)total_d;
  for (int j = 0; j < NB; ++j) {
    float b;
    if (p.inv_uniform) {
      int k = S - 2 - j;
      b = 0.5f * (1.0f / z[k + 1] + 1.0f / z[k]);
    } else {
      b = 0.5f * (z[j + 1] + z[j]);
    }
    bins[j * T + t] = b;
  }
  // torch.cumsum on the CPU accumulates fp32 inputs in double (at::acc_type<float, false>) and rounds every prefix to fp32
  double c = 0.0;
  cdf[t] = 0.f;
  for (int j = 0; j < M; ++j) {
    float wj = (p.inv_uniform ? w[S - 2 - j] : w[1 + j]) + 1e-5f;
    c += (double)(wj / total);
    cdf[(j + 1) * T + t] = (float)c;
  }
  for (int n = 0; n < N; ++n) {
    float u;
    if (p.u != nullptr) {
      u = p.u[(long)r * N + n];
    } else {
      // torch.linspace(0, 1, N): start + i*step below the midpoint, end - (N-1-i)*step above it
      float step = 1.0f / (float)(N - 1);
      u = (n < N / 2) ? ((float)n * step) : (1.0f - step * (float)(N - 1 - n));
    }
    // above = #{j < M : cdf_j <= u}  (render_ray.py:38-39 counts with a loop of M compares).  The cdf is non-decreasing (prefix sums of
    // non-negative terms, rounded monotonically), so the count is an upper bound found by bisection: same integer, log2(M) probes.
    int lo_i = 0, hi_i = M;
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (u >= cdf[mid * T + t]) lo_i = mid + 1; else hi_i = mid;
    }
    const int above = lo_i;
    int below = above - 1 < 0 ? 0 : above - 1;
    float c0 = cdf[below * T + t], c1 = cdf[above * T + t];
    float b0 = bins[below * T + t], b1 = bins[above * T + t];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    float tt = (u - c0) / denom;
    float smp = b0 + tt * (b1 - b0);
    if (p.inv_uniform) smp = 1.0f / smp;
    if (p.inds) p.inds[(long)r * N + n] = above;
    if (p.z_samples) p.z_samples[(long)r * N + n] = smp;
    news[(p.inv_uniform ? N - 1 - n : n) * T + t] = smp;  // ascending u is ascending 1/z: descending z
  }
  // sorted union (values only, so any correct sort equals torch.sort's values): the new depths are insertion-sorted among themselves
  // (already ascending for det=True, where u is a ramp and the map u -> depth is monotone up to rounding), then merged with the coarse ones
  for (int n = 1; n < N; ++n) {
    const float v = news[n * T + t];
    int k = n;
    while (k > 0 && news[(k - 1) * T + t] > v) {
      news[k * T + t] = news[(k - 1) * T + t];
      --k;
    }
    news[k * T + t] = v;
  }
  {
    float* zo = p.z_out + (long)r * (S + N);
    int a = 0, b = 0;
    for (int i = 0; i < S + N; ++i) {
      const bool take_a = b >= N || (a < S && z[a] <= news[b * T + t]);
      zo[i] = take_a ? z[a] : news[b * T + t];
      a += take_a ? 1 : 0;
      b += take_a ? 0 : 1;
    }
  }
}

extern "C" int dyn_fine_samples(const DynFineSampleParams* p, void* stream) {
  DYN_REQUIRE(p && p->z_vals && p->weights && p->z_out, "dyn_fine_samples: null pointer");
  DYN_REQUIRE(p->R > 0 && p->S > 2 && p->N > 1, "dyn_fine_samples: need R>0, S>2, N>1");
  const size_t per_thread = (size_t)(2 * (p->S - 1) + p->N) * 4;
  int T = 64;
  while (T > 1 && per_thread * T > 144 * 1024) T >>= 1;
  DYN_REQUIRE(per_thread * T <= 144 * 1024, "dyn_fine_samples: S+N too large for LDS");
  size_t shmem = (per_thread * T + 15) & ~(size_t)15;
  DYN_LAUNCH(DYN_K_FINE_SAMPLES, "dyn_fine_samples", k_fine_samples, dim3(dyn_cdiv(p->R, T)), dim3(T), shmem, (hipStream_t)stream, *p, T);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Helper exports of the reference's module surface (render_ray.py:19-64 sample_pdf, :372-396 Pluecker coordinates).  The render functions
// do not call these (k_fine_samples fuses sample_pdf with the merge, the network kernels form the Pluecker coordinates in registers); they
// exist so that code importing the helpers from ibrnet.render_ray finds them with the same semantics.
// ---------------------------------------------------------------------------------------------------------------
// sample_pdf: bins [R, M+1], weights [R, M] (1e-5 is added IN PLACE like the reference does, render_ray.py:23), u [R, N] or NULL (det=True) -> samples [R, N]
__global__ void k_sample_pdf(const float* __restrict__ bins, float* __restrict__ weights, const float* __restrict__ u_in, int R, int M, int N,
                             float* __restrict__ samples, int T) {
  float* lds = reinterpret_cast<float*>(dyn_smem);
  const int t = threadIdx.x;
  const int r = blockIdx.x * T + t;
  if (r >= R) return;
  float* cdf = lds;  // [M + 1][T]
  float* w = weights + (long)r * M;
  const float* b = bins + (long)r * (M + 1);
  double total_d = 0.0;
  for (int j = 0; j < M; ++j) {
    const float wj = w[j] + 1e-5f;
    w[j] = wj;
    total_d += (double)wj;
  }
  const float total = (float)total_d;
  double c = 0.0;
  cdf[t] = 0.f;
  for (int j = 0; j < M; ++j) {
    c += (double)(w[j] / total);
    cdf[(j + 1) * T + t] = (float)c;
  }
  for (int n = 0; n < N; ++n) {
    float u;
    if (u_in != nullptr) {
      u = u_in[(long)r * N + n];
    } else {
      const float step = 1.0f / (float)(N - 1);
      u = (n < N / 2) ? ((float)n * step) : (1.0f - step * (float)(N - 1 - n));
    }
    int lo_i = 0, hi_i = M;
    while (lo_i < hi_i) {
      const int mid = (lo_i + hi_i) >> 1;
      if (u >= cdf[mid * T + t]) lo_i = mid + 1; else hi_i = mid;
    }
    const int above = lo_i, below = above - 1 < 0 ? 0 : above - 1;
    const float c0 = cdf[below * T + t], c1 = cdf[above * T + t];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    const float tt = (u - c0) / denom;
    samples[(long)r * N + n] = b[below] + tt * (b[above] - b[below]);
  }
}
extern "C" int dyn_sample_pdf(const float* bins, float* weights, const float* u, int R, int M, int N, float* samples, void* stream) {
  DYN_REQUIRE(bins && weights && samples && R > 0 && M > 0 && N > 1, "dyn_sample_pdf: bad argument");
  const size_t per_thread = (size_t)(M + 1) * 4;
  int T = 64;
  while (T > 1 && per_thread * T > 144 * 1024) T >>= 1;
  DYN_REQUIRE(per_thread * T <= 144 * 1024, "dyn_sample_pdf: too many bins for LDS");
  DYN_LAUNCH(DYN_K_FINE_SAMPLES, "dyn_sample_pdf", k_sample_pdf, dim3(dyn_cdiv(R, T)), dim3(T), (per_thread * T + 15) & ~(size_t)15, (hipStream_t)stream, bins,
             weights, u, R, M, N, samples, T);
  return 0;
}

// [normalize(d), o x normalize(d)] per target ray (render_ray.py:372-377) and per (sample, source view) (:380-396, output [R,S,V,6]).  The product runs
// over the axis torch.cross(dim=None) picks for the call's shape -- the first of size 3 (csrc/dyn_device.h) --, like in the network kernels.
__global__ void k_plucker_ref(const float* __restrict__ ray_o, const float* __restrict__ ray_d, int R, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float dx, dy, dz;
  dyn_unit3(ray_d[r * 3], ray_d[r * 3 + 1], ray_d[r * 3 + 2], dx, dy, dz);
  const float ox = ray_o[r * 3], oy = ray_o[r * 3 + 1], oz = ray_o[r * 3 + 2];
  float m[3] = {oy * dz - oz * dy, oz * dx - ox * dz, ox * dy - oy * dx};
  if (dyn_ref_cross_axis(R) != DYN_CROSS_XYZ) dyn_ref_moment_over_rays(ray_o, ray_d, r, m);
  float* o = out + (long)r * 6;
  o[0] = dx; o[1] = dy; o[2] = dz;
  o[3] = m[0]; o[4] = m[1]; o[5] = m[2];
}
__global__ void k_plucker_src(const float* __restrict__ pts, long pts_view_stride, const float* __restrict__ cams, int R, int S, int V, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_pts = (long)R * S;
  if (i >= n_pts * V) return;
  const long pnt = i / V;
  const int v = (int)(i - pnt * V);
  auto ctr_at = [&](int vv, float (&c3)[3]) { const float* c = cams + (long)vv * 34 + 18; c3[0] = c[3]; c3[1] = c[7]; c3[2] = c[11]; };  // c2w
  auto pts_at = [&](int vv, int rr, int ss, float (&q)[3]) {
    const float* s3 = pts + (long)vv * pts_view_stride + ((long)rr * S + ss) * 3;
    q[0] = s3[0]; q[1] = s3[1]; q[2] = s3[2];
  };
  const int ray = (int)(pnt / S), smp = (int)(pnt - (long)ray * S);
  float c3[3], q[3], dx, dy, dz;
  ctr_at(v, c3);
  pts_at(v, ray, smp, q);
  dyn_unit3(q[0] - c3[0], q[1] - c3[1], q[2] - c3[2], dx, dy, dz);
  float m[3] = {c3[1] * dz - c3[2] * dy, c3[2] * dx - c3[0] * dz, c3[0] * dy - c3[1] * dx};
  const int axis = dyn_src_cross_axis(V, R, S);
  if (axis != DYN_CROSS_XYZ) dyn_src_moment_over_axis(axis, v, ray, smp, pts_at, ctr_at, m);
  float* o = out + i * 6;
  o[0] = dx; o[1] = dy; o[2] = dz;
  o[3] = m[0]; o[4] = m[1]; o[5] = m[2];
}
extern "C" int dyn_plucker_ref(const float* ray_o, const float* ray_d, int R, float* out, void* stream) {
  DYN_REQUIRE(ray_o && ray_d && out && R > 0, "dyn_plucker_ref: bad argument");
  DYN_LAUNCH(DYN_K_IMAGE_RAYS, "dyn_plucker_ref", k_plucker_ref, dim3(dyn_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, ray_o, ray_d, R, out);
  return 0;
}
extern "C" int dyn_plucker_src(const float* pts, int per_view_pts, const float* cams, int R, int S, int V, float* out, void* stream) {
  DYN_REQUIRE(pts && cams && out && R > 0 && S > 0 && V > 0, "dyn_plucker_src: bad argument");
  const long n_pts = (long)R * S;
  DYN_LAUNCH(DYN_K_IMAGE_RAYS, "dyn_plucker_src", k_plucker_src, dim3(dyn_cdiv(n_pts * V, 256)), dim3(256), 0, (hipStream_t)stream, pts,
             per_view_pts ? n_pts * 3 : 0L, cams, R, S, V, out);
  return 0;
}
