/* dynibar_hip.h -- C ABI of libdynibar_hip.so: the MI355X (gfx950) per-ray renderer kernels.
 *
 * The reference (google/dynibar) has no FFI layer: its hot path is Python calling PyTorch eager ops.  Each entry
 * point below replaces one group of those op sites; the citation gives the reference lines (paths under
 * /root/reference/ibrnet/).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions: every pointer is a DEVICE pointer to contiguous fp32 (or int32 where noted) unless marked HOST;
 * `stream` is a hipStream_t passed as void*; calls are stream-ordered, allocate nothing and keep no global state;
 * return 0 on success or a negative DYN_E_* code, with a message available from dyn_last_error().
 * All tensors are row-major with the shapes written next to each field.
 */
#ifndef DYNIBAR_HIP_H_
#define DYNIBAR_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYN_ABI_VERSION 1
#define DYN_E_INVALID (-1)  /* bad argument (NULL pointer, unsupported shape) */
#define DYN_E_LAUNCH (-2)   /* HIP launch error */

int dyn_abi_version(void);
const char* dyn_last_error(void);

/* ---- a8 (projection.py:42-47) camera preparation ----------------------------------------------------------
 * cams [V,34] = [h, w, K(4x4), c2w(4x4)] -> proj [V,16]: rows 0..2 of K.inv(c2w) (12 floats), then the source camera
 * centre c2w[:3,3] (3 floats), then 0.  query_cam [34] -> query_center[4] (c2w[:3,3], 0).  Replaces torch.inverse+bmm. */
int dyn_prepare_cameras(const float* cams, int V, const float* query_cam, float* proj, float* query_center, void* stream);

/* ---- featmaps [V,F,Hf,Wf] (NCHW, feature_network.py:302-311) -> channels-last [V,Hf,Wf,F] for 128-byte taps --- */
int dyn_nchw_to_nhwc(const float* src, float* dst, int V, int F, int Hf, int Wf, void* stream);

/* ---- a5 sample_along_camera_ray (render_ray.py:67-131) -------------------------------------------------------
 * depth_range: DEVICE [2] = (near, far).  t_rand: [R,S] uniform draws for det=False, NULL for det=True (the host owns the
 * RNG so tests can inject the reference's draws).  Outputs z_vals [R,S], s_vals [R,S] (may be NULL), pts [R,S,3] (may be NULL). */
typedef struct {
  int R, S;
  int inv_uniform;
  const float* ray_o;       /* [R,3] */
  const float* ray_d;       /* [R,3] */
  const float* depth_range; /* [2] */
  const float* t_rand;      /* [R,S] or NULL */
  float* z_vals;            /* [R,S] */
  float* s_vals;            /* [R,S] or NULL */
  float* pts;               /* [R,S,3] or NULL */
} DynSampleParams;
int dyn_sample_along_ray(const DynSampleParams* p, void* stream);

/* ---- z_vals -> pts, s_vals for the fine pass (render_ray.py:822-831, z_to_s :399-404) ---------------------- */
int dyn_points_from_z(const float* ray_o, const float* ray_d, const float* z_vals, const float* depth_range, int R, int S,
                      float* pts, float* s_vals, void* stream);

/* ---- a8-a11 Projector.compute_with_motions (projection.py:103-176), fused ----------------------------------
 * One kernel: K.inv(c2w) projection, clamp, in-front/in-bounds mask, bilinear zero-padded align_corners taps of the
 * RGB images and of the feature maps at the same normalised location, ray-direction difference.
 * Sample points come either from (ray_o, ray_d, z_vals) [pts_st = o + z d] or from an explicit pts_st array; the
 * per-view points xyz are pts_st itself (static branch, xyz = NULL), an explicit [V,R,S,3] array (scene motion) or -- traj_coeff != NULL, the fused form of
 * compute_traj_pts (render_ray.py:361-369, :691-725) -- formed in the kernel from the motion coefficients: view v sees pts_st + (traj(traj_rows[v]) - traj(traj_ref)),
 * traj(row)[a] = sum_b coeff[.., a B + b] basis[row, b], a negative row = no displacement (virtual views); the same arithmetic, bit for bit, as dyn_trajectory_points,
 * whose [V,R,S,3] output then never exists.  traj_rows is a DEVICE array of V ints. */
typedef struct {
  int R, S, V;
  int H, W;                 /* source image size (tensor dims) */
  int Hf, Wf, F;            /* feature map size / channels (F % 4 == 0) */
  float img_h, img_w;       /* train_cameras[0][:2], used by normalize()/inbound() (projection.py:136) */
  const float* ray_o;       /* [R,3]   (used when pts_st == NULL) */
  const float* ray_d;       /* [R,3] */
  const float* z_vals;      /* [R,S] */
  const float* pts_st;      /* [R,S,3] or NULL */
  const float* xyz;         /* [V,R,S,3] or NULL */
  const float* proj;        /* [V,16] from dyn_prepare_cameras */
  const float* query_center;/* [4] */
  const float* src_rgb;     /* [V,H,W,3] */
  const float* feat_cl;     /* [V,Hf,Wf,F] channels-last */
  float* rgb_feat;          /* [R,S,V,3+F] */
  float* ray_diff;          /* [R,S,V,4], or NULL when the caller has no use for it (the dynamic branch: DynibarDynamic takes none) */
  float* mask;              /* [R,S,V] (the reference's trailing singleton dim is a view) */
  float* pix_mask;          /* [R,S] or NULL: 1 where more than pix_mask_thresh views see the sample (render_ray.py:736-741), else 0 */
  float pix_mask_thresh;
  const float* traj_coeff;  /* [R,S,3 B] or NULL (then the fields below are ignored); needs pts_st, excludes xyz */
  const float* traj_basis;  /* [frames, B] */
  const int* traj_rows;     /* DEVICE [V]: basis row of each view, negative = the undisplaced point */
  int traj_B, traj_ref;     /* basis functions per axis; basis row of the reference time */
} DynProjectGatherParams;
int dyn_project_gather(const DynProjectGatherParams* p, void* stream);

/* ---- a20/a21 raw2outputs_vanilla / raw2outputs (render_ray.py:134-330): one wavefront per ray ------------------
 * raw_static == NULL selects the vanilla (single-branch) form.  pix_mask_*: [R,S] 0/1 floats (the "at least 2 observations"
 * sample masks).  Per-sample outputs may be NULL when the caller does not need them. */
typedef struct {
  int R, S;
  const float* raw_dy;      /* [R,S,4]  (vanilla: the single raw input) */
  const float* raw_static;  /* [R,S,4] or NULL */
  const float* z_vals;      /* [R,S] */
  const float* pix_mask_dy; /* [R,S] */
  const float* pix_mask_st; /* [R,S] or NULL */
  float* rgb;               /* [R,3] */
  float* rgb_static;        /* [R,3] or NULL */
  float* rgb_dy;            /* [R,3] or NULL */
  float* depth;             /* [R] */
  float* ray_mask;          /* [R] 0/1 */
  float* weights;           /* [R,S] */
  float* alpha;             /* [R,S] or NULL */
  float* alpha_dy;          /* [R,S] or NULL */
  float* weights_dy;        /* [R,S] or NULL */
  float* weights_st;        /* [R,S] or NULL */
} DynCompositeParams;
int dyn_composite(const DynCompositeParams* p, void* stream);

/* ---- sample masks: pixel_mask[r,s] = (sum_v mask[r,s,v]) > thresh (render_ray.py:736-741) -------------------- */
int dyn_sample_mask(const float* mask, int RS, int V, float thresh, float* pix_mask, void* stream);

/* ---- a6/a7 fine-sample assembly (render_ray.py:19-64, :790-821): pdf -> cdf -> inverse-CDF -> merge + sort ----
 * weights: coarse weights [R,S]; the kernel drops the two end samples, adds 1e-5, builds the cdf with a sequential
 * per-ray prefix sum accumulated in double and rounded to fp32 per element (what torch.cumsum does on the CPU), inverts it at u (NULL = linspace(0,1,N) i.e. det=True) with the
 * reference's count-of-(u >= cdf_i) rule, and writes the sorted union of coarse and new depths.  inds (optional) returns
 * the reference's `above_inds` for the bit-exact index check. */
typedef struct {
  int R, S, N;              /* S coarse samples, N = N_importance */
  int inv_uniform;
  const float* z_vals;      /* [R,S] */
  const float* weights;     /* [R,S] */
  const float* u;           /* [R,N] or NULL */
  float* z_out;             /* [R,S+N] sorted ascending */
  float* z_samples;         /* [R,N] or NULL: the new depths before the sort */
  int32_t* inds;            /* [R,N] or NULL */
} DynFineSampleParams;
int dyn_fine_samples(const DynFineSampleParams* p, void* stream);

/* ---- a17 DynibarStatic.forward (mlp_network.py:423-527) incl. a12 Pluecker coordinates (render_ray.py:372-396), a13 Fourier
 * features (:530-555), a18 ray attention (:13-31,:56-104), a19 weighted mean/variance over views (:115-119) ----------------
 * Weights: dyn_static_net_pack re-lays the module's state dict out as MFMA operand tiles.  `tensors` are HOST pointers to the
 * row-major fp32 tensors in this order (state-dict names):
 *   ray_dir_fc.0.weight .0.bias ray_dir_fc.2.weight .2.bias ref_feature_fc.0.weight .0.bias base_fc.0.weight .0.bias base_fc.2.weight
 *   .2.bias vis_fc.0.weight .0.bias vis_fc.2.weight .2.bias vis_fc2.0.weight .0.bias vis_fc2.2.weight .2.bias geometry_fc.0.weight
 *   .0.bias geometry_fc.2.weight .2.bias ray_attention.w_qs.weight ray_attention.w_ks.weight ray_attention.w_vs.weight
 *   ray_attention.fc.weight ray_attention.layer_norm.weight ray_attention.layer_norm.bias out_geometry_fc.0.weight .0.bias
 *   out_geometry_fc.2.weight .2.bias rgb_fc.0.weight .0.bias rgb_fc.2.weight .2.bias rgb_fc.4.weight .4.bias s
 * (39 tensors).  F = feature channels of the maps (must be 32).  blob: HOST buffer of dyn_static_net_blob_floats() floats; the
 * caller copies it to the device once per model. */
#define DYN_STATIC_NUM_TENSORS 39
size_t dyn_static_net_blob_floats(void);
int dyn_static_net_pack(const float* const* tensors, int F, float* blob, size_t blob_floats);
size_t dyn_static_net_workspace_bytes(int R, int S, int V);
typedef struct {
  int R, S, V;               /* rays, samples per ray (<= 256), source views (<= 32) */
  int anti_alias_pooling;    /* args.anti_alias_pooling (mlp_network.py:462) */
  int mask_rgb;              /* args.mask_rgb (:457) */
  const float* blob;         /* DEVICE copy of the packed weights */
  const float* ray_o;        /* [R,3] */
  const float* ray_d;        /* [R,3] */
  const float* pts;          /* [R,S,3] */
  const float* rgb_feat;     /* [R,S,V,35] from dyn_project_gather */
  const float* ray_diff;     /* [R,S,V,4] */
  const float* mask;         /* [R,S,V] */
  const float* centers;      /* [V,16]: the proj array of dyn_prepare_cameras (source camera centres at [12..14]) */
  float* raw;                /* [R,S,4] = (r, g, b, sigma) */
  void* workspace;           /* DEVICE scratch of dyn_static_net_workspace_bytes(R,S,V) bytes */
  size_t workspace_bytes;
} DynStaticNetParams;
int dyn_static_net(const DynStaticNetParams* p, void* stream);

/* ---- a16 DynibarDynamic.forward (mlp_network.py:236-316) ---------------------------------------------------------------
 * tensors for dyn_dynamic_net_pack (HOST pointers, state-dict names, 40 tensors):
 *   ray_dir_fc.0.weight .0.bias ray_dir_fc.2.weight .2.bias base_fc.0.weight .0.bias base_fc.2.weight .2.bias vis_fc.0.weight .0.bias
 *   vis_fc.2.weight .2.bias vis_fc2.0.weight .0.bias vis_fc2.2.weight .2.bias geometry_fc.0.weight .0.bias geometry_fc.2.weight .2.bias
 *   ray_attention.w_qs.weight ray_attention.w_ks.weight ray_attention.w_vs.weight ray_attention.fc.weight
 *   ray_attention.layer_norm.weight ray_attention.layer_norm.bias ref_pts_fc.0.weight .0.bias ref_pts_fc.2.weight .2.bias
 *   out_geometry_fc.0.weight .0.bias out_geometry_fc.2.weight .2.bias rgb_fc.0.weight .0.bias rgb_fc.2.weight .2.bias rgb_fc.4.weight .4.bias
 * The net ignores ray_diff and time_diff (anti_alias_pooling is hard-wired off, mlp_network.py:135), so they are not inputs. */
#define DYN_DYNAMIC_NUM_TENSORS 40
size_t dyn_dynamic_net_blob_floats(void);
int dyn_dynamic_net_pack(const float* const* tensors, int F, float* blob, size_t blob_floats);
size_t dyn_dynamic_net_workspace_bytes(int R, int S, int V);
typedef struct {
  int R, S, V;               /* rays, samples per ray (<= 256; must equal the module's n_samples), source views (<= 32) */
  float shift;               /* subtracted from sigma (DynibarMono builds its dynamic net with shift = 5, model.py:307) */
  const float* blob;         /* DEVICE copy of the packed weights */
  const float* ray_d;        /* [R,3] (normalised in the kernel, render_ray.py:655) */
  const float* pts;          /* [R,S,3] reference-time sample points */
  const float* rgb_feat;     /* [R,S,V,35] */
  const float* mask;         /* [R,S,V] */
  const float* time;         /* DEVICE [1]: reference time embedding */
  float* raw;                /* [R,S,4] */
  void* workspace;           /* DEVICE scratch of dyn_dynamic_net_workspace_bytes(R,S,V) bytes */
  size_t workspace_bytes;
} DynDynamicNetParams;
int dyn_dynamic_net(const DynDynamicNetParams* p, void* stream);

/* ---- a14 MotionMLP.forward (mlp_network.py:605-618) + the zeroing of the last samples' coefficients (render_ray.py:684) -------
 * tensors for dyn_motion_mlp_pack (HOST, 18 tensors): pts_linears.0.weight .0.bias ... pts_linears.7.weight .7.bias
 * coeff_linear.weight coeff_linear.bias.   coeff: [R,S,3*num_basis].  n_zero_last = int(round(S * 0.1)). */
#define DYN_MOTION_NUM_TENSORS 18
size_t dyn_motion_mlp_blob_floats(void);
int dyn_motion_mlp_pack(const float* const* tensors, int num_basis, float* blob, size_t blob_floats);
int dyn_motion_mlp(const float* blob, const float* pts, const float* time, int R, int S, int num_basis, int n_zero_last, float sf_mag_div,
                   float* coeff, void* stream);

/* ---- a15 compute_traj_pts + the per-view displaced points (render_ray.py:361-369, :686-709) -------------------------------
 * pts_seq[v] = pts + (traj(row_v) - traj(row_ref)), traj(row) = (sum_b cx_b basis[row,b], sum_b cy_b .., sum_b cz_b ..);
 * rows: HOST array of n_rows basis-row indices (frame + offset); a row index < 0 means "no motion" (virtual views, :988-989).
 * basis: DEVICE [num_frames, B].  pts_seq: DEVICE [n_rows, n_pts, 3]. */
int dyn_trajectory_points(const float* coeff, const float* basis, const float* pts, long n_pts, int B, const int* rows, int n_rows, int row_ref,
                          float* pts_seq, void* stream);

/* ---- a22 compute_optical_flow (render_ray.py:333-358): flows[v,r] = project_v(sum_s w[r,s] pts_seq[v,r,s]) - uv[r] ------------
 * proj: [V,16] from dyn_prepare_cameras (rows of K.inv(c2w)); flows: [V,R,2]. */
int dyn_render_flows(const float* weights, const float* pts_seq, const float* proj, const float* uv, int R, int S, int V, float* flows,
                     void* stream);
/* the same from the motion coefficients (the fused eval path: pts_seq [V,R,S,3] is never formed): pts [R,S,3], coeff [R,S,3 B], basis [frames,B],
 * rows_dev DEVICE [V] basis rows (negative: undisplaced), row_ref the reference time's row, B <= 16.  The expected point is formed through its linearity in the
 * coefficients (one pass over the samples for all views): equal to dyn_trajectory_points + dyn_render_flows up to the order of the fp32 sums (1e-7 relative) */
int dyn_render_flows_traj(const float* weights, const float* pts, const float* coeff, const float* basis, int B, const int* rows_dev, int row_ref,
                          const float* proj, const float* uv, int R, int S, int V, float* flows, void* stream);
/* ---- expected scene flow (render_ray.py:584-595 / :1086-1096): max(sum_s w (traj(row_p) - traj(row_ref)), sum_s w (traj(row_m) - traj(row_ref))) */
int dyn_expected_scene_flow(const float* weights, const float* coeff, const float* basis, int R, int S, int B, int row_p, int row_m, int row_ref,
                            float* exp_sf, void* stream);
/* ---- a2 RaySamplerSingleImage.get_rays_single_image (sample_ray.py:143-163): camera DEVICE [34]; rays_o, rays_d [(H/stride)*(W/stride),3] */
int dyn_image_rays(const float* camera, int H, int W, int render_stride, float* rays_o, float* rays_d, void* stream);

/* ---- helper exports of the reference's module surface (not on the render functions' own path, which fuses them) -------------------
 * sample_pdf (render_ray.py:19-64): bins [R,M+1], weights [R,M] (+1e-5 IN PLACE like the reference), u [R,N] or NULL (det) -> samples [R,N];
 * compute_ref_plucker_coordinate (:372-377): [R,6]; compute_src_plucker_coordinate (:380-396): pts [R,S,3] (per_view_pts = 0) or
 * [V,R,S,3] (per_view_pts = 1), cams [V,34] -> [R,S,V,6].  Both form the moment with torch.cross WITHOUT dim in the reference (:375, :392), which
 * crosses over the first axis of size 3 -- the views when V == 3, else the rays when R == 3, else the samples when S == 3, else xyz: these entry
 * points (and dyn_static_net / dyn_train_static_embed, which form the moments in their kernels) do the same, so R and S are separate arguments. */
int dyn_sample_pdf(const float* bins, float* weights, const float* u, int R, int M, int N, float* samples, void* stream);
int dyn_plucker_ref(const float* ray_o, const float* ray_d, int R, float* out, void* stream);
int dyn_plucker_src(const float* pts, int per_view_pts, const float* cams, int R, int S, int V, float* out, void* stream);
/* Projector.compute_projections (projection.py:32-59) and Projector.compute_angle (:61-101): xyz [V,n_pts,3] (+ xyz_st [V,n_pts,3], query_center [4] for
 * the angles), proj [V,16] from dyn_prepare_cameras -> pix [V,n_pts,2], in_front [V,n_pts] (0 / 1), ray_diff [V,n_pts,4]; either output group may be NULL. */
int dyn_project_points(const float* xyz, const float* xyz_st, const float* proj, const float* query_center, int V, long n_pts, float* pix,
                       float* in_front, float* ray_diff, void* stream);

/* ---- section 8(f)1: the 2-D feature encoder that feeds the path (feature_network.py:179-311, the executed part of ResNet.forward:
 * conv 7x7/2 -> InstanceNorm -> ReLU -> layer1 (3 BasicBlocks, reflect padding) -> 1x1 conv; 32 coarse + 32 fine channels at 1/4 resolution).
 * Channels-last throughout: images [N,H,W,3] as the data loader stores them, outputs [N,Hf,Wf,32] = the gather kernel's feat_cl layout.
 * dyn_encoder_pack: HOST pointers to the row-major fp32 tensors in this order (state-dict names of ibrnet.feature_network.ResNet):
 *   conv1.weight bn1.weight bn1.bias
 *   layer1.0.conv1.weight layer1.0.bn1.weight .bias layer1.0.conv2.weight layer1.0.bn2.weight .bias
 *   layer1.0.downsample.0.weight layer1.0.downsample.1.weight .bias
 *   layer1.1.conv1.weight layer1.1.bn1.weight .bias layer1.1.conv2.weight layer1.1.bn2.weight .bias
 *   layer1.2.conv1.weight layer1.2.bn1.weight .bias layer1.2.conv2.weight layer1.2.bn2.weight .bias
 *   out_conv.weight out_conv.bias */
size_t dyn_encoder_blob_floats(void);
int dyn_encoder_pack(const float* const* tensors, float* blob, size_t blob_floats);
size_t dyn_encoder_workspace_bytes(int N, int H, int W);
int dyn_encoder_out_size(int H, int W, int* Hf, int* Wf);
typedef struct {
  int N, H, W;              /* images */
  const float* blob;        /* DEVICE copy of the packed encoder */
  const float* images;      /* [N,H,W,3] */
  float* coarse;            /* [N,Hf,Wf,32]: channels 0..31 of out_conv (x_coarse) */
  float* fine;              /* [N,Hf,Wf,32]: channels 32..63 (x_fine) */
  void* workspace;
  size_t workspace_bytes;
} DynEncoderParams;
int dyn_encoder_forward(const DynEncoderParams* p, void* stream);

/* ---- training form of the encoder (autograd of ibrnet/feature_network.py:179-311; train.py:272-281 optimises feature_net): explicit
 * im2col + dyn_train_gemm for the convolutions, row kernels for what sits between; channels-last fp32 maps [N, H, W, C]. ----
 * dyn_enc_im2col: col[row, (ky * KW + kx) * C + c] = in[n, reflect(oy * stride - pad + ky), reflect(ox * stride - pad + kx), c],
 *   row = (n * Hout + oy) * Wout + ox (padding_mode='reflect' as conv3x3 / conv1 of the reference; pad 0 for the 1x1 convolutions); ldc a
 *   multiple of 4, columns KH * KW * C .. ldc - 1 are written as zeros;
 * dyn_enc_col2im: the adjoint, din += (a gather over the patches that contain a pixel; din holds whatever gradient the map already has);
 * dyn_enc_in_stats: stats[n][c] = {sum, sum of squares} over the HW pixels of image n (fp64, ZEROED by the caller), 64 channels;
 * dyn_enc_in_apply: y = [relu](InstanceNorm(x) * gamma + beta [+ res])  (nn.InstanceNorm2d(affine=True, eps=1e-5), BasicBlock.forward);
 * dyn_enc_in_bwd: its backward -- dyr = relu ? dy * (y > 0) : dy; dx = gamma * rstd * (dyr - mean(dyr) - xhat * mean(dyr * xhat));
 *   dres (may be NULL) = dyr (the residual branch); dgamma += sum dyr * xhat, dbeta += sum dyr; sums2 [N][64][2] doubles of workspace. */
int dyn_enc_im2col(const float* in, int N, int Hin, int Win, int C, int KH, int KW, int stride, int pad, int Hout, int Wout, float* col, long ldc,
                   void* stream);
int dyn_enc_col2im(const float* dcol, long ldc, int N, int Hin, int Win, int C, int KH, int KW, int stride, int pad, int Hout, int Wout, float* din,
                   void* stream);
int dyn_enc_in_stats(const float* x, int N, long HW, double* stats, void* stream);
int dyn_enc_in_apply(const float* x, const double* stats, const float* gamma, const float* beta, const float* res, int relu, int N, long HW, float* y,
                     void* stream);
int dyn_enc_in_bwd(const float* dy, const float* y, int relu, const float* x, const double* stats, const float* gamma, int N, long HW, double* sums2,
                   float* dx, float* dres, float* dgamma, float* dbeta, void* stream);

/* ---- how the network kernels of this build multiply (csrc/dyn_mlp.h): split terms = partial products kept per fp32 product (3 or 6;
 * 0 = native fp32 MFMA engine); split kind = what the operand parts are: 0 none (fp32 MFMA), 1 bf16 (3 terms: 16 mantissa bits per
 * operand; 6 terms: fp32-class), 2 IEEE half (3 terms: 22 mantissa bits per operand, fp32-class; the shipped engine) --------------- */
int dyn_mlp_split_terms(void);
int dyn_mlp_split_kind(void);

/* ---- self-test of the MFMA chain engine: y = elu(W elu(W x + b) + b), W [64,64], b [64] HOST; x, y [rows,64] DEVICE;
 * stream_buf: DEVICE scratch of 2 * 3 * 4096 floats ------------------------------------------------------------------- */
int dyn_mlp_selftest(const float* W, const float* b, const float* x, float* y, int rows, float* stream_buf, void* stream);

/* ---- per-kernel timing: when enabled, every kernel launched by this library is bracketed by HIP events on its launch stream.
 * dyn_profile_read synchronises the pending events, returns the summed milliseconds and launch counts per kernel slot
 * (dyn_profile_count() slots, named by dyn_profile_name) and resets the totals.  Not thread-safe; meant for bench.py. ------ */
int dyn_profile_enable(int on);
int dyn_profile_count(void);
const char* dyn_profile_name(int slot);
int dyn_profile_read(float* total_ms, int* launches);

/* ==== f3 (first slice): training of the static branch -- the reference's static bootstrap stage (train.py:116-199: loss on
 * ret['outputs_coarse_st']['rgb'], loss.backward() through raw2outputs_vanilla (render_ray.py:134-201), DynibarStatic.forward
 * (mlp_network.py:423-527) and Projector.compute's F.grid_sample (projection.py:160-167) into the module's parameters and the static
 * feature maps).  Replaces torch autograd for that graph.  The step is a sequence of the primitives below over row-major fp32
 * activation matrices kept in HBM (dynibar_amd/train_static.py holds the sequence); N = R S V rows (row = point * V + view),
 * P = R S points.  `ld*` are leading dimensions in floats, `*_stride` element strides of per-row scalars. =============================== */

/* C[M,N] (ldc) = epilogue(A . B): element (m,k) of A at A[m a_rs + k a_ks], element (k,n) of B at B[n b_rs + k b_ks] (one unit stride
 * each).  Forward of nn.Linear: A = X (a_ks = 1), B = W[N,K] (b_rs = ldw, b_ks = 1).  Data gradient dX = dZ W: A = dZ, B rows = input
 * features (b_rs = 1, b_ks = ldw).  Weight gradient dW = dZ^T X: A rows = output features (a_rs = 1, a_ks = ld_dz), B rows = input
 * features (b_rs = 1, b_ks = ldx), K = number of rows, k_split > 1 with accumulate = 2.
 * epilogue: + bias[n] + addend[(m / add_div) ld_add + n] (NULL to skip; with add_div = 1 the addend may be C itself -- C += A . B with plain
 * 16-byte stores, every element read and written by one thread), act (0 none, 1 ELU, 2 ReLU); accumulate 0 store, 1 or 2: += by fp32 atomics
 * (2 is required with k_split > 1).
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x16_f16: every operand split into two half parts (22 mantissa bits), 3 partial products; the
 * gradient operand A is first multiplied by the power of two that brings a_absmax to 2^14 (undone in the epilogue). */
typedef struct {
  const float* A;
  long a_rs, a_ks;
  const float* B;
  long b_rs, b_ks;
  float* C;
  long ldc;
  int M, N, K;
  const float* bias;
  const float* addend;
  long ld_add;
  int add_div;
  int act;
  int accumulate;
  int k_split;
  const float* a_absmax; /* DEVICE scalar: largest |A| when A is a gradient tensor (scaled into the half range), NULL for operands of order one */
  const float* act_y;    /* data gradient through the PRODUCING layer's activation: C = (A . B) * act'(act_y[m ld_y + n]) from that layer's saved
                          * OUTPUT (ELU' = y > 0 ? 1 : y + 1; ReLU' = y > 0): the autograd of nn.ELU / nn.ReLU fused into the Linear's backward
                          * (mlp_network.py:342-397, 587-603).  NULL to skip; needs accumulate = 0 */
  long ld_y;
  int act_y_kind;        /* 1 ELU, 2 ReLU */
  float* colsum_part;    /* [ceil(M / 128), ld_part]: a workgroup leaves the column sums of its 128-row result tile in row (tile index); every
                          * entry [tile, 0:N) is written, no initialisation needed -- the partial sums of the NEXT
                          * bias gradient: autograd of nn.Linear's bias, mlp_network.py:342-397) -- NULL to skip; needs accumulate = 0 and
                          * 16-byte-aligned result rows (N, ldc multiples of 4) */
  long ld_part;
  float* amax_part;      /* [ceil(M / 128) * ceil(N / 128)], all written: largest |result| per workgroup (required with colsum_part) */
  const float* rowscale; /* [M] or NULL: C = act(diag(rowscale) (A . B) + addend + bias) -- the forward of a Linear applied to x * s[row] (x * weight,
                          * x * vis: mlp_network.py:470, 474) without materialising the scaled input; not with a split reduction */
  const float* kscale;   /* [K] or NULL: B's element (n, k) is multiplied by kscale[k] -- the weight gradient dW = dZ^T (diag(s) X) of such a Linear
                          * from the unscaled X (the weight-gradient shape on the ring form only: both operands k-major with 16-byte-aligned rows, accumulate != 0) */
} DynTrainGemmParams;
int dyn_train_gemm(const DynTrainGemmParams* p, void* stream);
/* Developer / test knob: which kernel form dyn_train_gemm uses -- 0 automatic (default; the environment variable DYNIBAR_TRAIN_GEMM =
 * auto | tile | ring sets the initial value: the ring form for the backward products, the tile kernel for the forward ones), 1 the tile
 * kernel only, 2 the ring form wherever the operands allow it.  Results do not depend on it beyond the summation order. */
int dyn_train_gemm_mode(int mode);
/* second stage of colsum_part / amax_part: dbias[n] += sum over the tiles (dbias may be NULL), *absmax = max(*absmax, all of amax_part) */
int dyn_train_colsum_reduce(const float* colsum_part, long tiles, int N, long ld_part, float* dbias, const float* amax_part, long n_amax,
                            float* absmax, void* stream);

/* dZ = dY * act'(Y) in place (act 1: ELU, act 2: ReLU, both from the saved output Y; act 0: unchanged); dbias[c] += column sums (NULL to skip);
 * dseg[(row / seg), c] = sums over the seg rows of a point (gradient of a per-point addend; NULL to skip); absmax: see dyn_train_absmax. */
int dyn_train_act_bwd(float* dY, const float* Y, long rows, int cols, long ld_dy, long ld_y, int act, float* dbias, int seg, float* dseg,
                      long ld_seg, float* absmax, void* stream);
/* absmax[0] = max(absmax[0], largest |x|) over a [rows, cols] matrix: the scale of a gradient tensor for dyn_train_gemm's a_absmax (the
 * same by-product of dyn_train_act_bwd when `absmax` != NULL there); absmax is zeroed by the caller. */
int dyn_train_absmax(const float* x, long rows, int cols, long ld, float* absmax, void* stream);

/* mlp_network.py:423-448 + render_ray.py:372-396: a0 [N,104] = [PE(pts) 33 | PE(src Pluecker) 66 | ray_diff 4 | 0], ref_pe [R,68] =
 * [PE(ref Pluecker) 66 | 0 0], mask_eff [N] = mask (* (sum rgb > 1e-3) when mask_rgb).  centers: source camera centres, centers[v *
 * center_stride + 0..2] (proj + 12 with stride 16 from dyn_prepare_cameras). */
int dyn_train_static_embed(const float* pts, const float* ray_o, const float* ray_d, const float* centers, int center_stride,
                           const float* ray_diff, const float* rgb_feat, const float* mask, int R, int S, int V, int mask_rgb, float* a0,
                           float* ref_pe, float* mask_eff, void* stream);

/* mlp_network.py:450: f [N,72] = [rgb_feat 35 | src_feat * ref_feat[ray] 35 | 0 0]; backward: dsrc [N,.], dref [R,.] */
int dyn_train_build_f(const float* rgb_feat, const float* src_feat, long ld_src, const float* ref_feat, long ld_ref, long N, int rows_per_ray,
                      float* f, void* stream);
int dyn_train_build_f_bwd(const float* df, long ld_df, const float* src_feat, long ld_src, const float* ref_feat, long ld_ref, long R,
                          int rows_per_ray, float* dsrc, long ld_dsrc, float* dref, long ld_dref, void* stream);

/* pooling weights over the views of a point.  mode 0 (mlp_network.py:452-459): in = dot products (ray_diff + 3, stride 4), s_param =
 * the pooling temperature `s` (NULL: anti_alias_pooling = 0); mode 1 (:470-471,:476,:481): in = vis_fc2 logits; also writes vis, the mean
 * weight per point and the number of valid views.  backward: dw -> ds (mode 0, += one scalar) or dlogit (mode 1; dvis_direct and dwmean
 * are the other uses of vis and of the mean weight). */
int dyn_train_view_weights(int mode, const float* in, long in_stride, const float* mask, const float* s_param, long P, int V, float* w,
                           float* vis_out, long vis_stride, float* wmean, long wmean_stride, float* nvalid, void* stream);
int dyn_train_view_weights_bwd(int mode, const float* in, long in_stride, const float* mask, const float* s_param, long P, int V,
                               const float* w, const float* dw, const float* dvis_direct, long dvis_stride, const float* vis, long vis_stride,
                               const float* dwmean, long dwmean_stride, float* dlogit, long dlogit_stride, float* ds, void* stream);

/* fused_mean_variance (mlp_network.py:115-119) over the V rows of each point, C columns; backward into dx (accumulate 0/1) and dw. */
int dyn_train_meanvar(const float* x, long ldx, const float* w, long P, int V, int C, float* mean, float* var, long ld_out, void* stream);
int dyn_train_meanvar_bwd(const float* x, long ldx, const float* w, long P, int V, int C, const float* mean, const float* dmean,
                          const float* dvar, long ld_stat, float* dx, long ld_dx, int accumulate, float* dw, int dw_accumulate, void* stream);

/* y = x * s[row] (mlp_network.py:465,:469) and its backward */
int dyn_train_rowscale(const float* x, long ldx, const float* s, long s_stride, long N, int C, float* y, long ldy, void* stream);
int dyn_train_rowscale_bwd(const float* dy, long ld_dy, const float* x, long ldx, const float* s, long s_stride, long N, int C, float* dx,
                           long ld_dx, int accumulate, float* ds, long ds_stride, int ds_accumulate, void* stream);

/* mlp_network.py:466-468: x2 = x1 + xv[:, :128], vis0 = sigmoid(xv[:, 128]) mask; with ray_diff [N,4] != NULL the row of x2 is laid out as
 * rgb_fc.0's per-view input [x2 128 | vis (written later) | ray_diff 4 | 0 0 0] (ld2 >= 136, :495-503); backward fills dxv [N, 129] */
int dyn_train_vis_split(const float* x1, long ld1, const float* xv, long ldv, const float* mask, const float* ray_diff, long N, float* x2,
                        long ld2, float* vis0, void* stream);
int dyn_train_vis_split_bwd(const float* dx2, long ld_dx2, const float* dvis0, const float* xv, long ldv, const float* mask, long N, float* dxv,
                            long ld_dxv, void* stream);
/* The two calls above followed by dyn_train_act_bwd in ONE pass over the rows (128 columns, 16-byte-aligned rows):
 *  dyn_train_rowscale_act_bwd: dx = (dx + dy * s[row]) * act'(x), ds[row] (=|+=) <dy[row], x[row]> -- x is both what s multiplied in the
 *    forward pass and the saved output of the activation in front of it (mlp_network.py:466-470: x = base_fc(...); vis_fc(x * weight));
 *  dyn_train_vis_split_act_bwd: dxv[:, 0:128] = dx2 * ELU'(xv[:, 0:128]), dxv[:, 128] = dvis0 * mask * sigmoid'(xv[:, 128]) * ELU'(xv[:, 128])
 *    (mlp_network.py:470-473: vis_fc ends in an ELU over its 129 outputs); with dxs != NULL (then dvis0 may be NULL) the backward of
 *    vis_fc2's input x * vis (:474) comes first in the same pass: dx2 += dxs * vis0[row] (written back), dvis0[row] = <dxs[row], x2[row]>.
 * dbias (may be NULL) += the column sums of the result (128 / 129 entries), absmax (may be NULL) = max(absmax, largest |result|). */
int dyn_train_rowscale_act_bwd(const float* dy, long ld_dy, const float* x, long ldx, const float* s, long s_stride, long N, float* dx,
                               long ld_dx, float* ds, long ds_stride, int ds_accumulate, int act, float* dbias, float* absmax, void* stream);
int dyn_train_vis_split_act_bwd(float* dx2, long ld_dx2, const float* dvis0, const float* xv, long ldv, const float* mask, long N, float* dxv,
                                long ld_dxv, float* dbias, float* absmax, const float* dxs, long ld_dxs, const float* x2, long ldx2,
                                const float* vis0, void* stream);
/* Forward of a Linear with ONE output (mlp_network.py:474-476, 487-493, 505-507): y[row * y_stride] = <X[row, 0:C], w> + bias[0] (bias may be
 * NULL) for C = 4 * 2^k <= 256 columns in 16-byte-aligned rows; fp32 products and sums. */
int dyn_train_rowdot(const float* X, long ldx, const float* w, const float* bias, long N, int C, float* y, long y_stride, void* stream);
/* Data gradient of a Linear with ONE output through the activation in front of it (autograd of vis_fc2.2 / rgb_fc.4 / out_geometry_fc.2 and
 * of the ELU before them, mlp_network.py:474-476, 487-493, 505-507): dX[row, c] = dz[row] * w[c] * act'(Y[row, c]) for C = 4 * 2^k <= 256 columns
 * in 16-byte-aligned rows; dbias (may be NULL) += column sums, absmax (may be NULL) = max(absmax, largest |dX|); dW (may be NULL; needs
 * act != 0) [C] += sum over the rows of dz[row] * Y[row, :], the layer's own weight gradient (its input is Y), in the same pass. */
int dyn_train_outer_act_bwd(const float* dz, long dz_stride, const float* w, const float* Y, long ld_y, long N, int C, int act, float* dX,
                            long ld_dx, float* dbias, float* absmax, float* dW, void* stream);

/* ScaledDotProductAttention (mlp_network.py:13-31) for 4 heads of 32: qkv [P,384] = q | k | v, nvalid [P] = views that see the point
 * (query rows with nvalid <= 1 are masked, :24 and :486-488), out [P,128], prob [R,4,S,S] saved for the backward;
 * backward: dscore [R,4,S,S] scratch, dqkv [P,384]. */
int dyn_train_attn(const float* qkv, const float* nvalid, int R, int S, float* out, float* prob, void* stream);
int dyn_train_attn_bwd(const float* qkv, const float* nvalid, int R, int S, const float* prob, const float* dout, float* dscore, float* dqkv,
                       void* stream);

/* out = LayerNorm_128(a + b; eps 1e-6) gamma + beta (mlp_network.py:99-102); saves xhat [P,128], rstd [P]; backward: din (the gradient
 * of both a and b), dgamma / dbeta += */
int dyn_train_layernorm(const float* a, const float* b, const float* gamma, const float* beta, long P, float* out, float* xhat, float* rstd,
                        void* stream);
int dyn_train_layernorm_bwd(const float* dout, const float* xhat, const float* rstd, const float* gamma, long P, float* din, float* dgamma,
                            float* dbeta, void* stream);

/* mlp_network.py:503-527: masked softmax of the rgb_fc logits over views, rgb = sum_v rgb_in blend, sigma filled with -1e9 where no
 * view sees the point -> raw [P,4]; blend [N] saved.  backward: draw [P,4] -> dlogit [N], dsigma [P]. */
int dyn_train_blend(const float* logit, long logit_stride, const float* mask, const float* rgb_feat, const float* sigma, long sigma_stride,
                    const float* nvalid, long P, int V, float* blend, float* raw, void* stream);
int dyn_train_blend_bwd(const float* draw, const float* blend, const float* mask, const float* rgb_feat, const float* nvalid, long P, int V,
                        float* dlogit, long dlogit_stride, float* dsigma, long dsigma_stride, void* stream);

/* backward of raw2outputs_vanilla (render_ray.py:134-201): alpha, weights as saved by dyn_composite; drgb [R,3], ddepth [R],
 * dweights [R,S] (each may be NULL) -> draw [R,S,4].  S <= 4096.  The sums over the samples BEHIND a sample are formed directly, walking the ray from
 * its far end (not as total - prefix: far down a ray that difference is pure rounding, coherent along the ray). */
int dyn_train_composite_bwd(const float* raw, const float* z_vals, const float* alpha, const float* weights, const float* drgb,
                            const float* ddepth, const float* dweights, int R, int S, float* draw, void* stream);

/* backward of the bilinear feature gather (projection.py:160-167, F.grid_sample w.r.t. the maps): drgb_feat[row, col0 .. col0 + F)
 * scattered with the forward's taps into dfeat_cl [V,Hf,Wf,F] (channels-last, zeroed by the caller; atomic +=).  pts_st [R,S,3], or
 * xyz [V,R,S,3] != NULL: the per-view motion-displaced points of the dynamic branch. */
int dyn_gather_bwd(const float* pts_st, const float* xyz, const float* proj, int R, int S, int V, int Hf, int Wf, int F, float img_h, float img_w,
                   const float* drgb_feat, long ld_d, int col0, float* dfeat_cl, void* stream);

/* ==== f3, second slice: DynibarDynamic.forward (mlp_network.py:236-316) and the two-branch raw2outputs (render_ray.py:214-330) for
 * training: the dynamic net reuses the primitives above (dynibar_amd/train_dynamic.py holds its sequence) plus these. ================== */

/* y[row, c] = x[row, c] + tab[(row % period), c]: the time feature of mlp_network.py:244-249 on every row (period 1), the positional table
 * of :280 on every ray (period S).  The backward of both is the identity on x. */
int dyn_train_add_table(const float* x, long ldx, const float* tab, long ld_tab, int period, long rows, int C, float* y, long ldy, void* stream);

/* Fourier features of the dynamic net (mlp_network.py:147-160,:290,:303): pts_pe [P,36] = [PE_5(pts) 33 | 0 0 0], dir_pe [R,28] = [PE_4(dir) 27 | 0] */
int dyn_train_dynamic_embed(const float* pts, const float* ray_d, long P, int R, float* pts_pe, float* dir_pe, void* stream);

/* colour / density head (mlp_network.py:295-315): raw [P,4] = [sigmoid(logit) or 0 where no view sees the point | sigma - shift or -1e9] */
int dyn_train_dynamic_head(const float* logit, long ld_logit, const float* sigma, const float* nvalid, float shift, long P, float* raw, void* stream);
int dyn_train_dynamic_head_bwd(const float* draw, const float* raw, const float* nvalid, long P, float* dlogit, long ld_dlogit, float* dsigma,
                               void* stream);

/* backward of raw2outputs (render_ray.py:246-330): upstream gradients of rgb, rgb_static, rgb_dy [R,3], depth [R], weights_dy, weights_st,
 * weights [R,S] (each may be NULL) -> draw_dy, draw_st [R,S,4].  S <= 4096; suffix sums as in dyn_train_composite_bwd. */
int dyn_train_composite2_bwd(const float* raw_dy, const float* raw_st, const float* z_vals, const float* g_rgb, const float* g_rgb_st,
                             const float* g_rgb_dy, const float* g_depth, const float* g_wd, const float* g_ws, const float* g_w, int R, int S,
                             float* draw_dy, float* draw_st, void* stream);

/* ==== f3, third slice: the motion path of train.py:283-467 -- gradients w.r.t. the sample locations and through them into MotionMLP
 * (mlp_network.py:558-618; its Linear layers are dyn_train_gemm with act 2 = ReLU, dyn_train_act_bwd act 2) and the trajectory basis. ====== */

/* PeriodicEmbed (mlp_network.py:530-555) of x [rows, D] (ldx): out [rows, ld] = [x | cos(f_k x) k < n | sin(f_k x)]; freqs: HOST array of
 * n_freqs <= 16 frequencies.  backward w.r.t. x (accumulate 0/1). */
int dyn_train_embed(const float* x, long ldx, long rows, int D, const float* freqs, int n_freqs, float* out, long ld, void* stream);
int dyn_train_embed_bwd(const float* x, long ldx, long rows, int D, const float* freqs, int n_freqs, const float* dout, long ld, float* dx, long ld_dx,
                        int accumulate, void* stream);

/* x [R,S,C]: the last n_last samples of every ray zeroed, the rest multiplied by scale (render_ray.py:961,:1129 `raw_coeff[:, -n:, :] *= 0`,
 * mlp_network.py:618 `/ sf_mag_div`); applied to the gradient it is the backward of the same. */
int dyn_train_zero_tail(float* x, long R, int S, int C, int n_last, float scale, void* stream);

/* backward of dyn_trajectory_points (render_ray.py:361-369,:965-985): dseq [n_rows, n_pts, 3] -> dcoeff [n_pts, 3B] (=), dbasis [frames, B]
 * (atomic +=, zeroed by the caller), dpts [n_pts, 3] (= ; may be NULL) */
int dyn_trajectory_bwd(const float* dseq, const float* coeff, const float* basis, long n_pts, int B, const int* rows, int n_rows, int row_ref,
                       float* dcoeff, float* dbasis, float* dpts, void* stream);

/* backward of dyn_render_flows (render_ray.py:333-358): dflows [V,R,2] -> dweights [R,S] (=), dseq [V,R,S,3] (=) */
int dyn_render_flows_bwd(const float* dflows, const float* weights, const float* pts_seq, const float* proj, int R, int S, int V, float* dweights,
                         float* dseq, void* stream);

/* backward of the gather w.r.t. the sample locations (F.grid_sample w.r.t. its grid, normalize(), compute_projections: projection.py:32-59,
 * :134-167): drgb_feat [N, ld_d] (columns 0..2 colours, 3..3+F features) -> dxyz [V,R,S,3] (=).  pts_st / xyz as in dyn_gather_bwd. */
int dyn_gather_bwd_pts(const float* pts_st, const float* xyz, const float* proj, const float* src_rgb, const float* feat_cl, int R, int S, int V,
                       int H, int W, int Hf, int Wf, int F, float img_h, float img_w, const float* drgb_feat, long ld_d, float* dxyz, void* stream);

/* ====== multi-GPU pixel gather (the counterpart of nn.DataParallel's gather of the rendered outputs, reference
 * ibrnet/model.py:134-159, around the chunk loop of render_image.py:60-217) ============================================================
 * Rays shard across ranks as contiguous tiles with no data-path collective; the one exchange of a frame is an all-gather of every rank's
 * packed [rows_per_rank, cols] fp32 pixel rows (rgb, depth, mask ...) over RCCL / xGMI.  `comm` is an ncclComm_t: the caller's own, or one made
 * by dyn_comm_init_rank.  RCCL is taken from the process image at run time (a PyTorch host has loaded its own librccl; a communicator is
 * only valid in the library instance that made it), else from librccl.so / $DYNIBAR_RCCL_LIB; without it these calls return DYN_E_INVALID. */
int dyn_comm_available(void);
/* id128: HOST buffer of 128 bytes (ncclUniqueId).  Rank 0 makes it, the host's own channel hands it to every rank. */
int dyn_comm_unique_id(void* id128);
int dyn_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int dyn_comm_size_rank(void* comm, int* nranks, int* rank);
int dyn_comm_destroy(void* comm);
/* recv [nranks, rows_per_rank, cols] <- send [rows_per_rank, cols] of every rank, in rank order, on `stream` (one equal-count ncclAllGather;
 * tiles are padded to the common size by the caller: balanced tiles differ by at most one ray). */
int dyn_gather_tiles(const float* send, float* recv, long rows_per_rank, int cols, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNIBAR_HIP_H_ */
