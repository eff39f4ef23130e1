"""CPU oracle for the DynIBaR per-ray hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch functional restatement (torch CPU tensors, explicit
weight dictionaries, no nn.Module) of the reference's per-ray renderer.  It
exists so that the HIP kernels can be checked on a box where /root/reference
is absent.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``dynibar_amd`` never
does and fails loudly when its HIP library is missing.

Pinning: ``tests/golden/make_golden.py`` runs the *real* reference modules
(imported read-only from /root/reference in the build container) on seeded
inputs and commits their outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those
fixtures (bit-exact on the same torch build).

Each function cites the reference lines it follows (paths under
/root/reference/ibrnet/).  The arithmetic uses the same torch operators in the
same order as the reference so that on one torch build the results are
bit-identical; the structure (free functions over dicts of arrays) is ours.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------


def as_t(x):
  if isinstance(x, torch.Tensor):
    return x
  return torch.from_numpy(np.ascontiguousarray(x))


def tdict(sd):
  """numpy state-dict -> torch state-dict."""
  return {k: as_t(v) for k, v in sd.items()}


def _lin(sd, name, x):
  return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


# ----------------------------------------------------------------------------
# a1/a2  cameras and rays            (sample_ray.py:11-16, :143-163)
# ----------------------------------------------------------------------------


def parse_camera(params):
  H = params[:, 0]
  W = params[:, 1]
  K = params[:, 2:18].reshape(-1, 4, 4)
  c2w = params[:, 18:34].reshape(-1, 4, 4)
  return W, H, K, c2w


def image_rays(camera, render_stride=1):
  """All-pixel rays of a target view, row-major over (v,u)   (sample_ray.py:143-163).
  Returns rays_o [HW,3], rays_d [HW,3], uv_grid [HW,2]."""
  W, H, K, c2w = parse_camera(camera)
  H, W = int(H[0]), int(W[0])
  u, v = np.meshgrid(np.arange(W)[::render_stride], np.arange(H)[::render_stride])
  u = u.reshape(-1).astype(np.float32)
  v = v.reshape(-1).astype(np.float32)
  pix = torch.from_numpy(np.stack((u, v, np.ones_like(u)), axis=0))[None]
  d = c2w[:, :3, :3].bmm(torch.inverse(K[:, :3, :3])).bmm(pix).transpose(1, 2).reshape(-1, 3)
  o = c2w[:, :3, 3].unsqueeze(1).repeat(1, d.shape[0], 1).reshape(-1, 3)
  uv = torch.from_numpy(np.stack((u, v), -1))
  return o, d, uv


# ----------------------------------------------------------------------------
# a5  depth sampling                 (render_ray.py:67-131)
# ----------------------------------------------------------------------------


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False, t_rand=None):
  near_v = depth_range[0, 0]
  far_v = depth_range[0, 1]
  assert near_v > 0 and far_v > 0 and far_v > near_v
  near = near_v * torch.ones_like(ray_d[..., 0])
  far = far_v * torch.ones_like(ray_d[..., 0])
  if inv_uniform:
    start = 1.0 / near
    step = (1.0 / far - start) / (N_samples - 1)
    z_vals = 1.0 / torch.stack([start + i * step for i in range(N_samples)], dim=1)
  else:
    start = near
    step = (far - near) / (N_samples - 1)
    z_vals = torch.stack([start + i * step for i in range(N_samples)], dim=1)
  if not det:
    mids = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
    upper = torch.cat([mids, z_vals[:, -1:]], dim=-1)
    lower = torch.cat([z_vals[:, 0:1], mids], dim=-1)
    if t_rand is None:
      t_rand = torch.rand_like(z_vals)
    z_vals = lower + (upper - lower) * t_rand
  pts = z_vals.unsqueeze(2) * ray_d.unsqueeze(1) + ray_o.unsqueeze(1)
  s_vals = z_to_s(z_vals, near_v, far_v)
  return pts, z_vals, s_vals


def z_to_s(z_vals, near_v, far_v):
  """mip-NeRF-360 normalised disparity  (render_ray.py:399-404, :126-129)."""
  return ((1.0 / z_vals) - (1.0 / near_v)) / (1.0 / far_v - 1.0 / near_v)


# ----------------------------------------------------------------------------
# a6/a7  inverse-CDF resampling      (render_ray.py:19-64, :790-831)
# ----------------------------------------------------------------------------


def pdf_to_cdf(weights):
  """weights [R,M] (already the interior slice) -> cdf [R,M+1] (render_ray.py:22-27).
  NOTE the reference adds 1e-5 *in place*; here it is functional."""
  w = weights + 1e-5
  pdf = w / torch.sum(w, dim=-1, keepdim=True)
  cdf = torch.cumsum(pdf, dim=-1)
  return torch.cat([torch.zeros_like(cdf[:, 0:1]), cdf], dim=-1)


def invert_cdf(bins, cdf, u):
  """(render_ray.py:36-64) returns samples [R,N] and the bit-exact index tensor above_inds."""
  M = cdf.shape[1] - 1
  above = torch.zeros_like(u, dtype=torch.long)
  for i in range(M):
    above += (u >= cdf[:, i:i + 1]).long()
  below = torch.clamp(above - 1, min=0)
  cdf_lo = torch.gather(cdf, 1, below)
  cdf_hi = torch.gather(cdf, 1, above)
  b_lo = torch.gather(bins, 1, below)
  b_hi = torch.gather(bins, 1, above)
  denom = cdf_hi - cdf_lo
  denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
  t = (u - cdf_lo) / denom
  return b_lo + t * (b_hi - b_lo), above


def cdf_sample_conditioning(z_vals, weights, N_importance, inv_uniform, det, u=None, band=4e-7, cdf_err=4e-7):
  """Test helper (not in the reference): per drawn sample, (tie, tol) in the domain sample_pdf works in (1/z if inv_uniform).

  The reference's t = (u - cdf_lo) / (cdf_hi - cdf_lo) divides by the bin's probability mass, so an error `cdf_err` in the cdf
  (a few ulp of 1.0 from torch.sum / torch.cumsum, whose last bits depend on the CPU vector width) moves the sample by
  bin_width * cdf_err / mass: in empty bins (mass ~1e-5) that is a few percent of the bin.  `tol` is that bound.  `tie` marks
  samples whose bin mass is within `band` of the hard 1e-5 threshold (render_ray.py:57-58), where the last bit decides
  between t = (u-c)/mass and t = (u-c)/1, i.e. between the two edges of an empty bin; tied samples are excluded."""
  w = weights[:, 1:-1]
  if inv_uniform:
    inv_z = 1.0 / z_vals
    bins = torch.flip(0.5 * (inv_z[:, 1:] + inv_z[:, :-1]), dims=[1])
    w = torch.flip(w, dims=[1])
  else:
    bins = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
  cdf = pdf_to_cdf(w)
  if det:
    u = torch.linspace(0.0, 1.0, N_importance).unsqueeze(0).repeat(cdf.shape[0], 1)
  M = cdf.shape[1] - 1
  above = torch.zeros_like(u, dtype=torch.long)
  for i in range(M):
    above += (u >= cdf[:, i:i + 1]).long()
  below = torch.clamp(above - 1, min=0)
  mass = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)
  width = (torch.gather(bins, 1, above) - torch.gather(bins, 1, below)).abs()
  tie = (mass - 1e-5).abs() < band
  eff = torch.where(mass < 1e-5, torch.ones_like(mass), mass)
  return tie, width * cdf_err / eff


def sample_pdf(bins, weights, N_samples, det=False, u=None, return_inds=False):
  cdf = pdf_to_cdf(weights)
  if det:
    u = torch.linspace(0.0, 1.0, N_samples).unsqueeze(0).repeat(bins.shape[0], 1)
  elif u is None:
    u = torch.rand(bins.shape[0], N_samples)
  samples, inds = invert_cdf(bins, cdf, u)
  return (samples, inds) if return_inds else samples


def fine_z_vals(z_vals, weights, N_importance, inv_uniform, det, u=None, return_inds=False):
  """Coarse weights -> sorted coarse+fine depths  (render_ray.py:790-821)."""
  weights = weights.clone()
  if inv_uniform:
    inv_z = 1.0 / z_vals
    inv_mid = 0.5 * (inv_z[:, 1:] + inv_z[:, :-1])
    w = weights[:, 1:-1]
    inv_s, inds = sample_pdf(torch.flip(inv_mid, dims=[1]), torch.flip(w, dims=[1]), N_importance,
                             det=det, u=u, return_inds=True)
    z_samples = 1.0 / inv_s
  else:
    mid = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
    w = weights[:, 1:-1]
    z_samples, inds = sample_pdf(mid, w, N_importance, det=det, u=u, return_inds=True)
  z_all, _ = torch.sort(torch.cat((z_vals, z_samples), dim=-1), dim=-1)
  return (z_all, inds) if return_inds else z_all


# ----------------------------------------------------------------------------
# a8-a11  projector                  (projection.py:13-176)
# ----------------------------------------------------------------------------


# 'reference': K.inv(c2w) exactly as the reference forms it (fp32 LU inverse + fp32 bmm).  'double': the same product formed in
# float64 and rounded once to fp32 -- what k_prepare_cameras does.  The two differ in the last bits of the matrix; running the oracle
# in both modes measures how far the REFERENCE's own outputs move under that perturbation (the conditioning bound the parity tests
# add to their tolerance on white-noise maps, tests/parity.py:projection_sensitivity).  Never changed outside a `with` block there.
PROJECTION_MODE = 'reference'


def compute_projections(xyz, train_cameras):
  """xyz [V,...,3], train_cameras [V,34] -> pixel_locations [V,...,2], in-front mask [V,...]
  (projection.py:32-59)."""
  shp = xyz.shape[:-1]
  xyz = xyz.reshape(shp[0], -1, 3)
  V = len(train_cameras)
  K = train_cameras[:, 2:18].reshape(-1, 4, 4)
  c2w = train_cameras[:, -16:].reshape(-1, 4, 4)
  xyz_h = torch.cat([xyz, torch.ones_like(xyz[..., :1])], dim=-1)
  if PROJECTION_MODE == 'double':
    P = K.double().bmm(torch.inverse(c2w.double())).to(torch.get_default_dtype())
    proj = P.bmm(xyz_h.permute(0, 2, 1)).permute(0, 2, 1)
  else:
    proj = K.bmm(torch.inverse(c2w)).bmm(xyz_h.permute(0, 2, 1)).permute(0, 2, 1)
  pix = proj[..., :2] / torch.clamp(proj[..., 2:3], min=1e-8)
  pix = torch.clamp(pix, min=-1e6, max=1e6)
  mask = proj[..., 2] > 0
  return pix.reshape((V,) + shp[1:] + (2,)), mask.reshape((V,) + shp[1:])


def compute_angle(xyz_st, xyz, query_camera, train_cameras):
  """(projection.py:61-101)  -> [V,...,4] = [unit(a-b), a.b]."""
  shp = xyz.shape[:-1]
  xyz_st_ = xyz_st.reshape(xyz_st.shape[0], -1, 3)
  xyz_ = xyz.reshape(xyz.shape[0], -1, 3)
  src_c2w = train_cameras[:, -16:].reshape(-1, 4, 4)
  V = len(src_c2w)
  q_c2w = query_camera[-16:].reshape(-1, 4, 4).repeat(V, 1, 1)
  a = F.normalize(q_c2w[:, :3, 3].unsqueeze(1) - xyz_st_, dim=-1)
  b = F.normalize(src_c2w[:, :3, 3].unsqueeze(1) - xyz_, dim=-1)
  diff = a - b
  dot = torch.sum(a * b, dim=-1, keepdim=True)
  out = torch.cat([F.normalize(diff, dim=-1), dot], dim=-1)
  return out.reshape((V,) + shp[1:] + (4,))


def compute_with_motions(xyz_st, xyz, query_camera, train_imgs, train_cameras, featmaps):
  """(projection.py:103-176)  xyz_st [R,S,3], xyz [V,R,S,3], query_camera [1,34],
  train_imgs [1,V,H,W,3], train_cameras [1,V,34], featmaps [V,F,Hf,Wf]
  -> rgb_feat [R,S,V,3+F], ray_diff [R,S,V,4], mask [R,S,V,1]."""
  assert train_imgs.shape[0] == 1 and train_cameras.shape[0] == 1 and query_camera.shape[0] == 1
  xyz_st = xyz_st[None].expand(xyz.shape[0], -1, -1, -1)
  imgs = train_imgs.squeeze(0).permute(0, 3, 1, 2)
  cams = train_cameras.squeeze(0)
  qcam = query_camera.squeeze(0)
  h, w = cams[0][:2]
  pix, in_front = compute_projections(xyz, cams)
  resize = torch.tensor([w - 1.0, h - 1.0]).to(pix.device)[None, None, :]  # (.to(device): projection.py:25)
  norm = 2 * pix / resize - 1.0
  rgb = F.grid_sample(imgs, norm, align_corners=True).permute(2, 3, 0, 1)
  feat = F.grid_sample(featmaps, norm, align_corners=True).permute(2, 3, 0, 1)
  rgb_feat = torch.cat([rgb, feat], dim=-1)
  inb = (pix[..., 0] <= w - 1.0) & (pix[..., 0] >= 0) & (pix[..., 1] <= h - 1.0) & (pix[..., 1] >= 0)
  ray_diff = compute_angle(xyz_st, xyz, qcam, cams).permute(1, 2, 0, 3)
  mask = (inb * in_front).to(torch.get_default_dtype()).permute(1, 2, 0)[..., None]
  return rgb_feat, ray_diff, mask


# ----------------------------------------------------------------------------
# section 8(f)1  feature encoder     (feature_network.py:13-85, 179-311)
# ----------------------------------------------------------------------------


def _conv_reflect(x, w, stride, pad, bias=None):
  """nn.Conv2d(..., padding=pad, padding_mode='reflect', bias=...) (feature_network.py:13-38, 240-249)."""
  if pad > 0:
    x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
  return F.conv2d(x, w, bias, stride=stride)


def _inorm(sd, name, x):
  """nn.InstanceNorm2d(C, track_running_stats=False, affine=True) (feature_network.py:60-63)."""
  return F.instance_norm(x, weight=sd[name + '.weight'], bias=sd[name + '.bias'], eps=1e-5)


def resnet_encoder(sd, x, coarse_out_ch=32, fine_out_ch=32):
  """The executed part of ResNet.forward (feature_network.py:302-311): x [N,3,H,W] -> (x_coarse, x_fine) [N,32,H/4,W/4]."""
  x = F.relu(_inorm(sd, 'bn1', _conv_reflect(x, sd['conv1.weight'], 2, 3)))
  for b in range(3):
    pre = 'layer1.%d.' % b
    identity = x
    out = F.relu(_inorm(sd, pre + 'bn1', _conv_reflect(x, sd[pre + 'conv1.weight'], 2 if b == 0 else 1, 1)))
    out = _inorm(sd, pre + 'bn2', _conv_reflect(out, sd[pre + 'conv2.weight'], 1, 1))
    if b == 0:
      identity = _inorm(sd, pre + 'downsample.1', _conv_reflect(x, sd[pre + 'downsample.0.weight'], 2, 0))
    x = F.relu(out + identity)
  x_out = F.conv2d(x, sd['out_conv.weight'], sd['out_conv.bias'])
  return x_out[:, :coarse_out_ch], x_out[:, -fine_out_ch:]


# ----------------------------------------------------------------------------
# a12  Pluecker coordinates          (render_ray.py:372-396)
# ----------------------------------------------------------------------------


def cross_first_axis_of_3(a, b):
  """``torch.cross(a, b)`` WITHOUT ``dim`` as render_ray.py:375 and :392 call it: torch crosses over the FIRST axis of size 3 (the deprecated
  default, still what torch 2.x does) -- xyz only when no earlier axis has size 3.  With exactly 3 source views / a chunk of exactly 3 rays /
  3 samples per ray the moments are therefore products over that axis.  Pinned against the reference's own functions run on such shapes
  (tests/golden/cross_axis.npz)."""
  dim = next(i for i, n in enumerate(a.shape) if n == 3)
  return torch.linalg.cross(a, b, dim=dim)


def ref_plucker(ray_o, ray_d):
  d = F.normalize(ray_d, dim=-1)
  return torch.cat([d, cross_first_axis_of_3(ray_o, d)], dim=-1)


def src_plucker(pts, src_cameras):
  """pts [R,S,3] (or [V,R,S,3]), src_cameras [1,V,34] -> [R,S,V,6]."""
  c2w = src_cameras[0, :, -16:].reshape(-1, 4, 4)
  o = c2w[:, :3, 3].unsqueeze(1).unsqueeze(1)
  ray = (pts.unsqueeze(0) if pts.dim() == 3 else pts) - o
  ray = F.normalize(ray, dim=-1)
  mom = cross_first_axis_of_3(o.expand(-1, ray.shape[1], ray.shape[2], -1), ray)
  return torch.cat([ray, mom], dim=-1).permute(1, 2, 0, 3)


# ----------------------------------------------------------------------------
# a13  Fourier features              (mlp_network.py:530-555)
# ----------------------------------------------------------------------------


def periodic_embed(x, max_freq, n_freq, linspace):
  if linspace:
    freqs = torch.linspace(1, max_freq + 1, steps=n_freq)
  else:
    freqs = 2 ** torch.linspace(0, n_freq - 1, steps=n_freq)
  out = [x]
  for fn in (torch.cos, torch.sin):
    for f in freqs:
      out.append(fn(f * x))
  return torch.cat(out, -1)


# ----------------------------------------------------------------------------
# a14/a15  motion MLP + DCT trajectory   (mlp_network.py:558-618, render_ray.py:361-369, model.py:18-30)
# ----------------------------------------------------------------------------


def init_dct_basis(num_basis, num_frames):
  T, K = num_frames, num_basis
  b = torch.zeros([T, K])
  for t in range(T):
    for k in range(1, K + 1):
      b[t, k - 1] = np.sqrt(2.0 / T) * np.cos(np.pi / (2.0 * T) * (2 * t + 1) * k)
  return b


def motion_mlp(sd, x, D=8, skips=(4,), num_freqs=16, sf_mag_div=1.0):
  pe = periodic_embed(x, num_freqs, num_freqs, True)
  h = pe
  for i in range(D):
    h = F.relu(_lin(sd, f'pts_linears.{i}', h))
    if i in skips:
      h = torch.cat([pe, h], -1)
  return _lin(sd, 'coeff_linear', h) / sf_mag_div


def compute_traj_pts(cx, cy, cz, basis_row):
  return torch.cat([torch.sum(cx * basis_row, -1, keepdim=True), torch.sum(cy * basis_row, -1, keepdim=True),
                    torch.sum(cz * basis_row, -1, keepdim=True)], dim=-1)


def trajectory_points(raw_coeff, basis, frame_idx, offsets=(-3, -2, -1, 0, 1, 2, 3)):
  """raw_coeff [R,S,3B] (last 10% samples already zeroed) -> {offset: [R,S,3]}  (render_ray.py:686-701)."""
  B = basis.shape[1]
  cx, cy, cz = raw_coeff[..., 0:B], raw_coeff[..., B:2 * B], raw_coeff[..., 2 * B:3 * B]
  return {o: compute_traj_pts(cx, cy, cz, basis[None, None, frame_idx + o, :]) for o in offsets}


# ----------------------------------------------------------------------------
# a18/a19  attention + fused mean/var (mlp_network.py:13-31, :56-104, :115-119)
# ----------------------------------------------------------------------------


def fused_mean_variance(x, weight):
  mean = torch.sum(x * weight, dim=2, keepdim=True)
  var = torch.sum(weight * (x - mean) ** 2, dim=2, keepdim=True)
  return mean, var


def ray_attention(sd, x, mask, n_head=4, d_k=32, d_v=32, prefix='ray_attention'):
  """x [R,S,128], mask [R,S,1] (query-row validity, broadcast over keys: mlp_network.py:24,92)."""
  R, S, _ = x.shape
  res = x
  q = _lin(sd, prefix + '.w_qs', x).view(R, S, n_head, d_k).transpose(1, 2)
  k = _lin(sd, prefix + '.w_ks', x).view(R, S, n_head, d_k).transpose(1, 2)
  v = _lin(sd, prefix + '.w_vs', x).view(R, S, n_head, d_v).transpose(1, 2)
  attn = torch.matmul(q / (d_k ** 0.5), k.transpose(2, 3))
  attn = attn.masked_fill(mask.unsqueeze(1) == 0, -1e9)
  attn = F.softmax(attn, dim=-1)
  o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(R, S, -1)
  o = _lin(sd, prefix + '.fc', o)
  o = o + res
  return F.layer_norm(o, (o.shape[-1],), sd[prefix + '.layer_norm.weight'], sd[prefix + '.layer_norm.bias'], 1e-6)


def posenc_table(d_hid, n_samples):
  """(mlp_network.py:218-234)"""
  tab = np.array([[pos / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)] for pos in range(n_samples)])
  tab[:, 0::2] = np.sin(tab[:, 0::2])
  tab[:, 1::2] = np.cos(tab[:, 1::2])
  return torch.from_numpy(tab).to(torch.get_default_dtype()).unsqueeze(0)


# ----------------------------------------------------------------------------
# a16  dynamic net                   (mlp_network.py:236-316)
# ----------------------------------------------------------------------------


def _mlp2(sd, name, x, last_act=True):
  x = F.elu(_lin(sd, name + '.0', x))
  x = _lin(sd, name + '.2', x)
  return F.elu(x) if last_act else x


def dynamic_net(sd, pts_xyz, rgb_feat, glb_ray_dir, ray_diff, time_diff, mask, time, shift=0.0):
  V = rgb_feat.shape[2]
  time_pe = periodic_embed(time, 10, 10, False)[..., None, :].repeat(1, 1, V, 1).to(torch.get_default_dtype())
  rgb_feat = rgb_feat + _mlp2(sd, 'ray_dir_fc', time_pe)
  weight = mask / (torch.sum(mask, dim=2, keepdim=True) + 1e-8)
  mean, var = fused_mean_variance(rgb_feat, weight)
  x = torch.cat([torch.cat([mean, var], dim=-1).expand(-1, -1, V, -1), rgb_feat], dim=-1)
  x = _mlp2(sd, 'base_fc', x)
  x_vis = _mlp2(sd, 'vis_fc', x * weight)
  x_res, vis = torch.split(x_vis, [x_vis.shape[-1] - 1, 1], dim=-1)
  vis = torch.sigmoid(vis) * mask
  x = x + x_res
  vis = torch.sigmoid(_lin(sd, 'vis_fc2.2', F.elu(_lin(sd, 'vis_fc2.0', x * vis)))) * mask
  weight = vis / (torch.sum(vis, dim=2, keepdim=True) + 1e-8)
  mean, var = fused_mean_variance(x, weight)
  g = torch.cat([mean.squeeze(2), var.squeeze(2), weight.mean(dim=2)], dim=-1)
  g = _mlp2(sd, 'geometry_fc', g)
  n_valid = torch.sum(mask, dim=2)
  g = g + posenc_table(128, g.shape[1]).to(g.device)
  g = ray_attention(sd, g, (n_valid > 1).to(torch.get_default_dtype()))
  g = _mlp2(sd, 'ref_pts_fc', torch.cat([g, periodic_embed(pts_xyz, 5, 5, False)], dim=-1))
  sigma = _lin(sd, 'out_geometry_fc.2', F.elu(_lin(sd, 'out_geometry_fc.0', g))) - shift
  sigma = sigma.masked_fill(n_valid < 1, -1e9)
  dir_pe = periodic_embed(glb_ray_dir, 4, 4, False).to(torch.get_default_dtype())
  h = torch.cat([g, dir_pe[:, None, :].repeat(1, g.shape[1], 1)], dim=-1)
  h = F.elu(_lin(sd, 'rgb_fc.0', h))
  h = F.elu(_lin(sd, 'rgb_fc.2', h))
  rgb = torch.sigmoid(_lin(sd, 'rgb_fc.4', h))
  rgb = rgb.masked_fill(torch.sum(mask.repeat(1, 1, 1, 3), 2) == 0, 0)
  return torch.cat([rgb, sigma], dim=-1)


# ----------------------------------------------------------------------------
# a17  static net                    (mlp_network.py:423-527)
# ----------------------------------------------------------------------------


def static_net(sd, pts, ref_rays_coords, src_rays_coords, rgb_feat, glb_ray_dir, ray_diff, mask,
               anti_alias_pooling=True, mask_rgb=False, exp_jitter=None):
  """exp_jitter (test hook, not in the reference): multiplies exp(|s|(dot-1)) by (1 + jitter) to probe how strongly the
  anti-alias pooling weights (e - min_v e), a difference of nearly equal numbers, amplify 1-ulp differences of exp()."""
  V = rgb_feat.shape[2]
  ref_pe = periodic_embed(ref_rays_coords, 5, 5, False)
  src_pe = periodic_embed(src_rays_coords, 5, 5, False)
  pts_pe = periodic_embed(pts, 5, 5, False)
  ref_features = ref_pe[:, None, None, :].expand(-1, src_pe.shape[1], src_pe.shape[2], -1)
  src_features = torch.cat([pts_pe.unsqueeze(2).expand(-1, -1, src_pe.shape[2], -1), src_pe], dim=-1)
  src_feat = _mlp2(sd, 'ray_dir_fc', torch.cat([src_features, ray_diff], dim=-1), last_act=False)
  ref_feat = _lin(sd, 'ref_feature_fc.0', ref_features)
  rgb_in = rgb_feat[..., :3]
  if mask_rgb:
    mask = mask * (torch.sum(rgb_in, dim=-1, keepdim=True) > 1e-3).to(torch.get_default_dtype())
  rgb_feat = torch.cat([rgb_feat, src_feat * ref_feat], dim=-1)
  if anti_alias_pooling:
    dot = ray_diff[..., 3:4]
    e = torch.exp(torch.abs(sd['s']) * (dot - 1))
    if exp_jitter is not None:
      e = e * (1.0 + exp_jitter)
    weight = (e - torch.min(e, dim=2, keepdim=True)[0]) * mask
    weight = weight / (torch.sum(weight, dim=2, keepdim=True) + 1e-8)
  else:
    weight = mask / (torch.sum(mask, dim=2, keepdim=True) + 1e-8)
  mean, var = fused_mean_variance(rgb_feat, weight)
  x = torch.cat([torch.cat([mean, var], dim=-1).expand(-1, -1, V, -1), rgb_feat], dim=-1)
  x = _mlp2(sd, 'base_fc', x)
  x_vis = _mlp2(sd, 'vis_fc', x * weight)
  x_res, vis = torch.split(x_vis, [x_vis.shape[-1] - 1, 1], dim=-1)
  vis = torch.sigmoid(vis) * mask
  x = x + x_res
  vis = torch.sigmoid(_lin(sd, 'vis_fc2.2', F.elu(_lin(sd, 'vis_fc2.0', x * vis)))) * mask
  weight = vis / (torch.sum(vis, dim=2, keepdim=True) + 1e-8)
  mean, var = fused_mean_variance(x, weight)
  g = torch.cat([mean.squeeze(2), var.squeeze(2), weight.mean(dim=2)], dim=-1)
  g = _mlp2(sd, 'geometry_fc', g)
  n_valid = torch.sum(mask, dim=2)
  g = ray_attention(sd, g, (n_valid > 1).to(torch.get_default_dtype()))
  sigma = _lin(sd, 'out_geometry_fc.2', F.elu(_lin(sd, 'out_geometry_fc.0', g)))
  sigma = sigma.masked_fill(n_valid < 1, -1e9)
  x = torch.cat([g[:, :, None, :].expand(-1, -1, V, -1), x, vis, ray_diff], dim=-1)
  x = F.elu(_lin(sd, 'rgb_fc.0', x))
  x = F.elu(_lin(sd, 'rgb_fc.2', x))
  x = _lin(sd, 'rgb_fc.4', x)
  x = x.masked_fill(mask == 0, -1e9)
  blend = F.softmax(x, dim=2)
  rgb = torch.sum(rgb_in * blend, dim=2)
  return torch.cat([rgb, sigma], dim=-1)


# ----------------------------------------------------------------------------
# a20/a21  compositing               (render_ray.py:134-330)
# ----------------------------------------------------------------------------


def _sigma2alpha(sigma):
  """softplus(beta=1, threshold=20) then 1-exp(-s*dist), dist = 1 except last = 1e10
  (render_ray.py:154-182; USE_DISTANCE=False, USE_SOFTPLUS=True at :14-16)."""
  dists = torch.ones_like(sigma)
  dists[..., -1] = 1e10
  return 1.0 - torch.exp(-F.softplus(sigma) * dists)


def _transmittance(alpha):
  T = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)[:, :-1]
  return torch.cat((torch.ones_like(T[:, 0:1]), T), dim=-1)


def raw2outputs_vanilla(raw, z_vals, mask):
  rgb = raw[:, :, :3]
  alpha = _sigma2alpha(raw[:, :, 3])
  weights = alpha * _transmittance(alpha)
  return OrderedDict([
      ('rgb', torch.sum(weights.unsqueeze(2) * rgb, dim=1)),
      ('depth', torch.sum(weights * z_vals, dim=-1)),
      ('weights', weights),
      ('mask', mask.to(torch.get_default_dtype()).sum(dim=1) > 8),
      ('alpha', alpha),
      ('z_vals', z_vals),
  ])


def raw2outputs(raw_dy, raw_static, z_vals, mask_dy, mask_static):
  rgb_dy, rgb_st = raw_dy[:, :, :3], raw_static[:, :, :3]
  alpha_dy = _sigma2alpha(raw_dy[:, :, 3])
  alpha_st = _sigma2alpha(raw_static[:, :, 3])
  alpha = 1 - (1 - alpha_st) * (1 - alpha_dy)
  T = _transmittance(alpha)
  w_dy = alpha_dy * T
  w_st = alpha_st * T
  rgb_map_dy = torch.sum(w_dy.unsqueeze(2) * rgb_dy, dim=1)
  rgb_map_st = torch.sum(w_st.unsqueeze(2) * rgb_st, dim=1)
  weights = alpha * T
  return OrderedDict([
      ('rgb', rgb_map_dy + rgb_map_st),
      ('rgb_static', rgb_map_st),
      ('rgb_dy', rgb_map_dy),
      ('depth', torch.sum(weights * z_vals, dim=-1)),
      ('alpha_dy', alpha_dy),
      ('weights_dy', w_dy),
      ('weights_st', w_st),
      ('alpha', alpha),
      ('weights', weights),
      ('mask', torch.bitwise_or(mask_dy.to(torch.get_default_dtype()).sum(dim=1) > 8, mask_static.to(torch.get_default_dtype()).sum(dim=1) > 8)),
      ('z_vals', z_vals),
  ])


# ----------------------------------------------------------------------------
# a22  expected optical flow         (render_ray.py:333-358)
# ----------------------------------------------------------------------------


def compute_optical_flow(weights, pts_seq, src_cameras, uv_grid):
  cams = src_cameras.squeeze(0)
  K = cams[:, 2:18].reshape(-1, 4, 4)
  w2c = torch.inverse(cams[:, -16:].reshape(-1, 4, 4))
  exp_pts = torch.sum(weights[None, ..., None] * pts_seq, dim=-2).unsqueeze(-1)
  cam_pts = torch.matmul(w2c[:, None, :3, :3], exp_pts) + w2c[:, None, :3, 3:4]
  pix = torch.matmul(K[:, None, :3, :3], cam_pts)
  pix = pix / pix[:, :, -1:, :]
  return pix[..., :2, 0] - uv_grid[None, ...]


# ----------------------------------------------------------------------------
# composed passes
# ----------------------------------------------------------------------------


def static_branch_pass(sd_static, scene, ray_o, ray_d, N_samples, inv_uniform=True, det=True,
                       anti_alias_pooling=True, mask_rgb=False, t_rand=None, return_stages=False):
  """BASELINE config 2: sample -> project/gather (static views) -> Pluecker -> DynibarStatic ->
  raw2outputs_vanilla.  This is the composition render_rays_mono uses for ``outputs_coarse_st``
  (render_ray.py:946-1071)."""
  pts, z_vals, s_vals = sample_along_camera_ray(ray_o, ray_d, scene['depth_range'], N_samples, inv_uniform, det, t_rand)
  Vs = scene['static_src_rgbs'].shape[1]
  rgb_feat, ray_diff, mask = compute_with_motions(
      pts, pts[None].repeat(Vs, 1, 1, 1), scene['camera'], scene['static_src_rgbs'],
      scene['static_src_cameras'], scene['static_featmaps'])
  pix_mask = mask[..., 0].sum(dim=2) > 1
  refc = ref_plucker(ray_o, ray_d)
  srcc = src_plucker(pts, scene['static_src_cameras'])
  raw = static_net(sd_static, pts, refc, srcc, rgb_feat, F.normalize(ray_d, dim=-1), ray_diff, mask,
                   anti_alias_pooling, mask_rgb)
  out = raw2outputs_vanilla(raw, z_vals, pix_mask)
  if return_stages:
    return out, dict(pts=pts, z_vals=z_vals, s_vals=s_vals, rgb_feat=rgb_feat, ray_diff=ray_diff, mask=mask,
                     ref_rays_coords=refc, src_rays_coords=srcc, raw=raw, pixel_mask=pix_mask)
  return out


def dual_branch_stage(models, scene, featmaps_dy, featmaps_st, ray_o, ray_d, uv_grid, pts, z_vals, s_vals,
                      ref_frame_idx, ref_time_embedding, ref_time_offset, which, num_frames,
                      anti_alias_pooling=True, mask_rgb=False, num_vv=0, time_diff_scaled=True,
                      flow_views=None, sf_offsets=(2, -2), dy_shift=0.0):
  """One dynamic+static evaluation at given sample points: the shared body of the coarse stage of
  render_rays_mv (render_ray.py:672-784), of fine_render_rays (:461-597) and of the reference-time pass
  of render_rays_mono (:948-1098).  ``which`` = 'coarse' | 'fine' selects the nets in ``models``."""
  sfx = '' if which == 'coarse' else '_fine'
  sd_dy = models['net_%s_dy' % which]
  sd_st = models['net_%s_st' % which]
  sd_mo = models['motion_mlp' + sfx]
  basis = models['trajectory_basis' + sfx]
  R, S = pts.shape[:2]
  n_last = int(round(S * 0.1))
  t_emb = ref_time_embedding[None, None, :].repeat(R, S, 1)
  xyzt = torch.cat([pts, t_emb], dim=-1).to(torch.get_default_dtype())
  coeff = motion_mlp(sd_mo, xyzt)
  coeff[:, -n_last:, :] *= 0.0
  traj = trajectory_points(coeff, basis, ref_frame_idx)
  seq = [pts + (traj[o] - traj[0]) for o in ref_time_offset]
  for _ in range(num_vv):
    seq.append(pts)
  pts_seq = torch.stack(seq, 0)
  Vs = scene['static_src_rgbs'].shape[1]
  rgb_feat_dy, ray_diff_dy, mask_dy = compute_with_motions(pts, pts_seq, scene['camera'], scene['src_rgbs'],
                                                            scene['src_cameras'], featmaps_dy)
  rgb_feat_st, ray_diff_st, mask_st = compute_with_motions(pts, pts[None].repeat(Vs, 1, 1, 1), scene['camera'],
                                                            scene['static_src_rgbs'], scene['static_src_cameras'],
                                                            featmaps_st)
  pm_dy = mask_dy[..., 0].sum(dim=2) > 1
  pm_st = mask_st[..., 0].sum(dim=2) > 1
  tdiff = torch.from_numpy(np.array(ref_time_offset))
  if time_diff_scaled:
    tdiff = tdiff / float(num_frames)
  tdiff = tdiff[None, None, :, None].expand(R, S, -1, -1)
  ray_dir = F.normalize(ray_d, dim=-1)
  raw_dy = dynamic_net(sd_dy, pts, rgb_feat_dy, ray_dir, ray_diff_dy, tdiff, mask_dy, t_emb, shift=dy_shift)
  raw_st = static_net(sd_st, pts, ref_plucker(ray_o, ray_d), src_plucker(pts, scene['static_src_cameras']),
                      rgb_feat_st, ray_dir, ray_diff_st, mask_st, anti_alias_pooling, mask_rgb)
  out = raw2outputs(raw_dy, raw_st, z_vals, pm_dy, pm_st)
  out_dy = raw2outputs_vanilla(raw_dy, z_vals, pm_dy)
  out_st = raw2outputs_vanilla(raw_st, z_vals, pm_st)
  fv = pts_seq.shape[0] if flow_views is None else flow_views
  out['render_flows'] = compute_optical_flow(out['weights'], pts_seq[:fv], scene['src_cameras'][:, :fv], uv_grid)
  out['s_vals'] = s_vals
  sf_p = torch.sum(out['weights'][..., None] * (traj[sf_offsets[0]] - traj[0]), dim=-2)
  sf_m = torch.sum(out['weights'][..., None] * (traj[sf_offsets[1]] - traj[0]), dim=-2)
  out['exp_sf'] = torch.max(sf_p, sf_m).detach()  # detached in the monocular path (render_ray.py:1096); nothing differentiates the other callers' copy
  return out, out_dy, out_st, dict(raw_dy=raw_dy, raw_st=raw_st, coeff=coeff, pts_seq=pts_seq, pm_dy=pm_dy, pm_st=pm_st)


def render_rays_mv(models, scene, ray_o, ray_d, uv_grid, frame_idx, time_embedding, time_offset, N_samples,
                   N_importance, inv_uniform=True, det=True, anti_alias_pooling=True, mask_rgb=False,
                   t_rand=None, u=None):
  """Nvidia-benchmark path, eval semantics (render_ray.py:600-867 with fine_render_rays :407-597)."""
  num_frames = int(frame_idx / time_embedding)
  pts, z_vals, _ = sample_along_camera_ray(ray_o, ray_d, scene['depth_range'], N_samples, inv_uniform, det, t_rand)
  out_c, _, _, _ = dual_branch_stage(models, scene, scene['featmaps'], scene['static_featmaps'], ray_o, ray_d, uv_grid,
                                     pts, z_vals, None, frame_idx, time_embedding, time_offset, 'coarse', num_frames,
                                     anti_alias_pooling, mask_rgb)
  # the reference's coarse dict has no flows / s_vals / exp_sf (render_ray.py:776-784)
  for k in ('render_flows', 's_vals', 'exp_sf'):
    out_c.pop(k)
  z_all = fine_z_vals(z_vals, out_c['weights'], N_importance, inv_uniform, det, u)
  near_v, far_v = scene['depth_range'][0, 0], scene['depth_range'][0, 1]
  s_all = z_to_s(z_all, near_v, far_v)
  pts_f = z_all.unsqueeze(2) * ray_d.unsqueeze(1) + ray_o.unsqueeze(1)
  out_f, out_f_dy, _, _ = dual_branch_stage(models, scene, scene['featmaps_fine'], scene['static_featmaps_fine'], ray_o,
                                            ray_d, uv_grid, pts_f, z_all, s_all, frame_idx, time_embedding, time_offset,
                                            'fine', num_frames, anti_alias_pooling, mask_rgb)
  return {'outputs_coarse_ref': out_c, 'outputs_fine_ref': out_f, 'outputs_fine_ref_dy': out_f_dy,
          'outputs_fine_anchor': None, 'outputs_fine_anchor_dy': None}


def render_rays_mono_eval(models, scene, ray_o, ray_d, uv_grid, frame_idx, time_embedding, time_offset, N_samples,
                          inv_uniform=True, det=True, anti_alias_pooling=True, mask_rgb=False, num_vv=2, t_rand=None, dy_shift=0.0):
  """Monocular path with is_train=False (render_ray.py:870-1098, 1272-1277): coarse only, time_diff unscaled
  (:1031-1036), flows on the first 6 views (:1077-1082), exp_sf from offsets +-1 (:1086-1096)."""
  num_frames = int(frame_idx / time_embedding)
  pts, z_vals, s_vals = sample_along_camera_ray(ray_o, ray_d, scene['depth_range'], N_samples, inv_uniform, det, t_rand)
  out, out_dy, out_st, _ = dual_branch_stage(models, scene, scene['featmaps'], scene['static_featmaps'], ray_o, ray_d,
                                             uv_grid, pts, z_vals, s_vals, frame_idx, time_embedding, time_offset,
                                             'coarse', num_frames, anti_alias_pooling, mask_rgb, num_vv=num_vv,
                                             time_diff_scaled=False, flow_views=6, sf_offsets=(1, -1), dy_shift=dy_shift)
  return {'outputs_coarse_ref': out, 'outputs_coarse_ref_dy': out_dy, 'outputs_coarse_st': out_st}


def render_rays_mono_train(models, scene, ray_o, ray_d, uv_grid, frame_idx, time_embedding, time_offset, N_samples,
                           inv_uniform=True, det=True, anti_alias_pooling=True, mask_rgb=False, num_vv=2, occ_weights_mode=0,
                           t_rand=None, dy_shift=0.0):
  """Monocular path with is_train=True, forward values (render_ray.py:870-1277): the reference-time pass plus the cross-time
  rendering at the anchor time (:1099-1270).  frame_idx / time_embedding / time_offset are (ref, anchor) pairs; the anchor
  sources are scene['anchor_src_rgbs'], scene['anchor_src_cameras'], scene['featmaps_anchor']."""
  ref_idx, anc_idx = frame_idx
  ref_temb, anc_temb = time_embedding
  ref_off, anc_off = time_offset
  num_frames = int(ref_idx / ref_temb)
  pts, z_vals, s_vals = sample_along_camera_ray(ray_o, ray_d, scene['depth_range'], N_samples, inv_uniform, det, t_rand)
  out, out_dy, out_st, st = dual_branch_stage(models, scene, scene['featmaps'], scene['static_featmaps'], ray_o, ray_d, uv_grid, pts,
                                              z_vals, s_vals, ref_idx, ref_temb, ref_off, 'coarse', num_frames, anti_alias_pooling,
                                              mask_rgb, num_vv=num_vv, time_diff_scaled=False, flow_views=6, sf_offsets=(1, -1), dy_shift=dy_shift)
  basis = models['trajectory_basis']
  R, S = pts.shape[:2]
  n_last = int(round(S * 0.1))
  traj = trajectory_points(st['coeff'], basis, ref_idx)                      # ref_traj_pts_dict, offsets -3..3 (:965-979)
  sf_seq = torch.stack([traj[o] - traj[o - 1] for o in (-2, -1, 0, 1, 2, 3)], 0)   # :1101-1105
  pts_anchor = pts + (traj[anc_idx - ref_idx] - traj[0])                      # :1109-1112
  t_anc = anc_temb[None, None, :].repeat(R, S, 1).to(torch.get_default_dtype())
  coeff_a = motion_mlp(models['motion_mlp'], torch.cat([pts_anchor, t_anc], -1).to(torch.get_default_dtype()))
  coeff_a[:, -n_last:, :] *= 0.0
  B = basis.shape[1]
  cx, cy, cz = coeff_a[..., :B], coeff_a[..., B:2 * B], coeff_a[..., 2 * B:3 * B]
  traj_a0 = compute_traj_pts(cx, cy, cz, basis[None, None, anc_idx, :])
  seq, tr_ref, tr_anc = [], [], []
  for off in anc_off:                                                          # :1147-1168
    ref_offset = anc_idx + off - ref_idx
    tp = pts_anchor + (compute_traj_pts(cx, cy, cz, basis[None, None, anc_idx + off, :]) - traj_a0)
    seq.append(tp)
    if ref_offset not in traj:
      continue
    tr_anc.append(tp)
    tr_ref.append(pts + traj[ref_offset] - traj[0])
  for _ in range(num_vv):
    seq.append(pts_anchor)
  pts_seq_a = torch.stack(seq, 0)
  rf_a, rd_a, mk_a = compute_with_motions(pts, pts_seq_a, scene['camera'], scene['anchor_src_rgbs'], scene['anchor_src_cameras'],
                                          scene['featmaps_anchor'])
  tdiff = torch.from_numpy(np.array(anc_off))[None, None, :, None].expand(R, S, -1, -1)
  pm_a = mk_a[..., 0].sum(dim=2) > 0                                           # :1197-1199 (one observation is enough here)
  ray_dir = F.normalize(ray_d, dim=-1)
  raw_a = dynamic_net(models['net_coarse_dy'], pts_anchor, rf_a, ray_dir, rd_a, tdiff, mk_a, t_anc, shift=dy_shift)
  out_a = raw2outputs(raw_a, st['raw_st'], z_vals, pm_a, st['pm_st'])
  out_a_dy = raw2outputs_vanilla(raw_a, z_vals, pm_a)
  occ_dy = (out_dy['weights'] - out_a_dy['weights']).detach()  # disocclusion scores carry no gradient (render_ray.py:1216, :1243)
  if occ_weights_mode == 0:
    key = 'weights_dy' if abs(ref_idx - anc_idx) > 1 else 'weights'
  elif occ_weights_mode == 1:
    key = 'weights_dy'
  elif occ_weights_mode == 2:
    key = 'weights'
  else:
    raise NotImplementedError
  occ = (out[key] - out_a[key]).detach()
  out_a['occ_weights'] = 1.0 - occ.abs()
  out_a['occ_weight_map'] = 1.0 - occ.sum(dim=1).abs()
  out_a['pts_traj_ref'] = torch.stack(tr_ref, 0)
  out_a['pts_traj_anchor'] = torch.stack(tr_anc, 0)
  out_a['sf_seq'] = sf_seq
  out_a_dy['occ_weights'] = 1.0 - occ_dy.abs()
  out_a_dy['occ_weight_map'] = 1.0 - occ_dy.sum(dim=1).abs()
  return {'outputs_coarse_ref': out, 'outputs_coarse_ref_dy': out_dy, 'outputs_coarse_st': out_st,
          'outputs_coarse_anchor': out_a, 'outputs_coarse_anchor_dy': out_a_dy}

