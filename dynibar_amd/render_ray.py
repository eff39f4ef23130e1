"""Drop-in for the reference's ``ibrnet/render_ray.py``: ``render_rays_mv`` / ``render_rays_mono`` (and the helpers the
reference scripts import) with the reference signatures and output dictionaries (reference render_ray.py:600-867, :870-1277),
executed by the HIP kernels of libdynibar_hip.so.

``model`` is the reference's ``DynibarFF`` / ``DynibarMono`` object (or any object with the same attributes): its sub-networks
may be ``nn.Module``s (optionally ``DataParallel``-wrapped) or plain state dicts.  Their weights are packed into MFMA operand
tiles once and re-packed only when a parameter's version counter changes.

Scope: forward rendering, plus two slices of the backward pass.  Under grad mode, when a net's parameters or its
feature maps require grad, ``render_rays_mono`` evaluates DynibarStatic / DynibarDynamic and both compositing functions through
``dynibar_amd.train_static`` / ``train_dynamic`` (HIP training kernels with hand-written backward): the colour / depth / weight
outputs of ``outputs_coarse_st``, ``outputs_coarse_ref``, ``outputs_coarse_ref_dy`` (and of the anchor pass) then carry a graph to both
nets' parameters and to the feature maps, and -- third slice, ``train_motion`` -- the motion path does too: MotionMLP, the trajectory
points (``pts_traj_*``, ``sf_seq``), the gather w.r.t. the displaced points and ``render_flows`` are autograd Functions over HIP kernels,
so ``motion_mlp`` and ``trajectory_basis`` receive their gradients.  The reference's static bootstrap stage (train.py:116-199) and its
main loop's ``loss.backward()`` (train.py:283-467) run on this path; ``exp_sf`` and the disocclusion weights are detached exactly where
the reference detaches them.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops, train_dynamic, train_motion, train_static

USE_DISTANCE = False   # reference render_ray.py:14-16 (module constants; the kernels implement exactly this setting)
USE_SOFTPLUS = True


# ----------------------------------------------------------------------------------------------------------------------
# model adapter
# ----------------------------------------------------------------------------------------------------------------------
def _unwrap(net):
  return net.module if hasattr(net, 'module') and not isinstance(net, dict) else net


def _state_dict(net):
  net = _unwrap(net)
  return net.state_dict() if hasattr(net, 'state_dict') else net


def _version(net):
  net = _unwrap(net)
  if hasattr(net, 'parameters'):
    return tuple((p.data_ptr(), p._version) for p in net.parameters())
  return tuple((k, getattr(v, '_version', 0), v.data_ptr() if isinstance(v, torch.Tensor) else id(v)) for k, v in sorted(net.items()))


class _Packed:
  """Packed networks of one ``model`` on one device, keyed by attribute name."""

  def __init__(self):
    self.entries = {}

  def get(self, model, name, device, build):
    net = getattr(model, name)
    ver = (_version(net), str(device))
    ent = self.entries.get(name)
    if ent is None or ent[0] != ver:
      ent = (ver, build(_state_dict(net)))
      self.entries[name] = ent
    return ent[1]


_NET_CACHE = OrderedDict()  # explicit nets handed to fine_render_rays: (id(net), device) -> (version, packed, net)


def _packed_net(net, device, build):
  """Like _Packed.get for a network that does not hang off a model object; the entry keeps `net` alive so its id cannot be recycled."""
  key = (id(net), str(device))
  ver = _version(net)
  ent = _NET_CACHE.get(key)
  if ent is None or ent[0] != ver:
    ent = (ver, build(_state_dict(net)), net)
    _NET_CACHE[key] = ent
    while len(_NET_CACHE) > 32:
      _NET_CACHE.popitem(last=False)
  return ent[1]


def _packed(model):
  p = getattr(model, '_dynibar_amd_packed', None)
  if p is None:
    p = _Packed()
    try:
      model._dynibar_amd_packed = p
    except AttributeError:
      pass
  return p


def _flag(net, args, name, default):
  net = _unwrap(net)
  if hasattr(net, name):
    return bool(getattr(net, name))
  return bool(getattr(args, name, default))


def _static_net(model, name, args, device):
  net = getattr(model, name)
  aa, mr = _flag(net, args, 'anti_alias_pooling', True), _flag(net, args, 'mask_rgb', False)
  return _packed(model).get(model, name, device, lambda sd: ops.StaticNet(sd, device, aa, mr))


def _dynamic_net(model, name, device):
  shift = float(getattr(_unwrap(getattr(model, name)), 'shift', 0.0))
  return _packed(model).get(model, name, device, lambda sd: ops.DynamicNet(sd, device, shift=shift))


def _motion_mlp(model, name, device, num_basis):
  div = float(getattr(_unwrap(getattr(model, name)), 'sf_mag_div', 1.0))
  return _packed(model).get(model, name, device, lambda sd: ops.MotionMLP(sd, device, num_basis, div))


# ----------------------------------------------------------------------------------------------------------------------
# reference helpers with the reference signatures
# ----------------------------------------------------------------------------------------------------------------------
def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False):
  """(render_ray.py:67-131) -> pts [R,S,3], z_vals [R,S], s_vals [R,S]."""
  t_rand = None if det else torch.rand(ray_o.shape[0], N_samples, device=ray_o.device)
  return ops.sample_along_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform, t_rand)


def z_to_s(z_vals, near_depth_value, far_depth_value):
  """(render_ray.py:399-404); callers that also need the points use ops.points_from_z, which fuses both."""
  return ((1.0 / z_vals) - (1.0 / near_depth_value)) / (1.0 / far_depth_value - 1.0 / near_depth_value)


def fine_z_vals(z_vals, weights, N_importance, inv_uniform, det):
  """Coarse weights -> sorted coarse+fine depths: sample_pdf + cat + sort of render_ray.py:790-821 in one kernel."""
  u = None if det else torch.rand(z_vals.shape[0], N_importance, device=z_vals.device)
  return ops.fine_samples(z_vals, weights, N_importance, inv_uniform, u)[0]


def sample_pdf(bins, weights, N_samples, det=False):
  """(render_ray.py:19-64) bins [R,M+1], weights [R,M] -> samples [R,N_samples].  Like the reference it adds 1e-5 to `weights` IN PLACE."""
  assert weights.dtype == torch.float32 and weights.is_contiguous(), 'sample_pdf updates `weights` in place: pass a contiguous fp32 tensor'
  k = ops._Keep()
  R, M = weights.shape
  u = None if det else torch.rand(R, N_samples, device=weights.device)
  out = torch.empty((R, N_samples), dtype=torch.float32, device=weights.device)
  ops.call('dyn_sample_pdf', k(bins), ops.ptr(weights), k(u), R, M, int(N_samples), ops.ptr(out), ops.stream_of(out))
  return out


def compute_traj_pts(raw_coeff_x, raw_coeff_y, raw_coeff_z, trajectory_basis_i):
  """(render_ray.py:361-369) sum_b coeff_{x,y,z}[..., b] * basis_i[b] -> [..., 3]."""
  B = raw_coeff_x.shape[-1]
  shp = raw_coeff_x.shape[:-1]
  coeff = torch.cat([raw_coeff_x, raw_coeff_y, raw_coeff_z], dim=-1).reshape(-1, 1, 3 * B)
  basis = torch.cat([trajectory_basis_i.reshape(1, B).float(), torch.zeros(1, B, device=coeff.device)], dim=0)  # row 1 = zeros: the "reference" row
  zero = torch.zeros((coeff.shape[0], 1, 3), dtype=torch.float32, device=coeff.device)
  return ops.trajectory_points(coeff, basis, zero, [0], 1)[0].reshape(tuple(shp) + (3,))


def compute_optical_flow(outputs_coarse, raw_pts_3d_seq, src_cameras, uv_grid):
  """(render_ray.py:333-358) expected 2-D flow into every source view: [V,R,2]."""
  k = ops._Keep()
  w = outputs_coarse['weights']
  V, R, S = raw_pts_3d_seq.shape[:3]
  cams = src_cameras.squeeze(0)
  proj = torch.empty((V, 16), dtype=torch.float32, device=w.device)
  ops.call('dyn_prepare_cameras', k(cams), V, None, ops.ptr(proj), None, ops.stream_of(proj))
  flows = torch.empty((V, R, 2), dtype=torch.float32, device=w.device)
  ops.call('dyn_render_flows', k(w), k(raw_pts_3d_seq), ops.ptr(proj), k(uv_grid), R, S, V, ops.ptr(flows), ops.stream_of(flows))
  return flows


def compute_ref_plucker_coordinate(ray_o, ray_d):
  """(render_ray.py:372-377) [R,6] = [normalize(d), o x normalize(d)]."""
  k = ops._Keep()
  out = torch.empty((ray_o.shape[0], 6), dtype=torch.float32, device=ray_o.device)
  ops.call('dyn_plucker_ref', k(ray_o), k(ray_d), ray_o.shape[0], ops.ptr(out), ops.stream_of(out))
  return out


def compute_src_plucker_coordinate(pts, src_cameras):
  """(render_ray.py:380-396) pts [R,S,3] or [V,R,S,3], src_cameras [1,V,34] -> [R,S,V,6]; the moment is crossed over the axis the reference's
  ``torch.cross`` without ``dim`` picks for this shape (the first of size 3: csrc/dyn_device.h)."""
  k = ops._Keep()
  cams = src_cameras[0]
  V = cams.shape[0]
  per_view = pts.dim() == 4
  R, S = pts.shape[-3], pts.shape[-2]
  out = torch.empty((R, S, V, 6), dtype=torch.float32, device=pts.device)
  ops.call('dyn_plucker_src', k(pts), int(per_view), k(cams), R, S, V, ops.ptr(out), ops.stream_of(out))
  return out


def _as_out(d, keys):
  return OrderedDict((k, d[k]) for k in keys)


def raw2outputs_vanilla(raw, z_vals, mask, white_bkgd=False):
  """(render_ray.py:134-211) mask: [R,S] bool / 0-1."""
  out = ops.composite(raw, z_vals, mask.float())
  out['mask'] = out['mask'] > 0
  return _as_out(out, ('rgb', 'depth', 'weights', 'mask', 'alpha', 'z_vals'))


def raw2outputs(raw_dy, raw_static, z_vals, mask_dy, mask_static, raw_noise_std=0.0, white_bkgd=False):
  """(render_ray.py:214-330); raw_noise_std is accepted and unused, as in the reference."""
  out = ops.composite(raw_dy, z_vals, mask_dy.float(), raw_static, mask_static.float())
  out['mask'] = out['mask'] > 0
  return _as_out(out, ('rgb', 'rgb_static', 'rgb_dy', 'depth', 'alpha_dy', 'weights_dy', 'weights_st', 'alpha', 'weights', 'mask', 'z_vals'))


# ----------------------------------------------------------------------------------------------------------------------
# one dynamic + static evaluation at given sample points (the body shared by render_ray.py:672-784, :461-597, :948-1098)
# ----------------------------------------------------------------------------------------------------------------------
def _needs_graph(*tensors):
  return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


FUSED_TRAJ = os.environ.get('DYNIBAR_FUSED_TRAJ', '1') != '0'  # developer A/B: 0 materialises the displaced points [V,R,S,3] like rounds 1-5
_ROWS_DEV = {}


def _rows_on_device(rows, dev):
  """the basis rows of the dynamic source views as a small int32 device tensor (cached: a frame asks for the same rows in every chunk)"""
  key = (tuple(rows), dev.type, dev.index)
  t = _ROWS_DEV.get(key)
  if t is None:
    if len(_ROWS_DEV) > 64:
      if dev.type == 'cuda':
        torch.cuda.synchronize(dev)  # kernels of earlier chunks (other streams) may still read the cached arrays: nothing is freed under them
      _ROWS_DEV.clear()
    t = _ROWS_DEV[key] = torch.tensor(rows, dtype=torch.int32, device=dev)
  return t


def _traj(coeff, basis, pts, rows, row_ref):
  """k_trajectory_points; with a graph (train_motion.TrajectoryFunction) when the coefficients, the basis or the points carry one"""
  if _needs_graph(coeff, basis, pts):
    return train_motion.trajectory_points(coeff, basis, pts, rows, row_ref)
  return ops.trajectory_points(coeff, basis, pts, rows, row_ref)


def _coeff(model, name, dev, num_basis, pts, time, n_zero):
  """MotionMLP coefficients; through the training kernels when MotionMLP's parameters (or the points) take part in a graph"""
  net = getattr(model, name)
  if train_motion.wants_grad(net) or _needs_graph(pts):
    return train_motion.motion_coeff(net, pts, time, n_zero, float(getattr(_unwrap(net), 'sf_mag_div', 1.0)))
  return _motion_mlp(model, name, dev, num_basis)(pts, time, n_zero)


def _gather(views, featmaps, R, S, thresh, xyz=None, **kw):
  if _needs_graph(featmaps, xyz):
    return train_motion.gather(views, featmaps, R, S, xyz=xyz, pix_mask_thresh=thresh, **kw)
  return ops.project_gather(views, R, S, xyz=xyz, pix_mask_thresh=thresh, **kw)


def _dual_branch(model, names, args, projector, ray_batch, featmaps_dy, featmaps_st, pts, z_vals, ref_frame_idx, ref_time_embedding,
                 ref_time_offset, num_vv=0):
  dev = pts.device
  ray_o, ray_d = ray_batch['ray_o'], ray_batch['ray_d']
  R, S = z_vals.shape
  basis = getattr(model, names['basis'])
  if not _needs_graph(basis):
    basis = basis.detach()
  num_basis = basis.shape[1]
  n_last = int(round(S * 0.1))
  time = ref_time_embedding.reshape(-1)[:1].to(dev).float()
  # raw_coeff[:, -n_last:, :] *= 0 : with n_last == 0 the reference's slice [-0:] is the whole array (render_ray.py:684)
  coeff = _coeff(model, names['motion'], dev, num_basis, pts, time, n_last if n_last > 0 else S)
  nf = basis.shape[0]
  rows = [(int(ref_frame_idx) + int(o)) % nf for o in ref_time_offset] + [-1] * num_vv  # negative rows wrap like basis[idx] does
  # compute_traj_pts (render_ray.py:361-369, :691-725).  Without a graph the displaced points are never materialised: the gather kernel (and the flows) form
  # them from the coefficients with the same sums as k_trajectory_points, bit for bit (csrc/dyn_geometry.hip: traj_displace) -- [V,R,S,3] neither written nor read.
  fused = FUSED_TRAJ and not _needs_graph(coeff, basis, pts, featmaps_dy)
  traj = None
  if fused:
    traj = (coeff, basis.float().contiguous(), _rows_on_device(rows, dev), int(ref_frame_idx) % nf)
    pts_seq = None
  else:
    pts_seq = _traj(coeff, basis, pts, rows, int(ref_frame_idx) % nf)
  views_dy = projector.source_views(ray_batch['camera'], ray_batch['src_rgbs'], ray_batch['src_cameras'], featmaps_dy)
  views_st = projector.source_views(ray_batch['camera'], ray_batch['static_src_rgbs'], ray_batch['static_src_cameras'], featmaps_st)
  assert views_dy.V == len(rows), 'one time offset (or virtual view) per dynamic source view'
  # sample masks: at least 2 observations (render_ray.py:736-741), counted by the gather kernel itself
  # (the dynamic branch has no use for ray_diff -- DynibarDynamic takes none, mlp_network.py:236-317 --: the forward-only gather does not write it)
  if fused:
    rgb_feat_dy, _, mask_dy, pm_dy = ops.project_gather(views_dy, R, S, pts_st=pts, pix_mask_thresh=1.0, traj=traj, want_ray_diff=False)
  elif _needs_graph(featmaps_dy, pts_seq):
    rgb_feat_dy, _, mask_dy, pm_dy = _gather(views_dy, featmaps_dy, R, S, 1.0, xyz=pts_seq, pts_st=pts)
  else:
    rgb_feat_dy, _, mask_dy, pm_dy = ops.project_gather(views_dy, R, S, pts_st=pts, xyz=pts_seq, pix_mask_thresh=1.0, want_ray_diff=False)
  rgb_feat_st, ray_diff_st, mask_st, pm_st = _gather(views_st, featmaps_st, R, S, 1.0, ray_o=ray_o, ray_d=ray_d, z_vals=z_vals)
  net_dy = getattr(model, names['dy'])
  if train_dynamic.wants_grad(net_dy, rgb_feat_dy):
    # training: graph to DynibarDynamic's parameters, the gathered features (maps, displaced points) and the points
    raw_dy = train_dynamic.dynamic_raw(net_dy, float(getattr(_unwrap(net_dy), 'shift', 0.0)), rgb_feat_dy, ray_d, pts, mask_dy, time)
  else:
    raw_dy = _dynamic_net(model, names['dy'], dev)(ray_d, pts, rgb_feat_dy, mask_dy, time)
  net_st = getattr(model, names['st'])
  if train_static.wants_grad(net_st, rgb_feat_st):
    # training: the same network on the kernels that keep their activations, with an autograd graph to the parameters and the maps
    flags = (_flag(net_st, args, 'anti_alias_pooling', True), _flag(net_st, args, 'mask_rgb', False))
    raw_st = train_static.static_raw(net_st, flags, views_st, rgb_feat_st, ray_o, ray_d, pts, ray_diff_st, mask_st)
  else:
    raw_st = _static_net(model, names['st'], args, dev)(views_st, ray_o, ray_d, pts, rgb_feat_st, ray_diff_st, mask_st)
  return dict(raw_dy=raw_dy, raw_st=raw_st, pm_dy=pm_dy, pm_st=pm_st, coeff=coeff, pts_seq=pts_seq, views_dy=views_dy, basis=basis, traj=traj, pts=pts, n_views=len(rows))


def _finish(stage, z_vals, keys2, keys1):
  if stage['raw_dy'].requires_grad or stage['raw_st'].requires_grad:
    return train_dynamic.composite_dual(stage['raw_dy'], stage['raw_st'], z_vals, stage['pm_dy'], stage['pm_st'])  # same keys, with graph
  out = ops.composite(stage['raw_dy'], z_vals, stage['pm_dy'], stage['raw_st'], stage['pm_st'])
  out['mask'] = out['mask'] > 0
  return _as_out(out, keys2)


_KEYS2 = ('rgb', 'rgb_static', 'rgb_dy', 'depth', 'alpha_dy', 'weights_dy', 'weights_st', 'alpha', 'weights', 'mask', 'z_vals')
_KEYS1 = ('rgb', 'depth', 'weights', 'mask', 'alpha', 'z_vals')


def _vanilla(raw, z_vals, pm):
  if raw.requires_grad:
    return train_static.composite_vanilla(raw, z_vals, pm)
  out = ops.composite(raw, z_vals, pm)
  out['mask'] = out['mask'] > 0
  return _as_out(out, _KEYS1)


def _motion_outputs(out, stage, ray_batch, ref_frame_idx, sf_off, flow_views=None):
  """render_flows (render_ray.py:333-358) and exp_sf (:584-595 / :1086-1096) of a composited stage."""
  R, S = out['weights'].shape
  views = stage['views_dy']
  fv = stage['n_views'] if flow_views is None else min(flow_views, stage['n_views'])
  uv = ray_batch['uv_grid'].float().contiguous()
  if stage['pts_seq'] is None:  # the fused eval form: the flows form the displaced points themselves (the first fv views: a prefix of the rows)
    coeff, basis, rows_dev, ref = stage['traj']
    flows = torch.empty((fv, R, 2), dtype=torch.float32, device=out['weights'].device)
    k = ops._Keep()
    ops.call('dyn_render_flows_traj', k(out['weights']), k(stage['pts']), k(coeff), k(basis), int(basis.shape[1]), rows_dev.data_ptr(), ref, ops.ptr(views.proj), k(uv),
             R, S, fv, ops.ptr(flows), ops.stream_of(flows))
  elif _needs_graph(out['weights'], stage['pts_seq']):
    flows = train_motion.render_flows(out['weights'], stage['pts_seq'][:fv], views.proj, uv)  # (a slice of leading views is contiguous)
  else:
    flows = torch.empty((fv, R, 2), dtype=torch.float32, device=out['weights'].device)
    ops.call('dyn_render_flows', ops.ptr(out['weights']), ops.ptr(stage['pts_seq']), ops.ptr(views.proj), ops.ptr(uv), R, S, fv, ops.ptr(flows),
             ops.stream_of(flows))
  out['render_flows'] = flows
  exp_sf = torch.empty((R, 3), dtype=torch.float32, device=flows.device)
  basis = stage['basis'].detach().float().contiguous()  # exp_sf is detached in the reference (:1096)
  ops.call('dyn_expected_scene_flow', ops.ptr(out['weights'].detach()), ops.ptr(stage['coeff'].detach()), ops.ptr(basis), R, S, basis.shape[1],
           (int(ref_frame_idx) + sf_off) % basis.shape[0], (int(ref_frame_idx) - sf_off) % basis.shape[0], int(ref_frame_idx) % basis.shape[0],
           ops.ptr(exp_sf), ops.stream_of(exp_sf))
  return exp_sf


def fine_render_rays(projector, ray_batch, featmaps, pts_ref, z_vals, s_vals, ref_time_embedding, anchor_time_embedding, ref_frame_idx,
                     anchor_frame_idx, ref_time_offset, anchor_time_offset, net_dy, net_st, motion_mlp, trajectory_basis, occ_weights_mode,
                     is_train):
  """Reference render_ray.py:407-597: the fine pass of the Nvidia path on explicit networks -> (outputs_ref, outputs_ref_dy, None, None)."""
  import types
  holder = types.SimpleNamespace(dy=net_dy, st=net_st, motion=motion_mlp, basis=trajectory_basis, _dynibar_amd_packed=_ExplicitNets())
  names = dict(dy='dy', st='st', motion='motion', basis='basis')
  stage = _dual_branch(holder, names, None, projector, ray_batch, featmaps[0], featmaps[2], pts_ref, z_vals, ref_frame_idx, ref_time_embedding,
                       ref_time_offset)
  out = _finish(stage, z_vals, _KEYS2, _KEYS1)
  out_dy = _vanilla(stage['raw_dy'], z_vals, stage['pm_dy'])
  exp_sf = _motion_outputs(out, stage, ray_batch, ref_frame_idx, 2)
  out['s_vals'] = s_vals
  out['exp_sf'] = exp_sf
  return out, out_dy, None, None


class _ExplicitNets:
  """_Packed look-alike whose entries live in the module-level cache keyed by the network objects themselves."""

  def get(self, model, name, device, build):
    return _packed_net(getattr(model, name), device, build)


# ----------------------------------------------------------------------------------------------------------------------
# render_rays_mv  (Nvidia dynamic-scenes path: coarse + fine)
# ----------------------------------------------------------------------------------------------------------------------
def render_rays_mv(frame_idx, time_embedding, time_offset, ray_batch, model, projector, coarse_featmaps, fine_featmaps, N_samples, args,
                   inv_uniform=False, N_importance=0, raw_noise_std=0.0, det=False, white_bkgd=False, is_train=True):
  """Reference render_ray.py:600-867.  Returns the same dictionary: outputs_coarse_ref, outputs_fine_ref, outputs_fine_ref_dy,
  outputs_fine_anchor (None), outputs_fine_anchor_dy (None)."""
  ref_frame_idx, ref_time_embedding, ref_time_offset = frame_idx[0], time_embedding[0], time_offset[0]
  ret = {'outputs_coarse': None, 'outputs_fine': None}
  ray_o, ray_d = ray_batch['ray_o'], ray_batch['ray_d']
  pts, z_vals, _ = sample_along_camera_ray(ray_o, ray_d, ray_batch['depth_range'], N_samples, inv_uniform, det)
  names = dict(dy='net_coarse_dy', st='net_coarse_st', motion='motion_mlp', basis='trajectory_basis')
  stage = _dual_branch(model, names, args, projector, ray_batch, coarse_featmaps[0], coarse_featmaps[2], pts, z_vals, ref_frame_idx,
                       ref_time_embedding, ref_time_offset)
  out_c = _finish(stage, z_vals, _KEYS2, _KEYS1)
  ret['outputs_coarse_ref'] = out_c
  assert N_importance > 0
  z_all = fine_z_vals(z_vals, out_c['weights'], N_importance, inv_uniform, det)
  pts_f, s_all = ops.points_from_z(ray_o, ray_d, z_all, ray_batch['depth_range'])
  names = dict(dy='net_fine_dy', st='net_fine_st', motion='motion_mlp_fine', basis='trajectory_basis_fine')
  stage = _dual_branch(model, names, args, projector, ray_batch, fine_featmaps[0], fine_featmaps[2], pts_f, z_all, ref_frame_idx,
                       ref_time_embedding, ref_time_offset)
  out_f = _finish(stage, z_all, _KEYS2, _KEYS1)
  out_f_dy = _vanilla(stage['raw_dy'], z_all, stage['pm_dy'])
  exp_sf = _motion_outputs(out_f, stage, ray_batch, ref_frame_idx, 2)
  out_f['s_vals'] = s_all
  out_f['exp_sf'] = exp_sf
  ret['outputs_fine_ref'] = out_f
  ret['outputs_fine_ref_dy'] = out_f_dy
  ret['outputs_fine_anchor'] = None
  ret['outputs_fine_anchor_dy'] = None
  return ret


# ----------------------------------------------------------------------------------------------------------------------
# render_rays_mono  (monocular path: coarse only)
# ----------------------------------------------------------------------------------------------------------------------
def render_rays_mono(frame_idx, time_embedding, time_offset, ray_batch, model, featmaps, projector, N_samples, args, inv_uniform=False,
                     N_importance=0, raw_noise_std=0.0, det=False, white_bkgd=False, is_train=True, num_vv=2):
  """Reference render_ray.py:870-1277: outputs_coarse_ref, outputs_coarse_ref_dy, outputs_coarse_st and, with is_train=True, the
  cross-time rendering at the anchor frame (outputs_coarse_anchor, outputs_coarse_anchor_dy; :1099-1270).  Under grad mode the
  returned tensors carry the autograd graph train.py differentiates (its nodes are the backward kernels of train_static / train_dynamic /
  train_motion); under torch.no_grad() the forward-only inference kernels run."""
  ref_frame_idx, ref_time_embedding, ref_time_offset = frame_idx[0], time_embedding[0], time_offset[0]
  ray_o, ray_d = ray_batch['ray_o'], ray_batch['ray_d']
  pts, z_vals, s_vals = sample_along_camera_ray(ray_o, ray_d, ray_batch['depth_range'], N_samples, inv_uniform, det)
  names = dict(dy='net_coarse_dy', st='net_coarse_st', motion='motion_mlp', basis='trajectory_basis')
  stage = _dual_branch(model, names, args, projector, ray_batch, featmaps[0], featmaps[2], pts, z_vals, ref_frame_idx, ref_time_embedding,
                       ref_time_offset, num_vv=num_vv)
  out = _finish(stage, z_vals, _KEYS2, _KEYS1)
  out_st = _vanilla(stage['raw_st'], z_vals, stage['pm_st'])  # under grad mode: the graph the static bootstrap stage differentiates
  out_dy = _vanilla(stage['raw_dy'], z_vals, stage['pm_dy'])
  exp_sf = _motion_outputs(out, stage, ray_batch, ref_frame_idx, 1, flow_views=6)
  out['s_vals'] = s_vals
  out['exp_sf'] = exp_sf
  ret = {'outputs_coarse': None, 'outputs_fine': None}
  if is_train:
    out_a, out_a_dy = _anchor_pass(model, names, args, projector, ray_batch, featmaps[1], stage, pts, z_vals, out, out_dy, frame_idx,
                                   time_embedding[1], time_offset[1], num_vv)
    ret['outputs_coarse_anchor'] = out_a
    ret['outputs_coarse_anchor_dy'] = out_a_dy
  ret['outputs_coarse_ref'] = out
  ret['outputs_coarse_ref_dy'] = out_dy
  ret['outputs_coarse_st'] = out_st
  return ret


def _anchor_pass(model, names, args, projector, ray_batch, featmaps_anchor, stage, pts, z_vals, out_ref, out_ref_dy, frame_idx,
                 anchor_time_embedding, anchor_time_offset, num_vv):
  """Cross-time rendering for temporal consistency (render_ray.py:1099-1270): the reference-time samples are moved to the anchor
  frame along their own trajectories, the motion MLP is evaluated again there, the displaced points are projected into the anchor
  frame's source views and rendered with the same dynamic net; disocclusion weights compare the two renderings' sample weights."""
  dev = pts.device
  R, S = z_vals.shape
  basis = stage['basis']
  nf, num_basis = basis.shape[0], basis.shape[1]
  r, a = int(frame_idx[0]), int(frame_idx[1])
  if not -3 <= a - r <= 3:
    raise KeyError(a - r)  # the reference's trajectory dictionary holds offsets -3..3 only (render_ray.py:965-979, :1109-1112)
  n_last = int(round(S * 0.1))
  coeff = stage['coeff']
  # scene flow between consecutive frames around the reference time (:1101-1105): differences of (traj[o] - traj[0]), o = -3..3
  rel = _traj(coeff, basis, torch.zeros_like(pts), [(r + o) % nf for o in range(-3, 4)], r % nf)
  sf_seq = rel[1:7] - rel[0:6]
  pts_anchor = _traj(coeff, basis, pts, [a % nf], r % nf)[0]
  time_a = anchor_time_embedding.reshape(-1)[:1].to(dev).float()
  coeff_a = _coeff(model, names['motion'], dev, num_basis, pts_anchor, time_a, n_last if n_last > 0 else S)
  rows_a = [(a + int(o)) % nf for o in anchor_time_offset] + [-1] * num_vv
  pts_seq_a = _traj(coeff_a, basis, pts_anchor, rows_a, a % nf)
  # the trajectory of the reference-time point and of its anchor-time correspondence at the frames both passes look at (:1147-1168)
  both = [(i, a + int(o) - r) for i, o in enumerate(anchor_time_offset) if -3 <= a + int(o) - r <= 3]
  pts_traj_anchor = torch.stack([pts_seq_a[i] for i, _ in both], 0)
  pts_traj_ref = _traj(coeff, basis, pts, [(r + ro) % nf for _, ro in both], r % nf)
  views_a = projector.source_views(ray_batch['camera'], ray_batch['anchor_src_rgbs'], ray_batch['anchor_src_cameras'], featmaps_anchor)
  assert views_a.V == len(rows_a), 'one time offset (or virtual view) per anchor source view'
  rgb_feat_a, _, mask_a, pm_a = _gather(views_a, featmaps_anchor, R, S, 0.0, xyz=pts_seq_a, pts_st=pts)  # one observation is enough here (:1197-1199)
  net_dy = getattr(model, names['dy'])
  if train_dynamic.wants_grad(net_dy, rgb_feat_a) or _needs_graph(pts_anchor):
    raw_a = train_dynamic.dynamic_raw(net_dy, float(getattr(_unwrap(net_dy), 'shift', 0.0)), rgb_feat_a, ray_batch['ray_d'], pts_anchor, mask_a, time_a)
  else:
    raw_a = _dynamic_net(model, names['dy'], dev)(ray_batch['ray_d'], pts_anchor, rgb_feat_a, mask_a, time_a)
  out_a = _finish(dict(raw_dy=raw_a, raw_st=stage['raw_st'], pm_dy=pm_a, pm_st=stage['pm_st']), z_vals, _KEYS2, _KEYS1)
  out_a_dy = _vanilla(raw_a, z_vals, pm_a)
  occ_dy = (out_ref_dy['weights'] - out_a_dy['weights']).detach()  # disocclusion scores are detached in the reference (:1216, :1243)
  mode = int(getattr(args, 'occ_weights_mode', 0))
  if mode == 0:    # mix-mode: composite-dy weights when the anchor is more than one frame away, full weights otherwise
    key = 'weights_dy' if abs(r - a) > 1 else 'weights'
  elif mode == 1:  # composite-dy
    key = 'weights_dy'
  elif mode == 2:  # full
    key = 'weights'
  else:
    raise NotImplementedError
  occ = (out_ref[key] - out_a[key]).detach()
  out_a['occ_weights'] = 1.0 - occ.abs()
  out_a['occ_weight_map'] = 1.0 - occ.sum(dim=1).abs()
  out_a['pts_traj_ref'] = pts_traj_ref
  out_a['pts_traj_anchor'] = pts_traj_anchor
  out_a['sf_seq'] = sf_seq
  out_a_dy['occ_weights'] = 1.0 - occ_dy.abs()
  out_a_dy['occ_weight_map'] = 1.0 - occ_dy.sum(dim=1).abs()
  return out_a, out_a_dy
