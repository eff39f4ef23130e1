"""Tensor-level wrappers of the C-ABI entry points (one function per kernel launch).

PyTorch is used for device memory and streams only; every function below allocates its outputs on the inputs' device
and launches exactly the HIP kernels named in its docstring on torch's current stream.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import call, params, ptr, stream_of


def _f32c(t):
  if t is None:
    return None
  if t.dtype != torch.float32:
    t = t.float()
  return t if t.is_contiguous() else t.contiguous()


class _Keep:
  """Converts call arguments to contiguous fp32 and keeps the converted tensors alive until the launch has been enqueued
  (a temporary made by .float()/.contiguous() must outlive the pointer taken from it)."""

  def __init__(self):
    self.held = []

  def __call__(self, t, dtype=torch.float32):
    if t is None:
      return None
    if dtype == torch.float32:
      t = _f32c(t)
    self.held.append(t)
    return ptr(t, dtype)


def _channels_last_view(fm):
  """[V,F,Hf,Wf] tensor whose memory is already [V,Hf,Wf,F] (what dynibar_amd.feature_network returns) -> that contiguous tensor, else None."""
  if fm.dim() == 4 and fm.dtype == torch.float32:
    cl = fm.permute(0, 2, 3, 1)
    if cl.is_contiguous():
      return cl
  return None


class SourceViews:
  """Per-target-view prepared source data for one branch: projection matrices, camera centres, channels-last maps.

  Built once per (target view, branch) and reused by every ray chunk, like the reference reuses ``featmaps`` /
  ``src_cameras`` across the chunk loop (render_image.py:68-117)."""

  def __init__(self, query_camera, src_rgbs, src_cameras, featmaps, proj_matrices=None):
    """proj_matrices: optional [V,4,4] fp32 ``K . inv(c2w)`` of the source views formed by the caller (``Projector(matrix_mode='torch')``
    forms them exactly as the reference does, torch.inverse + bmm on the tensors' device, projection.py:42-47); default: formed in double
    by k_prepare_cameras and rounded once."""
    assert query_camera.shape[0] == 1 and src_rgbs.shape[0] == 1 and src_cameras.shape[0] == 1, \
        'only support batch_size=1 for now'  # projection.py:122-126
    dev = src_rgbs.device
    self.src_rgbs = _f32c(src_rgbs[0])          # [V,H,W,3] already channels-last in the reference
    self.cams = _f32c(src_cameras[0])           # [V,34]
    self.query = _f32c(query_camera[0])         # [34]
    self.V, self.H, self.W = self.src_rgbs.shape[:3]
    assert featmaps.shape[0] == self.V
    self.F, self.Hf, self.Wf = featmaps.shape[1:]
    cl = _channels_last_view(featmaps)
    if cl is not None:   # maps from the HIP encoder are channels-last in memory already: tapped in place
      self.feat_cl = cl
      st = stream_of(cl)
    else:                # maps from a PyTorch encoder (NCHW): one repack per target view
      fm = _f32c(featmaps)
      self.feat_cl = torch.empty((self.V, self.Hf, self.Wf, self.F), dtype=torch.float32, device=dev)
      st = stream_of(fm)
      call('dyn_nchw_to_nhwc', ptr(fm), ptr(self.feat_cl), self.V, self.F, self.Hf, self.Wf, st)
    self.proj = torch.empty((self.V, 16), dtype=torch.float32, device=dev)
    self.query_center = torch.empty((4,), dtype=torch.float32, device=dev)
    call('dyn_prepare_cameras', ptr(self.cams), self.V, ptr(self.query), ptr(self.proj), ptr(self.query_center), st)
    if proj_matrices is not None:
      P = _f32c(proj_matrices.to(dev))
      assert tuple(P.shape) == (self.V, 4, 4), 'proj_matrices must be [V,4,4]'
      self.proj[:, :12] = P[:, :3, :].reshape(self.V, 12)  # rows 0..2 of K . inv(c2w); the camera centres at [12..14] stay
    # normalize()/inbound() use h, w from train_cameras[0][:2] (projection.py:136); read once on the host
    hw = self.cams[0, :2].tolist()
    self.img_h, self.img_w = float(hw[0]), float(hw[1])


def sample_along_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform, t_rand=None, want_pts=True, want_s=True):
  """k_sample_along_ray.  depth_range: device tensor [1,2].  -> pts [R,S,3] | None, z_vals [R,S], s_vals [R,S] | None"""
  k = _Keep()
  ray_o, ray_d = _f32c(ray_o), _f32c(ray_d)
  R = ray_o.shape[0]
  dev = ray_o.device
  z = torch.empty((R, N_samples), dtype=torch.float32, device=dev)
  s = torch.empty_like(z) if want_s else None
  pts = torch.empty((R, N_samples, 3), dtype=torch.float32, device=dev) if want_pts else None
  dr = _f32c(depth_range.reshape(-1))
  p = params('DynSampleParams', R=R, S=N_samples, inv_uniform=int(bool(inv_uniform)), ray_o=ptr(ray_o), ray_d=ptr(ray_d),
             depth_range=ptr(dr), t_rand=k(t_rand), z_vals=ptr(z), s_vals=ptr(s), pts=ptr(pts))
  call('dyn_sample_along_ray', ctypes.byref(p), stream_of(ray_o))
  return pts, z, s


def points_from_z(ray_o, ray_d, z_vals, depth_range=None, want_pts=True):
  ray_o, ray_d, z_vals = _f32c(ray_o), _f32c(ray_d), _f32c(z_vals)
  R, S = z_vals.shape
  pts = torch.empty((R, S, 3), dtype=torch.float32, device=z_vals.device) if want_pts else None
  s = torch.empty_like(z_vals) if depth_range is not None else None
  dr = _f32c(depth_range.reshape(-1)) if depth_range is not None else None
  call('dyn_points_from_z', ptr(ray_o), ptr(ray_d), ptr(z_vals), ptr(dr), R, S, ptr(pts), ptr(s), stream_of(z_vals))
  return pts, s


def project_points(cams, xyz, xyz_st=None, query_camera=None, proj_matrices=None, want_pix=True):
  """k_project_points: Projector.compute_projections / compute_angle as stand-alone calls.  cams [V,34], xyz [V,...,3] (+ xyz_st [V,...,3] and
  query_camera [34] for the viewing-angle differences) -> (pix [V,...,2], in_front [V,...] bool) and / or ray_diff [V,...,4]."""
  cams, xyz = _f32c(cams), _f32c(xyz)
  V = cams.shape[0]
  assert xyz.shape[0] == V and xyz.shape[-1] == 3
  shp = tuple(xyz.shape[:-1])
  n = xyz[0].numel() // 3
  dev = xyz.device
  proj = torch.empty((V, 16), dtype=torch.float32, device=dev)
  qc = torch.zeros((4,), dtype=torch.float32, device=dev)
  q = _f32c(query_camera.reshape(-1)) if query_camera is not None else None
  st = stream_of(xyz)
  call('dyn_prepare_cameras', ptr(cams), V, ptr(q), ptr(proj), ptr(qc), st)
  if proj_matrices is not None:
    proj[:, :12] = _f32c(proj_matrices.to(dev))[:, :3, :].reshape(V, 12)
  pix = torch.empty(shp + (2,), dtype=torch.float32, device=dev) if want_pix else None
  front = torch.empty(shp, dtype=torch.float32, device=dev) if want_pix else None
  rd = xs = None
  if xyz_st is not None:
    xs = _f32c(xyz_st.expand(xyz.shape))
    rd = torch.empty(shp + (4,), dtype=torch.float32, device=dev)
  call('dyn_project_points', ptr(xyz), ptr(xs), ptr(proj), ptr(qc), V, n, ptr(pix), ptr(front), ptr(rd), st)
  return pix, (front > 0 if front is not None else None), rd


GATHER_STATS = None  # set to {} to tally calls and algorithmic bytes of project_gather (a measurement hook; never read by the product)


def project_gather(views: SourceViews, R, S, ray_o=None, ray_d=None, z_vals=None, pts_st=None, xyz=None, pix_mask_thresh=None, traj=None, want_ray_diff=True):
  """k_project_gather_tile -> rgb_feat [R,S,V,3+F], ray_diff [R,S,V,4], mask [R,S,V,1]; with pix_mask_thresh also the per-sample
  observation mask ``mask[..., 0].sum(dim=2) > thresh`` [R,S] (render_ray.py:736-741) as a fourth result, from the same launch.
  ``traj`` = (coeff [R,S,3B], basis [frames,B], rows int32 DEVICE [V], ref row): the fused form of compute_traj_pts -- view v sees
  pts_st + (traj(rows[v]) - traj(ref)), formed inside the kernel; excludes ``xyz`` (which is that array, materialised).
  ``want_ray_diff=False``: ray_diff is neither written nor returned (None in its place) -- the dynamic branch has no use for it."""
  k = _Keep()
  tj = {}
  if traj is not None:
    assert xyz is None and pts_st is not None, 'the fused trajectory form displaces pts_st itself'
    coeff, basis, rows_dev, ref = traj
    assert rows_dev.dtype == torch.int32 and rows_dev.numel() == views.V and rows_dev.device == views.proj.device
    k.held.append(rows_dev)
    tj = dict(traj_coeff=k(coeff), traj_basis=k(basis), traj_rows=rows_dev.data_ptr(), traj_B=int(basis.shape[1]), traj_ref=int(ref))
  dev = views.proj.device
  V, C = views.V, 3 + views.F
  rgb_feat = torch.empty((R, S, V, C), dtype=torch.float32, device=dev)
  ray_diff = torch.empty((R, S, V, 4), dtype=torch.float32, device=dev) if want_ray_diff else None
  mask = torch.empty((R, S, V, 1), dtype=torch.float32, device=dev)
  pix = torch.empty((R, S), dtype=torch.float32, device=dev) if pix_mask_thresh is not None else None
  p = params('DynProjectGatherParams', R=R, S=S, V=V, H=views.H, W=views.W, Hf=views.Hf, Wf=views.Wf, F=views.F,
             img_h=views.img_h, img_w=views.img_w, ray_o=k(ray_o), ray_d=k(ray_d), z_vals=k(z_vals),
             pts_st=k(pts_st), xyz=k(xyz), proj=ptr(views.proj), query_center=ptr(views.query_center),
             src_rgb=ptr(views.src_rgbs), feat_cl=ptr(views.feat_cl), rgb_feat=ptr(rgb_feat), ray_diff=ptr(ray_diff), mask=ptr(mask),
             pix_mask=ptr(pix), pix_mask_thresh=float(pix_mask_thresh) if pix_mask_thresh is not None else 0.0, **tj)
  call('dyn_project_gather', ctypes.byref(p), stream_of(rgb_feat))
  if GATHER_STATS is not None:  # opt-in tally of the launches' algorithmic bytes: bench.py prices the in-frame gather with it
    GATHER_STATS['calls'] = GATHER_STATS.get('calls', 0) + 1
    GATHER_STATS['bytes'] = GATHER_STATS.get('bytes', 0) + R * S * V * 160 + V * (views.Hf * views.Wf * views.F + views.H * views.W * 3) * 4 + R * (24 + 4 * S)
  if pix_mask_thresh is not None:
    return rgb_feat, ray_diff, mask, pix
  return rgb_feat, ray_diff, mask


def sample_mask(mask, thresh):
  """pixel_mask = mask[..., 0].sum(dim=2) > thresh  (render_ray.py:736-741) as 0/1 floats [R,S]."""
  R, S, V = mask.shape[:3]
  out = torch.empty((R, S), dtype=torch.float32, device=mask.device)
  call('dyn_sample_mask', ptr(mask), R * S, V, float(thresh), ptr(out), stream_of(mask))
  return out


def composite(raw_dy, z_vals, pix_mask_dy, raw_static=None, pix_mask_st=None, per_sample=True):
  """k_composite -> dict of tensors with the reference's key set (render_ray.py:202-211 / 316-328)."""
  k = _Keep()
  raw_dy, z_vals = _f32c(raw_dy), _f32c(z_vals)
  R, S = z_vals.shape
  dev = z_vals.device
  two = raw_static is not None
  new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
  rgb, depth, rmask, weights = new(R, 3), new(R), new(R), new(R, S)
  alpha = new(R, S) if per_sample else None
  rgb_st = new(R, 3) if two else None
  rgb_dy = new(R, 3) if two else None
  a_dy = new(R, S) if two and per_sample else None
  w_dy = new(R, S) if two and per_sample else None
  w_st = new(R, S) if two and per_sample else None
  p = params('DynCompositeParams', R=R, S=S, raw_dy=ptr(raw_dy), raw_static=k(raw_static), z_vals=ptr(z_vals),
             pix_mask_dy=k(pix_mask_dy), pix_mask_st=k(pix_mask_st), rgb=ptr(rgb), rgb_static=ptr(rgb_st),
             rgb_dy=ptr(rgb_dy), depth=ptr(depth), ray_mask=ptr(rmask), weights=ptr(weights), alpha=ptr(alpha),
             alpha_dy=ptr(a_dy), weights_dy=ptr(w_dy), weights_st=ptr(w_st))
  call('dyn_composite', ctypes.byref(p), stream_of(z_vals))
  return dict(rgb=rgb, rgb_static=rgb_st, rgb_dy=rgb_dy, depth=depth, mask=rmask, weights=weights, alpha=alpha,
              alpha_dy=a_dy, weights_dy=w_dy, weights_st=w_st, z_vals=z_vals)


def fine_samples(z_vals, weights, N_importance, inv_uniform, u=None, want_inds=False):
  """k_fine_samples -> z_all [R,S+N] sorted, z_samples [R,N], inds [R,N] int32 | None."""
  k = _Keep()
  z_vals, weights = _f32c(z_vals), _f32c(weights)
  R, S = z_vals.shape
  dev = z_vals.device
  z_out = torch.empty((R, S + N_importance), dtype=torch.float32, device=dev)
  z_s = torch.empty((R, N_importance), dtype=torch.float32, device=dev)
  inds = torch.empty((R, N_importance), dtype=torch.int32, device=dev) if want_inds else None
  p = params('DynFineSampleParams', R=R, S=S, N=N_importance, inv_uniform=int(bool(inv_uniform)), z_vals=ptr(z_vals),
             weights=ptr(weights), u=k(u), z_out=ptr(z_out), z_samples=ptr(z_s), inds=ptr(inds, torch.int32))
  call('dyn_fine_samples', ctypes.byref(p), stream_of(z_vals))
  return z_out, z_s, inds


# ----------------------------------------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------------------------------------
STATIC_TENSORS = (
    'ray_dir_fc.0.weight', 'ray_dir_fc.0.bias', 'ray_dir_fc.2.weight', 'ray_dir_fc.2.bias', 'ref_feature_fc.0.weight',
    'ref_feature_fc.0.bias', 'base_fc.0.weight', 'base_fc.0.bias', 'base_fc.2.weight', 'base_fc.2.bias', 'vis_fc.0.weight',
    'vis_fc.0.bias', 'vis_fc.2.weight', 'vis_fc.2.bias', 'vis_fc2.0.weight', 'vis_fc2.0.bias', 'vis_fc2.2.weight', 'vis_fc2.2.bias',
    'geometry_fc.0.weight', 'geometry_fc.0.bias', 'geometry_fc.2.weight', 'geometry_fc.2.bias', 'ray_attention.w_qs.weight',
    'ray_attention.w_ks.weight', 'ray_attention.w_vs.weight', 'ray_attention.fc.weight', 'ray_attention.layer_norm.weight',
    'ray_attention.layer_norm.bias', 'out_geometry_fc.0.weight', 'out_geometry_fc.0.bias', 'out_geometry_fc.2.weight',
    'out_geometry_fc.2.bias', 'rgb_fc.0.weight', 'rgb_fc.0.bias', 'rgb_fc.2.weight', 'rgb_fc.2.bias', 'rgb_fc.4.weight', 'rgb_fc.4.bias', 's')


def _host_f32(t):
  """state-dict entry (torch tensor on any device, or numpy array) -> contiguous fp32 numpy array on the host."""
  import numpy as np
  if isinstance(t, torch.Tensor):
    t = t.detach().to('cpu', torch.float32).contiguous().numpy()
  return np.ascontiguousarray(t, dtype=np.float32)


def _expected_shapes(kind, F=32, num_basis=6):
  """{state-dict key: shape} of the reference module the C packers are written for (they index with these strides and check nothing):
  DynibarStatic / DynibarDynamic with input_dir=True and in_feat_ch=F (mlp_network.py:319-421, :129-215), MotionMLP (:558-603)."""
  from . import synthetic
  table = {'static': synthetic.static_layer_table, 'dynamic': synthetic.dynamic_layer_table}[kind](F) if kind != 'motion' else \
      synthetic.motion_layer_table(num_basis)
  shapes = {}
  for name, nout, nin, has_bias in table:
    shapes[name + '.weight'] = (nout, nin)
    if has_bias:
      shapes[name + '.bias'] = (nout,)
  if kind != 'motion':
    shapes['ray_attention.layer_norm.weight'] = (128,)
    shapes['ray_attention.layer_norm.bias'] = (128,)
  if kind == 'static':
    shapes['s'] = ()
  return shapes


def _pack(fn_pack, fn_size, names, state_dict, F, kind, optional=()):
  """Validates every tensor against the reference module's shapes, then hands raw host pointers to the C packer.  Entries listed in
  ``optional`` may be absent (they are packed as zeros).  F: in_feat_ch (static / dynamic) or num_basis (motion), as the C packer takes it."""
  import numpy as np
  shapes = _expected_shapes(kind, num_basis=F) if kind == 'motion' else _expected_shapes(kind, F=F)
  arrs = []
  for n in names:
    if n not in state_dict:
      if n in optional:
        arrs.append(np.zeros(max(1, int(np.prod(shapes[n]))), dtype=np.float32))
        continue
      raise KeyError(f'{kind} network: state dict has no {n!r} (keys: {sorted(state_dict)[:6]}...)')
    a = _host_f32(state_dict[n])
    if tuple(a.shape) != tuple(shapes[n]) and not (shapes[n] == () and a.size == 1):  # ascontiguousarray promotes 0-d to (1,)
      hint = ''
      if n == 'rgb_fc.0.weight' and kind == 'static' and tuple(a.shape) == (32, 33):
        hint = ' (a DynibarStatic built with input_dir=False; the kernels implement input_dir=True, as every shipped config sets)'
      raise ValueError(f'{kind} network: {n} has shape {tuple(a.shape)}, the kernels are built for {tuple(shapes[n])}{hint}')
    arrs.append(a.reshape(-1))
  ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
  n = int(getattr(_lib.lib(), fn_size)())
  blob = np.zeros(n, dtype=np.float32)
  call(fn_pack, ptrs, int(F), ctypes.c_void_p(blob.ctypes.data), n)
  return torch.from_numpy(blob)


def _infer_F(state_dict, key='ray_dir_fc.2.weight'):
  """in_feat_ch of a DynibarStatic / DynibarDynamic state dict: ray_dir_fc.2 maps 256 -> in_feat_ch + 3."""
  w = state_dict.get(key)
  if w is None:
    raise KeyError(f'state dict has no {key!r}')
  return int(w.shape[0]) - 3


class StaticNet:
  """DynibarStatic (mlp_network.py:319-527) as packed MFMA operand tiles on one device.  ``state_dict``: the module's
  state dict (torch tensors or numpy arrays; a DataParallel 'module.' prefix is accepted)."""

  def __init__(self, state_dict, device, anti_alias_pooling=True, mask_rgb=False, F=None):
    sd = _strip_module(state_dict)
    F = _infer_F(sd) if F is None else F
    # the module only owns the pooling temperature `s` when it was built with anti_alias_pooling (mlp_network.py:330-331); every
    # monocular config ships anti_alias_pooling = 0 (configs/train_kid-running.txt:41), and the kernels never read it then
    if anti_alias_pooling and 's' not in sd:
      raise KeyError("DynibarStatic state dict has no 's' but anti_alias_pooling is on (mlp_network.py:330-331)")
    self.blob = _pack('dyn_static_net_pack', 'dyn_static_net_blob_floats', STATIC_TENSORS, sd, F, 'static',
                      optional=() if anti_alias_pooling else ('s',)).to(device)
    self.anti_alias_pooling, self.mask_rgb = int(bool(anti_alias_pooling)), int(bool(mask_rgb))
    self._ws = _Workspace()

  def workspace(self, R, S, V, device):
    need = int(_lib.lib().dyn_static_net_workspace_bytes(R, S, V))
    if need == 0:
      raise ValueError(f'dyn_static_net: unsupported shape R={R} S={S} V={V}')
    return self._ws.get(need, device), need

  def __call__(self, views: SourceViews, ray_o, ray_d, pts, rgb_feat, ray_diff, mask):
    """-> raw [R,S,4]  (k_static_ref_feat, k_static_views, k_static_points, k_static_blend)."""
    k = _Keep()
    R, S, V = rgb_feat.shape[:3]
    dev = rgb_feat.device
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=dev)
    ws, need = self.workspace(R, S, V, dev)
    p = params('DynStaticNetParams', R=R, S=S, V=V, anti_alias_pooling=self.anti_alias_pooling, mask_rgb=self.mask_rgb,
               blob=ptr(self.blob), ray_o=k(ray_o), ray_d=k(ray_d), pts=k(pts), rgb_feat=k(rgb_feat),
               ray_diff=k(ray_diff), mask=k(mask), centers=ptr(views.proj), raw=ptr(raw), workspace=ptr(ws),
               workspace_bytes=need)
    call('dyn_static_net', ctypes.byref(p), stream_of(raw))
    return raw


DYNAMIC_TENSORS = (
    'ray_dir_fc.0.weight', 'ray_dir_fc.0.bias', 'ray_dir_fc.2.weight', 'ray_dir_fc.2.bias', 'base_fc.0.weight', 'base_fc.0.bias',
    'base_fc.2.weight', 'base_fc.2.bias', 'vis_fc.0.weight', 'vis_fc.0.bias', 'vis_fc.2.weight', 'vis_fc.2.bias', 'vis_fc2.0.weight',
    'vis_fc2.0.bias', 'vis_fc2.2.weight', 'vis_fc2.2.bias', 'geometry_fc.0.weight', 'geometry_fc.0.bias', 'geometry_fc.2.weight',
    'geometry_fc.2.bias', 'ray_attention.w_qs.weight', 'ray_attention.w_ks.weight', 'ray_attention.w_vs.weight', 'ray_attention.fc.weight',
    'ray_attention.layer_norm.weight', 'ray_attention.layer_norm.bias', 'ref_pts_fc.0.weight', 'ref_pts_fc.0.bias', 'ref_pts_fc.2.weight',
    'ref_pts_fc.2.bias', 'out_geometry_fc.0.weight', 'out_geometry_fc.0.bias', 'out_geometry_fc.2.weight', 'out_geometry_fc.2.bias',
    'rgb_fc.0.weight', 'rgb_fc.0.bias', 'rgb_fc.2.weight', 'rgb_fc.2.bias', 'rgb_fc.4.weight', 'rgb_fc.4.bias')
MOTION_TENSORS = tuple(f'pts_linears.{i}.{n}' for i in range(8) for n in ('weight', 'bias')) + ('coeff_linear.weight', 'coeff_linear.bias')


def _strip_module(state_dict):
  return {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}


class _Workspace:
  """A network's scratch between its kernels, kept across calls -- ONE PER STREAM: two ray chunks in flight on two streams (render_image.CHUNK_STREAMS)
  must not share it, and a buffer that only ever serves one stream needs no cross-stream bookkeeping with the caching allocator.  Resident memory is
  therefore (streams in use) x (the largest workspace asked for): about 2 GiB per stream at R = 8192, S = 64, V = 8 (DYNIBAR_CHUNK_STREAMS);
  at most MAX_ENTRIES buffers are kept.  An evicted buffer whose kernels are still queued is safe: it was allocated and is freed on the stream that used it."""

  MAX_ENTRIES = 4  # the chunk loop's side streams + the caller's stream (render_image.CHUNK_STREAMS + 1 by default); the least recently used entry goes first

  def __init__(self):
    self.bufs = {}  # (device index, stream handle) -> buffer, in order of last use

  def get(self, need, device):
    if need == 0:
      raise ValueError('unsupported network shape (R, S, V)')
    device = torch.device(device)
    if device.type == 'cuda':
      index = device.index if device.index is not None else torch.cuda.current_device()  # 'cuda' and 'cuda:0' are one device
      device = torch.device('cuda', index)
      key = (index, torch.cuda.current_stream(device).cuda_stream)
    else:
      key = (-1, 0)
    buf = self.bufs.pop(key, None)
    if buf is None or buf.numel() * 4 < need:
      buf = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
    self.bufs[key] = buf  # (re-inserted: most recently used last)
    while len(self.bufs) > self.MAX_ENTRIES:  # transient streams must not grow the resident memory without bound
      self.bufs.pop(next(iter(self.bufs)))
    return buf


class DynamicNet:
  """DynibarDynamic (mlp_network.py:129-316) on one device.  ``shift`` as passed to the module's constructor."""

  def __init__(self, state_dict, device, shift=0.0, F=None):
    sd = _strip_module(state_dict)
    F = _infer_F(sd) if F is None else F
    self.blob = _pack('dyn_dynamic_net_pack', 'dyn_dynamic_net_blob_floats', DYNAMIC_TENSORS, sd, F, 'dynamic').to(device)
    self.shift = float(shift)
    self._ws = _Workspace()

  def __call__(self, ray_d, pts, rgb_feat, mask, time):
    """time: device tensor [1] (the reference time embedding) -> raw [R,S,4]  (k_dynamic_time_feat, k_dynamic_views, k_dynamic_points)."""
    k = _Keep()
    R, S, V = rgb_feat.shape[:3]
    dev = rgb_feat.device
    raw = torch.empty((R, S, 4), dtype=torch.float32, device=dev)
    need = int(_lib.lib().dyn_dynamic_net_workspace_bytes(R, S, V))
    ws = self._ws.get(need, dev)
    p = params('DynDynamicNetParams', R=R, S=S, V=V, shift=self.shift, blob=ptr(self.blob), ray_d=k(ray_d), pts=k(pts), rgb_feat=k(rgb_feat),
               mask=k(mask), time=k(time.reshape(-1)), raw=ptr(raw), workspace=ptr(ws), workspace_bytes=need)
    call('dyn_dynamic_net', ctypes.byref(p), stream_of(raw))
    return raw


class MotionMLP:
  """MotionMLP (mlp_network.py:558-618) on one device."""

  def __init__(self, state_dict, device, num_basis=6, sf_mag_div=1.0):
    sd = _strip_module(state_dict)
    cw = sd.get('coeff_linear.weight')
    if cw is not None and int(cw.shape[0]) != 3 * int(num_basis):
      raise ValueError(f'MotionMLP: coeff_linear has {int(cw.shape[0])} outputs but the trajectory basis has {num_basis} columns '
                       f'(expected {3 * int(num_basis)}; mlp_network.py:598-603)')
    self.blob = _pack('dyn_motion_mlp_pack', 'dyn_motion_mlp_blob_floats', MOTION_TENSORS, sd, num_basis, 'motion').to(device)
    self.num_basis, self.sf_mag_div = int(num_basis), float(sf_mag_div)

  def __call__(self, pts, time, n_zero_last):
    """pts [R,S,3], time device [1] -> raw coefficients [R,S,3B] with the last n_zero_last samples of every ray zeroed (k_motion_mlp)."""
    k = _Keep()
    R, S = pts.shape[:2]
    coeff = torch.empty((R, S, 3 * self.num_basis), dtype=torch.float32, device=pts.device)
    call('dyn_motion_mlp', ptr(self.blob), k(pts), k(time.reshape(-1)), R, S, self.num_basis, int(n_zero_last), self.sf_mag_div, ptr(coeff),
         stream_of(coeff))
    return coeff


def trajectory_points(coeff, basis, pts, rows, row_ref):
  """k_trajectory_points: pts_seq [len(rows), R, S, 3]; rows = basis row per source view (frame + offset), < 0 = undisplaced."""
  k = _Keep()
  R, S = pts.shape[:2]
  B = basis.shape[1]
  out = torch.empty((len(rows), R, S, 3), dtype=torch.float32, device=pts.device)
  arr = (ctypes.c_int * len(rows))(*[int(r) for r in rows])
  call('dyn_trajectory_points', k(coeff), k(basis), k(pts), R * S, B, arr, len(rows), int(row_ref), ptr(out), stream_of(out))
  return out


# ----------------------------------------------------------------------------------------------------------------------
# feature encoder
# ----------------------------------------------------------------------------------------------------------------------
class Encoder:
  """The executed part of the reference's ResNet (feature_network.py:179-311) as packed convolution images on one device."""

  def __init__(self, state_dict, device):
    import numpy as np
    from . import synthetic
    sd = _strip_module(state_dict)
    arrs = []
    for name, shape in synthetic.ENCODER_TENSORS:
      if name not in sd:
        raise KeyError(f'encoder: state dict has no {name!r}')
      a = _host_f32(sd[name])
      if tuple(a.shape) != tuple(shape):
        raise ValueError(f'encoder: {name} has shape {tuple(a.shape)}, the kernels are built for {tuple(shape)} '
                         "(resnet34-style BasicBlock encoder with coarse_out_ch + fine_out_ch = 64)")
      arrs.append(a.reshape(-1))
    ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    n = int(_lib.lib().dyn_encoder_blob_floats())
    blob = np.zeros(n, dtype=np.float32)
    call('dyn_encoder_pack', ptrs, ctypes.c_void_p(blob.ctypes.data), n)
    self.blob = torch.from_numpy(blob).to(device)
    self._ws = _Workspace()

  def __call__(self, images):
    """images [N,H,W,3] channels-last fp32 on the device -> (coarse [N,Hf,Wf,32], fine [N,Hf,Wf,32]) channels-last."""
    images = _f32c(images)
    N, H, W, C = images.shape
    assert C == 3
    dev = images.device
    hf, wf = ctypes.c_int(0), ctypes.c_int(0)
    call('dyn_encoder_out_size', H, W, ctypes.byref(hf), ctypes.byref(wf))
    coarse = torch.empty((N, hf.value, wf.value, 32), dtype=torch.float32, device=dev)
    fine = torch.empty_like(coarse)
    need = int(_lib.lib().dyn_encoder_workspace_bytes(N, H, W))
    ws = self._ws.get(need, dev)
    p = params('DynEncoderParams', N=N, H=H, W=W, blob=ptr(self.blob), images=ptr(images), coarse=ptr(coarse), fine=ptr(fine),
               workspace=ptr(ws), workspace_bytes=need)
    call('dyn_encoder_forward', ctypes.byref(p), stream_of(images))
    return coarse, fine
