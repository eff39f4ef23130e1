"""Training, third slice: the motion path of train.py:283-467 on the HIP kernels -- every place where a
gradient reaches the sample *locations* and, through them, MotionMLP and the trajectory basis:

    MotionMLPFunction      coeff = MotionMLP([pts, t]) with the last samples zeroed (mlp_network.py:558-618, render_ray.py:955-961)
    TrajectoryFunction     pts_seq[v] = pts + coeff . (basis[row_v] - basis[ref])      (render_ray.py:361-369, :965-985, :1109-1168)
    GatherFunction         rgb_feat = Projector.compute_with_motions(pts, pts_seq, ...)  (projection.py:103-176): d feature maps AND d points
    RenderFlowsFunction    compute_optical_flow                                          (render_ray.py:333-358)

as ``torch.autograd.Function``s whose forward and backward are HIP kernels (``dyn_train_gemm`` with ReLU for the MLP, ``dyn_train_embed*``,
``dyn_trajectory_bwd``, ``dyn_gather_bwd`` / ``dyn_gather_bwd_pts``, ``dyn_render_flows_bwd``).  Autograd only chains them.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import ops
from ._lib import call, stream_of
from .train_static import NONE, _Lin, _act_bwd, _p, zero_grads

RELU = 2
MOTION_FREQS = np.linspace(1.0, 17.0, 16).astype(np.float32)   # PeriodicEmbed(max_freq=16, N_freq=16, linspace=True): mlp_network.py:589
OCTAVES5 = (2.0 ** np.arange(5)).astype(np.float32)


def _freqs(a):
  arr = (ctypes.c_float * len(a))(*[float(x) for x in a])
  return arr, ctypes.cast(arr, ctypes.c_void_p)


def _f32(t):
  return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _rows_ld(g, width):
  """(tensor, ld) of a gradient whose last dim is `width` laid out as rows of a wider matrix (a column slice view) or contiguous"""
  if g.is_contiguous():
    return g, width
  st = g.stride()
  shape = g.shape
  if g.dtype == torch.float32 and st[-1] == 1:
    ld = st[-2]
    ok = all(st[i] == st[i + 1] * shape[i + 1] for i in range(len(shape) - 2))
    if ok and ld >= width:
      return g, ld
  return g.contiguous(), width


# ----------------------------------------------------------------------------------------------------------------------
class GatherFunction(torch.autograd.Function):
  """rgb_feat [R,S,V,3+F] (differentiable w.r.t. the feature maps and the per-view points), ray_diff, mask, sample mask (forward values)"""

  @staticmethod
  def forward(ctx, featmaps, xyz, meta):
    views, R, S, kw, thresh = meta
    out = ops.project_gather(views, R, S, xyz=xyz.detach() if xyz is not None else None, pix_mask_thresh=thresh, **kw)
    rgb_feat, ray_diff, mask, pm = out
    ctx.views, ctx.R, ctx.S = views, R, S
    ctx.xyz = _f32(xyz.detach()) if xyz is not None else None
    if kw.get('pts_st') is not None:
      ctx.pts = _f32(kw['pts_st'])
    else:
      ctx.pts = ops.points_from_z(kw['ray_o'], kw['ray_d'], kw['z_vals'])[0]
    ctx.mark_non_differentiable(ray_diff, mask, pm)
    return rgb_feat, ray_diff, mask, pm

  @staticmethod
  def backward(ctx, g, _rd, _m, _pm):
    v, R, S = ctx.views, ctx.R, ctx.S
    g, ld = _rows_ld(g.float() if g.dtype != torch.float32 else g, 3 + v.F)
    st = stream_of(g)
    gf = gx = None
    if ctx.needs_input_grad[0]:
      dfeat = torch.zeros((v.V, v.Hf, v.Wf, v.F), dtype=torch.float32, device=g.device)
      call('dyn_gather_bwd', _p(ctx.pts), _p(ctx.xyz) if ctx.xyz is not None else None, _p(v.proj), R, S, v.V, v.Hf, v.Wf, v.F, v.img_h, v.img_w,
           _p_any(g), ld, 3, _p(dfeat), st)
      gf = dfeat.permute(0, 3, 1, 2)
    if ctx.xyz is not None and ctx.needs_input_grad[1]:
      gx = torch.empty_like(ctx.xyz)
      call('dyn_gather_bwd_pts', _p(ctx.pts), _p(ctx.xyz), _p(v.proj), _p(v.src_rgbs), _p(v.feat_cl), R, S, v.V, v.H, v.W, v.Hf, v.Wf, v.F, v.img_h,
           v.img_w, _p_any(g), ld, _p(gx), st)
    return gf, gx, None


def _p_any(t):
  """pointer to the first element of a (possibly column-sliced) fp32 tensor"""
  assert t.dtype == torch.float32
  return ctypes.c_void_p(t.data_ptr())


def gather(views, featmaps, R, S, xyz=None, pix_mask_thresh=1.0, **kw):
  """ops.project_gather with a graph: kw = ray_o / ray_d / z_vals or pts_st (sample points), xyz = per-view displaced points or None"""
  return GatherFunction.apply(featmaps, xyz, (views, R, S, kw, pix_mask_thresh))


# ----------------------------------------------------------------------------------------------------------------------
class TrajectoryFunction(torch.autograd.Function):
  @staticmethod
  def forward(ctx, coeff, basis, pts, meta):
    rows, row_ref = meta
    out = ops.trajectory_points(coeff.detach(), basis.detach(), pts.detach(), rows, row_ref)
    ctx.rows, ctx.row_ref = [int(r) for r in rows], int(row_ref)
    ctx.save_for_backward(_f32(coeff.detach()), _f32(basis.detach()))
    ctx.pshape = tuple(pts.shape)
    return out

  @staticmethod
  def backward(ctx, g):
    coeff, basis = ctx.saved_tensors
    g = _f32(g)
    n_pts = coeff.numel() // coeff.shape[-1]
    B = basis.shape[1]
    dcoeff = torch.empty_like(coeff)
    dbasis = torch.zeros_like(basis)
    dpts = torch.empty(ctx.pshape, dtype=torch.float32, device=g.device)
    arr = (ctypes.c_int * len(ctx.rows))(*ctx.rows)
    call('dyn_trajectory_bwd', _p(g), _p(coeff), _p(basis), n_pts, B, arr, len(ctx.rows), ctx.row_ref, _p(dcoeff), _p(dbasis), _p(dpts), stream_of(g))
    return dcoeff, dbasis, dpts, None


def trajectory_points(coeff, basis, pts, rows, row_ref):
  return TrajectoryFunction.apply(coeff, basis, pts, (list(rows), row_ref))


# ----------------------------------------------------------------------------------------------------------------------
class RenderFlowsFunction(torch.autograd.Function):
  @staticmethod
  def forward(ctx, weights, pts_seq, meta):
    proj, uv = meta
    V, R, S = pts_seq.shape[:3]
    w, q = _f32(weights.detach()), _f32(pts_seq.detach())
    flows = torch.empty((V, R, 2), dtype=torch.float32, device=w.device)
    call('dyn_render_flows', _p(w), _p(q), _p(proj), _p(_f32(uv)), R, S, V, _p(flows), stream_of(w))
    ctx.save_for_backward(w, q, proj)
    return flows

  @staticmethod
  def backward(ctx, g):
    w, q, proj = ctx.saved_tensors
    V, R, S = q.shape[:3]
    dw, dq = torch.empty_like(w), torch.empty_like(q)
    call('dyn_render_flows_bwd', _p(_f32(g)), _p(w), _p(q), _p(proj), R, S, V, _p(dw), _p(dq), stream_of(w))
    return dw, dq, None


def render_flows(weights, pts_seq, proj, uv):
  return RenderFlowsFunction.apply(weights, pts_seq, (proj, uv))


# ----------------------------------------------------------------------------------------------------------------------
MOTION_NAMES = ops.MOTION_TENSORS


def _motion_params(net):
  net = net.module if hasattr(net, 'module') and not isinstance(net, dict) else net
  sd = dict(net.named_parameters()) if hasattr(net, 'named_parameters') else dict(net)
  sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
  missing = [n for n in MOTION_NAMES if n not in sd]
  if missing:
    raise KeyError(f'MotionMLP parameters missing: {missing[:4]}')
  return list(MOTION_NAMES), [sd[n] for n in MOTION_NAMES]


class MotionMLPFunction(torch.autograd.Function):
  """coeff [R,S,3B] = MotionMLP(PE([pts, t])) / sf_mag_div with the last n_zero samples of every ray zeroed; gradients to the 18
  parameters and to pts (the anchor pass evaluates the MLP at points that themselves depend on the coefficients)."""

  @staticmethod
  def forward(ctx, pts, meta, *param_tensors):
    names, time, n_zero, sf_div = meta
    w = {n: _f32(t.detach()) for n, t in zip(names, param_tensors)}
    R, S = pts.shape[:2]
    P = R * S
    dev = pts.device
    st = stream_of(pts)
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    xyzt = torch.cat([_f32(pts.detach()).reshape(P, 3), time.reshape(-1)[:1].to(dev).float().expand(P, 1)], dim=1).contiguous()  # input assembly
    fa, fp = _freqs(MOTION_FREQS)
    X0 = new(P, 132)
    call('dyn_train_embed', _p(xyzt), 4, P, 4, fp, 16, _p(X0), 132, st)
    L, H = [], []
    x = X0
    for i in range(8):
      W, b = w[f'pts_linears.{i}.weight'], w[f'pts_linears.{i}.bias']
      h = new(P, 256)
      if i == 5:   # the layer after the skip: its input is [PE | h] (mlp_network.py:611-613)
        la, lb = _Lin(W, None, 0, 132), _Lin(W, b, 132, 256)
        T = new(P, 256)
        la.fwd(st, X0, 0, 132, T, 0, 256, P)
        lb.fwd(st, x, 0, 256, h, 0, 256, P, RELU, addend=T, ld_add=256, add_div=1)
        L.append((la, lb))
      else:
        l = _Lin(W, b)
        l.fwd(st, x, 0, 132 if i == 0 else 256, h, 0, 256, P, RELU)
        L.append(l)
      H.append(h)
      x = h
    lc = _Lin(w['coeff_linear.weight'], w['coeff_linear.bias'])
    C = lc.n_out
    coeff = new(R, S, C)
    lc.fwd(st, x, 0, 256, coeff, 0, C, P)
    call('dyn_train_zero_tail', _p(coeff), R, S, C, int(n_zero), 1.0 / float(sf_div), st)
    ctx.names, ctx.w, ctx.L, ctx.H, ctx.lc, ctx.X0, ctx.xyzt, ctx.fa = names, w, L, H, lc, X0, xyzt, fa
    ctx.dims, ctx.n_zero, ctx.sf_div = (R, S, P, C), int(n_zero), float(sf_div)
    return coeff

  @staticmethod
  def backward(ctx, g):
    if ctx.H is None:
      raise RuntimeError('MotionMLPFunction: the saved activations were released by the first backward pass; call the renderer again instead of backward(retain_graph=True)')
    R, S, P, C = ctx.dims
    w, L, H, X0 = ctx.w, ctx.L, ctx.H, ctx.X0
    dev = g.device
    st = stream_of(g)
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    grads = zero_grads(w)
    dC = g.float().contiguous().clone()
    call('dyn_train_zero_tail', _p(dC), R, S, C, ctx.n_zero, 1.0 / ctx.sf_div, st)
    _act_bwd(st, dC, 0, C, None, 0, C, P, C, NONE, grads['coeff_linear.bias'])
    dH = new(P, 256)
    dX0 = torch.zeros((P, 132), dtype=torch.float32, device=dev)
    # dH arrives already multiplied by ReLU'(H[i]) and -- when its rows are 16-byte aligned -- with the bias gradient of layer i and its scale
    # taken from the producing GEMM's tiles (summed); otherwise the read-only pass below does that
    summed = ctx.lc.bwd(st, dC, 0, C, H[7], 0, 256, grads['coeff_linear.weight'], P, dH, 0, 256, act_y=(H[7], 0, 256, RELU),
                        dbias=grads['pts_linears.7.bias'])
    for i in range(7, -1, -1):
      if not summed:
        _act_bwd(st, dH, 0, 256, None, 0, 256, P, 256, NONE, grads[f'pts_linears.{i}.bias'])
      gw = grads[f'pts_linears.{i}.weight']
      if i == 5:
        la, lb = L[5]
        dprev = new(P, 256)
        summed = lb.bwd(st, dH, 0, 256, H[4], 0, 256, gw, P, dprev, 0, 256, act_y=(H[4], 0, 256, RELU), dbias=grads['pts_linears.4.bias'])
        la.bwd(st, dH, 0, 256, X0, 0, 132, gw, P, dX0, 0, 132, acc_dx=1)
        dH = dprev
      elif i == 0:
        L[0].bwd(st, dH, 0, 256, X0, 0, 132, gw, P, dX0, 0, 132, acc_dx=1)
      else:
        dprev = new(P, 256)
        summed = L[i].bwd(st, dH, 0, 256, H[i - 1], 0, 256, gw, P, dprev, 0, 256, act_y=(H[i - 1], 0, 256, RELU),
                          dbias=grads[f'pts_linears.{i - 1}.bias'])
        dH = dprev
    gp = None
    if ctx.needs_input_grad[0]:
      dxyzt = new(P, 4)
      _, fp = ctx.fa, ctypes.cast(ctx.fa, ctypes.c_void_p)
      call('dyn_train_embed_bwd', _p(ctx.xyzt), 4, P, 4, fp, 16, _p(dX0), 132, _p(dxyzt), 4, 0, st)
      gp = dxyzt[:, :3].reshape(R, S, 3)
    out = tuple(grads[n] if ctx.needs_input_grad[2 + i] else None for i, n in enumerate(ctx.names))
    ctx.H = ctx.L = ctx.X0 = None
    return (gp, None) + out


def motion_coeff(net, pts, time, n_zero_last, sf_mag_div=1.0):
  names, tensors = _motion_params(net)
  return MotionMLPFunction.apply(pts, (names, time, int(n_zero_last), float(sf_mag_div)), *tensors)


def wants_grad(net, basis=None):
  if not torch.is_grad_enabled():
    return False
  if isinstance(basis, torch.Tensor) and basis.requires_grad:
    return True
  net = net.module if hasattr(net, 'module') and not isinstance(net, dict) else net
  ps = net.parameters() if hasattr(net, 'parameters') else [v for v in net.values() if isinstance(v, torch.Tensor)]
  return any(p.requires_grad for p in ps)
