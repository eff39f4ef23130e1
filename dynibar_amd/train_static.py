"""Training of the static branch on the HIP kernels (first slice): the graph the reference's static
bootstrap stage differentiates (train.py:116-199) --

    loss(ret['outputs_coarse_st']['rgb'])  ->  raw2outputs_vanilla (render_ray.py:134-201)
                                           ->  DynibarStatic.forward (mlp_network.py:423-527)
                                           ->  Projector.compute_with_motions' F.grid_sample (projection.py:160-167)
                                           ->  DynibarStatic's parameters and the static feature maps (feature_net_st's output)

as two ``torch.autograd.Function``s whose forward AND backward are sequences of the ``dyn_train_*`` kernels
(csrc/dyn_train.hip): a tiled split-half-float MFMA GEMM for every Linear (forward, data gradient, weight gradient) and small
row / per-point kernels for what sits between them.  PyTorch's role is the one it has in the rest of the package: device memory,
the stream, and the autograd *graph* that carries the gradients on into whichever encoder produced the feature maps and into the
optimizer; no ATen kernel computes any of the values or gradients of this graph.

Activations are row-major fp32 matrices kept in HBM between the kernels of a step: N = R*S*V rows (row = point*V + view) or
P = R*S rows; the buffers and their widths are listed in ``_forward``.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib, ops
from ._lib import call, params, stream_of

ELU, NONE = 1, 0

# Memory for time: with RECOMPUTE_HIDDEN the 256-wide hidden layers of ray_dir_fc / base_fc (1 KB per row each) are not kept from the forward
# pass but recomputed from their narrow inputs right before the backward pass needs them (the same launches: bit-identical results).
# Measured at 3072 rays (MI355X): peak 46.4 -> 37.0 GB, iteration 107.2 -> 112.6 ms; off by default (a 288 GB device has the room).
RECOMPUTE_HIDDEN = os.environ.get('DYNIBAR_TRAIN_RECOMPUTE', '0') == '1'
# Tests set this: buffers a kernel is documented to write completely (the GEMM's partial column sums) start as NaN instead of uninitialised
# memory, so an entry a launch skipped fails a gradient check every time rather than by luck.
POISON_SCRATCH = os.environ.get('DYNIBAR_TRAIN_POISON', '0') == '1'


def _p(t, off=0):
  """device pointer of element `off` of a contiguous fp32 tensor"""
  _lib.ptr(t)  # dtype / contiguity / device checks
  return ctypes.c_void_p(t.data_ptr() + 4 * int(off))


class _Scalars:
  """Zero-initialised device scalars handed out from 512-element blocks (one fill kernel per block instead of one per scalar); an
  element is handed out once, so it is still zero when the kernel that accumulates into it runs."""
  blocks = {}

  @classmethod
  def take(cls, device):
    blk = cls.blocks.get(device)
    if blk is None or blk[1] >= blk[0].numel():
      blk = cls.blocks[device] = [torch.zeros(512, dtype=torch.float32, device=device), 0]
    blk[1] += 1
    return blk[0][blk[1] - 1:blk[1]]


def zero_grads(w):
  """{name: zero tensor shaped like w[name]} as 256-byte aligned views of ONE zero-filled buffer (a step has ~40 parameter tensors)."""
  offs, total = {}, 0
  for n, t in w.items():
    offs[n] = total
    total += (t.numel() + 63) // 64 * 64
  dev = next(iter(w.values())).device
  flat = torch.zeros(total, dtype=torch.float32, device=dev)
  return {n: flat[offs[n]:offs[n] + t.numel()].view(t.shape) for n, t in w.items()}


GEMM_STATS = None  # bench.py sets this to {'bytes': 0, 'flops': 0, 'calls': 0} to total the algorithmic traffic / work of the GEMM launches


def _gemm(st, A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, M, N, K, bias=None, addend=None, ld_add=0, add_div=1, act=NONE, accumulate=0, k_split=1,
          a_absmax=None, act_y=None, ld_y=0, act_y_kind=0, colsum_part=None, ld_part=0, amax_part=None, rowscale=None, kscale=None):
  if GEMM_STATS is not None:  # algorithmic: every operand element read once, every result element written once
    GEMM_STATS['bytes'] += 4 * (M * K + N * K + M * N + (M * N if act_y is not None else 0))
    GEMM_STATS['flops'] += 2 * M * N * K
    GEMM_STATS['calls'] += 1
  p = params('DynTrainGemmParams', A=A, a_rs=a_rs, a_ks=a_ks, B=B, b_rs=b_rs, b_ks=b_ks, C=C, ldc=ldc, M=M, N=N, K=K, bias=bias, addend=addend,
             ld_add=ld_add, add_div=add_div, act=act, accumulate=accumulate, k_split=k_split, a_absmax=a_absmax, act_y=act_y, ld_y=ld_y,
             act_y_kind=act_y_kind, colsum_part=colsum_part, ld_part=ld_part, amax_part=amax_part, rowscale=rowscale, kscale=kscale)
  call('dyn_train_gemm', ctypes.byref(p), st)


def _untag(t):
  """Drop the cached largest-magnitude tag of a gradient tensor that a kernel is about to ACCUMULATE into in place: the tag (set by
  whatever produced the tensor) is the power-of-two scale of the split-half backward GEMMs, and a stale, too-small scale overflows
  their f16 operands.  The next consumer re-measures (dyn_train_absmax)."""
  if t is not None:
    t.__dict__.pop('_dyn_absmax', None)


class _Lin:
  """Columns [col0, col0 + K) of an nn.Linear weight [n_out, k_full] (+ its bias): one GEMM operand."""

  def __init__(self, W, bias=None, col0=0, K=None):
    self.W, self.bias, self.col0 = W, bias, col0
    self.n_out, self.k_full = W.shape
    self.K = self.k_full - col0 if K is None else K
    # The operand of the forward and data-gradient products: the weight slice itself when its rows are 16-byte aligned, else a copy with
    # the row stride padded to a multiple of four floats (made once per step; the GEMM's ring form takes aligned quads only, and an odd
    # stride -- ray_dir_fc.0: 103, base_fc.0: 210, rgb_fc.0: 261 ... -- would leave these layers on the slower tile form).
    if self.k_full % 4 == 0 and col0 % 4 == 0 and W.data_ptr() % 16 == 0:
      self.Wop, self.op_off, self.op_ld = W, col0, self.k_full
    else:
      k4 = (self.K + 3) // 4 * 4
      self.Wop = torch.zeros((self.n_out, k4), dtype=torch.float32, device=W.device)
      self.Wop[:, :self.K].copy_(W[:, col0:col0 + self.K])
      self.op_off, self.op_ld = 0, k4

  def fwd(self, st, X, x_off, ldx, Y, y_off, ldy, M, act=NONE, addend=None, ld_add=0, add_div=1, bias=True, rowscale=None):
    """Y = act((X * rowscale[:, None]) W^T + b + addend[row // add_div]); the row scale is applied to the product, X * rowscale is not formed."""
    if (self.n_out == 1 and act == NONE and addend is None and rowscale is None and self.col0 == 0 and self.K % 4 == 0 and self.K <= 256 and
        (self.K // 4) & (self.K // 4 - 1) == 0 and ldx % 4 == 0 and x_off % 4 == 0 and self.W.data_ptr() % 16 == 0):
      # one output: a dot product per row -- a row kernel, not a GEMM tile with one useful column
      call('dyn_train_rowdot', _p(X, x_off), ldx, _p(self.W), _p(self.bias) if (bias and self.bias is not None) else None, M, self.K,
           _p(Y, y_off), ldy, st)
      return
    _gemm(st, _p(X, x_off), ldx, 1, _p(self.Wop, self.op_off), self.op_ld, 1, _p(Y, y_off), ldy, M, self.n_out, self.K,
          bias=_p(self.bias) if (bias and self.bias is not None) else None, addend=_p(addend) if addend is not None else None,
          ld_add=ld_add, add_div=add_div, act=act, rowscale=_p(rowscale) if rowscale is not None else None)

  def bwd(self, st, dZ, dz_off, ld_dz, X, x_off, ldx, dW, M, dX=None, dx_off=0, ld_dx=0, acc_dx=0, act_y=None, dbias=None, x_scale=None):
    # act_y = (Y, y_off, ld_y, kind): X is the output Y of an ELU / ReLU layer and dX comes out already multiplied by act'(Y).
    # dbias: the bias gradient of THAT layer (column sums of dX) and the scale of dX are taken from the result tiles on their way out;
    # returns True when that happened (16-byte-aligned dX rows), False when the caller still has to run _act_bwd for them.
    """dW[:, col0:col0+K] += dZ^T X (split over the rows, atomics); dX (=|+=) dZ W[:, col0:col0+K].  dZ is the GEMMs' scaled operand:
    its largest magnitude comes from the activation-derivative pass that made it (_act_bwd leaves it on the tensor) or is measured here."""
    one_out = (dX is not None and self.n_out == 1 and acc_dx == 0 and dx_off == 0 and self.col0 == 0 and self.K % 4 == 0 and self.K <= 256 and
               (self.K // 4) & (self.K // 4 - 1) == 0 and ld_dx % 4 == 0 and self.W.data_ptr() % 16 == 0 and
               (act_y is None or (act_y[1] == 0 and act_y[2] % 4 == 0)))
    # one output whose input is the saved activation the data gradient goes back through: the weight gradient is a weighted column sum
    # of the rows that pass already reads (as a GEMM it was a 128-row tile with one useful row)
    fused_w = (one_out and act_y is not None and x_scale is None and self.k_full == self.K and x_off == 0 and act_y[2] == ldx and
               act_y[0].data_ptr() == X.data_ptr())
    if not fused_w:
      tag = getattr(dZ, '_dyn_absmax', None)
      if tag is not None and tag[0] == (dz_off, ld_dz, M, self.n_out):
        am = tag[1]
      else:
        am = _Scalars.take(dZ.device)
        call('dyn_train_absmax', _p(dZ, dz_off), M, self.n_out, ld_dz, _p(am), st)
        dZ._dyn_absmax = ((dz_off, ld_dz, M, self.n_out), am)
      ks = max(1, min(1024, M // 1024))  # reduction chunks of the weight gradient: enough (row tile, chunk) units for every resident workgroup
      # x_scale [M]: the layer ran on X * x_scale[:, None] (fwd's rowscale); its weight gradient takes the scale on the reduction index
      _gemm(st, _p(dZ, dz_off), 1, ld_dz, _p(X, x_off), 1, ldx, _p(dW, self.col0), self.k_full, self.n_out, self.K, M, accumulate=2, k_split=ks,
            a_absmax=_p(am), kscale=_p(x_scale) if x_scale is not None else None)
    if one_out:
      # one output: the data gradient is a rank-one product -- a row kernel, not a GEMM
      am2 = _Scalars.take(dX.device)
      call('dyn_train_outer_act_bwd', _p(dZ, dz_off), ld_dz, _p(self.W), _p(act_y[0]) if act_y is not None else None,
           act_y[2] if act_y is not None else 0, M, self.K, act_y[3] if act_y is not None else NONE, _p(dX), ld_dx,
           _p(dbias) if dbias is not None else None, _p(am2), _p(dW) if fused_w else None, st)
      dX._dyn_absmax = ((0, ld_dx, M, self.K), am2)
      return dbias is not None
    if dX is not None:
      fy = {} if act_y is None else dict(act_y=_p(act_y[0], act_y[1]), ld_y=act_y[2], act_y_kind=act_y[3])
      sums = dbias is not None and acc_dx == 0 and self.K % 4 == 0 and ld_dx % 4 == 0 and (dX.data_ptr() + 4 * dx_off) % 16 == 0
      if sums:
        tiles, ctiles = (M + 127) // 128, (self.K + 127) // 128
        part = torch.empty((tiles, self.K), dtype=torch.float32, device=dX.device)  # every (128-row tile, column) entry is written by the product
        apart = torch.empty(tiles * ctiles, dtype=torch.float32, device=dX.device)
        if POISON_SCRATCH:
          part.fill_(float('nan')); apart.fill_(float('inf'))  # (the reduction takes fmaxf, which drops a NaN)
        fy.update(colsum_part=_p(part), ld_part=self.K, amax_part=_p(apart))
      if acc_dx != 0:
        _untag(dX)
      # a slice whose width is not a multiple of four (rgb_fc.0: 133, base_fc.0: 70 / 35 ...) is computed over the padded operand copy's
      # full width: the extra columns are exact zeros landing in dX's padding columns, and the product gets 16-byte result rows (the
      # epilogue's fast form) instead of 4-byte stores
      Nd = self.K
      if (self.K % 4 != 0 and self.Wop is not self.W and acc_dx == 0 and act_y is None and ld_dx % 4 == 0 and dx_off % 4 == 0 and
          dx_off + self.op_ld <= ld_dx):  # columns [K, op_ld) of the slice must be dX's own padding, never a neighbour's live columns
        Nd = self.op_ld
      if acc_dx == 1 and act_y is None and self.K % 4 == 0 and ld_dx % 4 == 0 and (dX.data_ptr() + 4 * dx_off) % 16 == 0:
        # dX += dZ W as "dX = dZ W + addend" with the addend dX itself: every element is read and written by the one thread that owns it, and the
        # product keeps the epilogue's 16-byte form (the accumulate flag takes the 4-byte atomic form)
        _gemm(st, _p(dZ, dz_off), ld_dz, 1, _p(self.Wop, self.op_off), 1, self.op_ld, _p(dX, dx_off), ld_dx, M, Nd, self.n_out, accumulate=0,
              a_absmax=_p(am), addend=_p(dX, dx_off), ld_add=ld_dx, add_div=1)
      else:
        _gemm(st, _p(dZ, dz_off), ld_dz, 1, _p(self.Wop, self.op_off), 1, self.op_ld, _p(dX, dx_off), ld_dx, M, Nd, self.n_out, accumulate=acc_dx,
              a_absmax=_p(am), **fy)
      if sums:
        am2 = _Scalars.take(dX.device)
        call('dyn_train_colsum_reduce', _p(part), tiles, self.K, self.K, _p(dbias), _p(apart), tiles * ctiles, _p(am2), st)
        dX._dyn_absmax = ((dx_off, ld_dx, M, self.K), am2)
      return sums
    return False


def _act_bwd(st, dY, dy_off, ld_dy, Y, y_off, ld_y, rows, cols, act, dbias=None, seg=1, dseg=None, ld_seg=0):
  am = _Scalars.take(dY.device)
  call('dyn_train_act_bwd', _p(dY, dy_off), _p(Y, y_off) if Y is not None else None, rows, cols, ld_dy, ld_y, act,
       _p(dbias) if dbias is not None else None, seg, _p(dseg) if dseg is not None else None, ld_seg, _p(am), st)
  dY._dyn_absmax = ((dy_off, ld_dy, rows, cols), am)  # largest |dZ|: the scale of the backward GEMMs that consume this tensor


def _rowscale_act_bwd(st, dY, ld_dy, X, ldx, s_row, rows, dX, ld_dx, ds, ds_accumulate, act, dbias):
  """dX[:, :128] = (dX[:, :128] + dY * s_row[:, None]) * act'(X); ds (=|+=) <dY, X> per row; dbias += column sums; dX tagged with its scale."""
  am = _Scalars.take(dX.device)
  call('dyn_train_rowscale_act_bwd', _p(dY), ld_dy, _p(X), ldx, _p(s_row), 1, rows, _p(dX), ld_dx, _p(ds), 1, ds_accumulate, act,
       _p(dbias) if dbias is not None else None, _p(am), st)
  dX._dyn_absmax = ((0, ld_dx, rows, 128), am)


def _split_act_bwd(st, dX2, ld_dx2, dvis0, XV, mask_eff, rows, dXV, dbias, dXS=None, X2=None, ldx2=0, vis0=None):
  """dXV [rows, 132] (129 used) = backward of vis_split through vis_fc.2's ELU (saved output XV [rows, 132]); dbias[129] += column sums.
  With dXS (the gradient of x * vis, vis_fc2.0's input): dX2[:, :128] += dXS * vis0[:, None] and dvis0 = <dXS, X2> first, in the same pass."""
  am = _Scalars.take(dXV.device)
  if dXS is not None:
    _untag(dX2)
  call('dyn_train_vis_split_act_bwd', _p(dX2), ld_dx2, _p(dvis0) if dvis0 is not None else None, _p(XV), 132, _p(mask_eff), rows, _p(dXV), 132,
       _p(dbias), _p(am), _p(dXS) if dXS is not None else None, 128, _p(X2) if X2 is not None else None, ldx2,
       _p(vis0) if vis0 is not None else None, st)
  dXV._dyn_absmax = ((0, 132, rows, 129), am)


PARAM_NAMES = tuple(n for n in ops.STATIC_TENSORS)  # state-dict order; 's' (last) only exists with anti_alias_pooling


def _param_list(net):
  """(names, tensors) of a DynibarStatic module / state dict in PARAM_NAMES order ('s' dropped when the module has none)."""
  net = net.module if hasattr(net, 'module') and not isinstance(net, dict) else net
  sd = dict(net.named_parameters()) if hasattr(net, 'named_parameters') else dict(net)
  sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
  names = [n for n in PARAM_NAMES if n in sd]
  missing = [n for n in PARAM_NAMES if n not in sd and n != 's']
  if missing:
    raise KeyError(f'DynibarStatic parameters missing: {missing[:4]}')
  return names, [sd[n] for n in names]


def wants_grad(net, featmaps):
  """True when a backward pass through the static branch can reach a leaf: grad mode on and either the maps or a parameter require grad."""
  if not torch.is_grad_enabled():
    return False
  if isinstance(featmaps, torch.Tensor) and featmaps.requires_grad:
    return True
  net = net.module if hasattr(net, 'module') and not isinstance(net, dict) else net
  ps = net.parameters() if hasattr(net, 'parameters') else [v for v in net.values() if isinstance(v, torch.Tensor)]
  return any(p.requires_grad for p in ps)


class _Step:
  """One forward pass with everything the backward pass needs (buffers named as in the module docstring)."""

  def drop(self, *names):
    """release saved activations whose last reader has been launched (the allocator hands the memory to the backward pass's own
    buffers: stream-ordered, so the kernels in flight are safe) -- a step's peak is its forward total, not forward + backward"""
    for n in names:
      setattr(self, n, None)


def _forward(w, aa, mask_rgb, views, ray_o, ray_d, pts, rgb_feat, ray_diff, mask):
  """w: {name: device tensor}.  -> (raw [R,S,4], _Step)"""
  R, S, V = rgb_feat.shape[:3]
  P, N = R * S, R * S * V
  dev = rgb_feat.device
  st = stream_of(rgb_feat)
  new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
  s = _Step()
  s.R, s.S, s.V, s.P, s.N, s.aa, s.w, s.views = R, S, V, P, N, aa, w, views
  f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
  ray_o, ray_d, pts, mask, rgb_feat, ray_diff = f32(ray_o), f32(ray_d), f32(pts), f32(mask), f32(rgb_feat), f32(ray_diff)
  s.rgb_feat, s.ray_diff, s.pts = rgb_feat, ray_diff, pts
  # Fourier features / Pluecker coordinates / effective mask
  s.A0, s.REFPE, s.M = new(N, 104), new(R, 68), new(N)
  call('dyn_train_static_embed', _p(pts), _p(ray_o), _p(ray_d), _p(views.proj, 12), 16, _p(ray_diff), _p(rgb_feat), _p(mask), R, S, V,
       int(mask_rgb), _p(s.A0), _p(s.REFPE), _p(s.M), st)
  L = s.L = {}
  L['rd0'] = _Lin(w['ray_dir_fc.0.weight'], w['ray_dir_fc.0.bias'])
  L['rd2'] = _Lin(w['ray_dir_fc.2.weight'], w['ray_dir_fc.2.bias'])
  L['ref'] = _Lin(w['ref_feature_fc.0.weight'], w['ref_feature_fc.0.bias'])
  L['b0g'] = _Lin(w['base_fc.0.weight'], w['base_fc.0.bias'], 0, 140)
  L['b0f'] = _Lin(w['base_fc.0.weight'], None, 140, 70)
  L['b2'] = _Lin(w['base_fc.2.weight'], w['base_fc.2.bias'])
  L['v0'] = _Lin(w['vis_fc.0.weight'], w['vis_fc.0.bias'])
  L['v2'] = _Lin(w['vis_fc.2.weight'], w['vis_fc.2.bias'])
  L['w0'] = _Lin(w['vis_fc2.0.weight'], w['vis_fc2.0.bias'])
  L['w2'] = _Lin(w['vis_fc2.2.weight'], w['vis_fc2.2.bias'])
  L['g0'] = _Lin(w['geometry_fc.0.weight'], w['geometry_fc.0.bias'])
  L['g2'] = _Lin(w['geometry_fc.2.weight'], w['geometry_fc.2.bias'])
  s.Wqkv = torch.cat([w['ray_attention.w_qs.weight'], w['ray_attention.w_ks.weight'], w['ray_attention.w_vs.weight']], 0).contiguous()
  L['qkv'] = _Lin(s.Wqkv)
  L['fc'] = _Lin(w['ray_attention.fc.weight'])
  L['o0'] = _Lin(w['out_geometry_fc.0.weight'], w['out_geometry_fc.0.bias'])
  L['o2'] = _Lin(w['out_geometry_fc.2.weight'], w['out_geometry_fc.2.bias'])
  L['r0g'] = _Lin(w['rgb_fc.0.weight'], w['rgb_fc.0.bias'], 0, 128)
  L['r0x'] = _Lin(w['rgb_fc.0.weight'], None, 128, 133)
  L['r2'] = _Lin(w['rgb_fc.2.weight'], w['rgb_fc.2.bias'])
  L['r4'] = _Lin(w['rgb_fc.4.weight'], w['rgb_fc.4.bias'])
  # ray_dir_fc, ref_feature_fc (mlp_network.py:440-441)
  s.H1, s.SRCF, s.REFF = new(N, 256), new(N, 36), new(R, 36)
  L['rd0'].fwd(st, s.A0, 0, 104, s.H1, 0, 256, N, ELU)
  L['rd2'].fwd(st, s.H1, 0, 256, s.SRCF, 0, 36, N)
  if RECOMPUTE_HIDDEN:
    s.drop('H1')  # recomputed from the 104-wide input in the backward pass
  L['ref'].fwd(st, s.REFPE, 0, 68, s.REFF, 0, 36, R)
  # f = [rgb_feat | src_feat * ref_feat], pooling weights, mean / variance (:450-462)
  s.F, s.w1, s.G1 = new(N, 72), new(N), new(P, 140)
  call('dyn_train_build_f', _p(rgb_feat), _p(s.SRCF), 36, _p(s.REFF), 36, N, S * V, _p(s.F), st)
  call('dyn_train_view_weights', 0, _p(ray_diff, 3), 4, _p(s.M), _p(w['s']) if aa else None, P, V, _p(s.w1), None, 0, None, 0, None, st)
  call('dyn_train_meanvar', _p(s.F), 72, _p(s.w1), P, V, 70, _p(s.G1), _p(s.G1, 70), 140, st)
  # base_fc: the [mean | var] columns once per point, the per-view columns per row (:464-468)
  s.PP1, s.H2, s.X1 = new(P, 256), new(N, 256), new(N, 128)
  L['b0g'].fwd(st, s.G1, 0, 140, s.PP1, 0, 256, P)
  L['b0f'].fwd(st, s.F, 0, 72, s.H2, 0, 256, N, ELU, addend=s.PP1, ld_add=256, add_div=V)
  L['b2'].fwd(st, s.H2, 0, 256, s.X1, 0, 128, N, ELU)
  if RECOMPUTE_HIDDEN:
    s.drop('H2')  # recomputed from f (72 wide) and the per-point part
  # vis_fc on x * weight, residual, first visibility (:470-473)
  s.H3, s.XV = new(N, 128), new(N, 132)
  L['v0'].fwd(st, s.X1, 0, 128, s.H3, 0, 128, N, ELU, rowscale=s.w1)  # vis_fc.0 on x * weight: the scale rides in the epilogue
  L['v2'].fwd(st, s.H3, 0, 128, s.XV, 0, 132, N, ELU)
  s.RIN, s.vis0 = new(N, 136), new(N)   # RIN = [x2 128 | vis 1 | ray_diff 4 | 0 0 0]: rgb_fc.0's per-view input
  call('dyn_train_vis_split', _p(s.X1), 128, _p(s.XV), 132, _p(s.M), _p(ray_diff), N, _p(s.RIN), 136, _p(s.vis0), st)
  # vis_fc2 on x * vis, second visibility, pooled statistics (:474-481)
  s.H4, s.VL = new(N, 128), new(N)
  L['w0'].fwd(st, s.RIN, 0, 136, s.H4, 0, 128, N, ELU, rowscale=s.vis0)  # vis_fc2.0 on x * vis
  L['w2'].fwd(st, s.H4, 0, 128, s.VL, 0, 1, N)
  s.w2, s.G0, s.nvalid = new(N), new(P, 260), new(P)
  call('dyn_train_view_weights', 1, _p(s.VL), 1, _p(s.M), None, P, V, _p(s.w2), _p(s.RIN, 128), 136, _p(s.G0, 256), 260, _p(s.nvalid), st)
  call('dyn_train_meanvar', _p(s.RIN), 136, _p(s.w2), P, V, 128, _p(s.G0), _p(s.G0, 128), 260, st)
  # geometry_fc, ray attention, out_geometry_fc (:482-493)
  s.GH1, s.G2, s.QKV = new(P, 256), new(P, 128), new(P, 384)
  L['g0'].fwd(st, s.G0, 0, 260, s.GH1, 0, 256, P, ELU)
  L['g2'].fwd(st, s.GH1, 0, 256, s.G2, 0, 128, P, ELU)
  L['qkv'].fwd(st, s.G2, 0, 128, s.QKV, 0, 384, P)
  s.AO, s.PROB = new(P, 128), new(R * 4, S, S)
  call('dyn_train_attn', _p(s.QKV), _p(s.nvalid), R, S, _p(s.AO), _p(s.PROB), st)
  s.FCO, s.G3, s.XHAT, s.RSTD = new(P, 128), new(P, 128), new(P, 128), new(P)
  L['fc'].fwd(st, s.AO, 0, 128, s.FCO, 0, 128, P)
  call('dyn_train_layernorm', _p(s.FCO), _p(s.G2), _p(w['ray_attention.layer_norm.weight']), _p(w['ray_attention.layer_norm.bias']), P,
       _p(s.G3), _p(s.XHAT), _p(s.RSTD), st)
  s.O1, s.SIG = new(P, 128), new(P)
  L['o0'].fwd(st, s.G3, 0, 128, s.O1, 0, 128, P, ELU)
  L['o2'].fwd(st, s.O1, 0, 128, s.SIG, 0, 1, P)
  # rgb_fc: the attention output once per point, [x | vis | ray_diff] per row; blending (:495-527)
  s.PP2, s.R1, s.R2, s.RL = new(P, 128), new(N, 128), new(N, 64), new(N)
  L['r0g'].fwd(st, s.G3, 0, 128, s.PP2, 0, 128, P)
  L['r0x'].fwd(st, s.RIN, 0, 136, s.R1, 0, 128, N, ELU, addend=s.PP2, ld_add=128, add_div=V)
  L['r2'].fwd(st, s.R1, 0, 128, s.R2, 0, 64, N, ELU)
  L['r4'].fwd(st, s.R2, 0, 64, s.RL, 0, 1, N)
  s.BW, raw = new(N), new(R, S, 4)
  call('dyn_train_blend', _p(s.RL), 1, _p(s.M), _p(rgb_feat), _p(s.SIG), 1, _p(s.nvalid), P, V, _p(s.BW), _p(raw), st)
  return raw, s


def _backward(s, draw):
  """draw [R,S,4] -> ({param name: grad}, dF [N,72] whose columns 0..34 are d rgb_feat)"""
  R, S, V, P, N, w, L = s.R, s.S, s.V, s.P, s.N, s.w, s.L
  dev = draw.device
  st = stream_of(draw)
  new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
  g = zero_grads(dict(w, **{'__qkv': s.Wqkv}))
  gqkv = g.pop('__qkv')
  draw = draw.contiguous()
  # blending softmax / density fill
  dRL, dSIG = new(N), new(P)
  call('dyn_train_blend_bwd', _p(draw), _p(s.BW), _p(s.M), _p(s.rgb_feat), _p(s.nvalid), P, V, _p(dRL), 1, _p(dSIG), 1, st)
  s.drop('BW', 'RL')
  # rgb_fc.4 / .2 / .0
  dR2, dR1 = new(N, 64), new(N, 128)
  _act_bwd(st, dRL, 0, 1, None, 0, 1, N, 1, NONE, g['rgb_fc.4.bias'])
  if not L['r4'].bwd(st, dRL, 0, 1, s.R2, 0, 64, g['rgb_fc.4.weight'], N, dR2, 0, 64, act_y=(s.R2, 0, 64, ELU), dbias=g['rgb_fc.2.bias']):  # dR2 arrives times ELU'(R2)
    _act_bwd(st, dR2, 0, 64, None, 0, 64, N, 64, NONE, g['rgb_fc.2.bias'])
  del dRL
  L['r2'].bwd(st, dR2, 0, 64, s.R1, 0, 128, g['rgb_fc.2.weight'], N, dR1, 0, 128, act_y=(s.R1, 0, 128, ELU))
  del dR2
  s.drop('R2', 'R1')
  dPP2 = new(P, 128)
  _act_bwd(st, dR1, 0, 128, None, 0, 128, N, 128, NONE, g['rgb_fc.0.bias'], V, dPP2, 128)
  dRIN = new(N, 136)  # gradient of [x2 | vis | ray_diff]; its first 128 columns go on to collect every gradient of x2, then of x1
  L['r0x'].bwd(st, dR1, 0, 128, s.RIN, 0, 136, g['rgb_fc.0.weight'], N, dRIN, 0, 136)
  del dR1
  dG3 = new(P, 128)
  L['r0g'].bwd(st, dPP2, 0, 128, s.G3, 0, 128, g['rgb_fc.0.weight'], P, dG3, 0, 128)
  # out_geometry_fc
  dO1 = new(P, 128)
  _act_bwd(st, dSIG, 0, 1, None, 0, 1, P, 1, NONE, g['out_geometry_fc.2.bias'])
  if not L['o2'].bwd(st, dSIG, 0, 1, s.O1, 0, 128, g['out_geometry_fc.2.weight'], P, dO1, 0, 128, act_y=(s.O1, 0, 128, ELU), dbias=g['out_geometry_fc.0.bias']):
    _act_bwd(st, dO1, 0, 128, None, 0, 128, P, 128, NONE, g['out_geometry_fc.0.bias'])
  L['o0'].bwd(st, dO1, 0, 128, s.G3, 0, 128, g['out_geometry_fc.0.weight'], P, dG3, 0, 128, acc_dx=1)
  # LayerNorm(fc(attention) + g2): dY is the gradient of both summands; it then collects the rest of g2's gradient
  dY = new(P, 128)
  call('dyn_train_layernorm_bwd', _p(dG3), _p(s.XHAT), _p(s.RSTD), _p(w['ray_attention.layer_norm.weight']), P, _p(dY),
       _p(g['ray_attention.layer_norm.weight']), _p(g['ray_attention.layer_norm.bias']), st)
  dAO, dQKV, dSC = new(P, 128), new(P, 384), new(R * 4, S, S)
  L['fc'].bwd(st, dY, 0, 128, s.AO, 0, 128, g['ray_attention.fc.weight'], P, dAO, 0, 128)
  call('dyn_train_attn_bwd', _p(s.QKV), _p(s.nvalid), R, S, _p(s.PROB), _p(dAO), _p(dSC), _p(dQKV), st)
  L['qkv'].bwd(st, dQKV, 0, 384, s.G2, 0, 128, gqkv, P, dY, 0, 128, acc_dx=1)
  g['ray_attention.w_qs.weight'], g['ray_attention.w_ks.weight'], g['ray_attention.w_vs.weight'] = gqkv[0:128], gqkv[128:256], gqkv[256:384]
  # geometry_fc
  dGH1, dG0 = new(P, 256), new(P, 260)
  _act_bwd(st, dY, 0, 128, s.G2, 0, 128, P, 128, ELU, g['geometry_fc.2.bias'])
  if not L['g2'].bwd(st, dY, 0, 128, s.GH1, 0, 256, g['geometry_fc.2.weight'], P, dGH1, 0, 256, act_y=(s.GH1, 0, 256, ELU), dbias=g['geometry_fc.0.bias']):
    _act_bwd(st, dGH1, 0, 256, None, 0, 256, P, 256, NONE, g['geometry_fc.0.bias'])
  L['g0'].bwd(st, dGH1, 0, 256, s.G0, 0, 260, g['geometry_fc.0.weight'], P, dG0, 0, 260)
  # pooled statistics of x2 under the visibility weights; the weights themselves
  dw2, dVL = new(N), new(N)
  _untag(dRIN)
  call('dyn_train_meanvar_bwd', _p(s.RIN), 136, _p(s.w2), P, V, 128, _p(s.G0), _p(dG0), _p(dG0, 128), 260, _p(dRIN), 136, 1, _p(dw2), 0, st)
  call('dyn_train_view_weights_bwd', 1, _p(s.VL), 1, _p(s.M), None, P, V, _p(s.w2), _p(dw2), _p(dRIN, 128), 136, _p(s.RIN, 128), 136,
       _p(dG0, 256), 260, _p(dVL), 1, None, st)
  # vis_fc2
  dH4, dXS = new(N, 128), new(N, 128)
  _act_bwd(st, dVL, 0, 1, None, 0, 1, N, 1, NONE, g['vis_fc2.2.bias'])
  if not L['w2'].bwd(st, dVL, 0, 1, s.H4, 0, 128, g['vis_fc2.2.weight'], N, dH4, 0, 128, act_y=(s.H4, 0, 128, ELU), dbias=g['vis_fc2.0.bias']):
    _act_bwd(st, dH4, 0, 128, None, 0, 128, N, 128, NONE, g['vis_fc2.0.bias'])
  s.drop('H4')
  L['w0'].bwd(st, dH4, 0, 128, s.RIN, 0, 136, g['vis_fc2.0.weight'], N, dXS, 0, 128, x_scale=s.vis0)  # the layer ran on x * vis
  del dH4
  # x2 = x1 + x_res, vis0 = sigmoid(.) mask: dRIN[:, :128] is now d x2 = d x1 (so far) = d x_res
  dXV = new(N, 132)
  # the row-scale backward of x * vis, the split's backward and vis_fc.2's ELU in one pass
  _split_act_bwd(st, dRIN, 136, None, s.XV, s.M, N, dXV, g['vis_fc.2.bias'], dXS=dXS, X2=s.RIN, ldx2=136, vis0=s.vis0)
  del dXS
  s.drop('RIN')
  dH3, dXW = new(N, 128), new(N, 128)
  if not L['v2'].bwd(st, dXV, 0, 132, s.H3, 0, 128, g['vis_fc.2.weight'], N, dH3, 0, 128, act_y=(s.H3, 0, 128, ELU), dbias=g['vis_fc.0.bias']):
    _act_bwd(st, dH3, 0, 128, None, 0, 128, N, 128, NONE, g['vis_fc.0.bias'])
  del dXV
  s.drop('XV', 'H3')
  L['v0'].bwd(st, dH3, 0, 128, s.X1, 0, 128, g['vis_fc.0.weight'], N, dXW, 0, 128, x_scale=s.w1)  # the layer ran on x * weight
  del dH3
  dw1 = new(N)
  # d x1 is complete with this term: its row-scale backward and base_fc.2's ELU in one pass
  _rowscale_act_bwd(st, dXW, 128, s.X1, 128, s.w1, N, dRIN, 136, dw1, 0, ELU, g['base_fc.2.bias'])
  # base_fc
  dH2, dPP1, dF, dG1 = new(N, 256), new(P, 256), new(N, 72), new(P, 140)
  H2 = s.H2
  if H2 is None:  # RECOMPUTE_HIDDEN: the hidden layer of base_fc again (the same launch as in the forward pass: bit-identical)
    H2 = new(N, 256)
    L['b0f'].fwd(st, s.F, 0, 72, H2, 0, 256, N, ELU, addend=s.PP1, ld_add=256, add_div=V)
  L['b2'].bwd(st, dRIN, 0, 136, H2, 0, 256, g['base_fc.2.weight'], N, dH2, 0, 256, act_y=(H2, 0, 256, ELU))
  del dRIN, dXW, H2
  s.drop('X1', 'H2')
  _act_bwd(st, dH2, 0, 256, None, 0, 256, N, 256, NONE, g['base_fc.0.bias'], V, dPP1, 256)
  L['b0f'].bwd(st, dH2, 0, 256, s.F, 0, 72, g['base_fc.0.weight'], N, dF, 0, 72)
  del dH2
  L['b0g'].bwd(st, dPP1, 0, 256, s.G1, 0, 140, g['base_fc.0.weight'], P, dG1, 0, 140)
  _untag(dF)
  call('dyn_train_meanvar_bwd', _p(s.F), 72, _p(s.w1), P, V, 70, _p(s.G1), _p(dG1), _p(dG1, 70), 140, _p(dF), 72, 1, _p(dw1), 1, st)
  if s.aa:
    call('dyn_train_view_weights_bwd', 0, _p(s.ray_diff, 3), 4, _p(s.M), _p(w['s']), P, V, _p(s.w1), _p(dw1), None, 0, None, 0, None, 0, None, 0,
         _p(g['s']), st)
  # f = [rgb_feat | src_feat * ref_feat]; ref_feature_fc; ray_dir_fc
  dSRCF, dREFF, dH1 = new(N, 36), new(R, 36), new(N, 256)
  call('dyn_train_build_f_bwd', _p(dF), 72, _p(s.SRCF), 36, _p(s.REFF), 36, R, S * V, _p(dSRCF), 36, _p(dREFF), 36, st)
  _act_bwd(st, dREFF, 0, 36, None, 0, 36, R, 35, NONE, g['ref_feature_fc.0.bias'])
  L['ref'].bwd(st, dREFF, 0, 36, s.REFPE, 0, 68, g['ref_feature_fc.0.weight'], R)
  _act_bwd(st, dSRCF, 0, 36, None, 0, 36, N, 35, NONE, g['ray_dir_fc.2.bias'])
  H1 = s.H1
  if H1 is None:  # RECOMPUTE_HIDDEN: the hidden layer of ray_dir_fc again
    H1 = new(N, 256)
    L['rd0'].fwd(st, s.A0, 0, 104, H1, 0, 256, N, ELU)
  if not L['rd2'].bwd(st, dSRCF, 0, 36, H1, 0, 256, g['ray_dir_fc.2.weight'], N, dH1, 0, 256, act_y=(H1, 0, 256, ELU), dbias=g['ray_dir_fc.0.bias']):
    _act_bwd(st, dH1, 0, 256, None, 0, 256, N, 256, NONE, g['ray_dir_fc.0.bias'])
  del dSRCF, H1
  s.drop('F', 'SRCF', 'H1')
  L['rd0'].bwd(st, dH1, 0, 256, s.A0, 0, 104, g['ray_dir_fc.0.weight'], N)
  return g, dF  # dF[:, 0:35] = d rgb_feat (the gather's backward, train_motion.GatherFunction, carries it on into the maps)


class StaticNetFunction(torch.autograd.Function):
  """raw_static [R,S,4] = DynibarStatic(gathered static features) with gradients to rgb_feat (and through the gather to the feature
  maps) and to the parameters."""

  @staticmethod
  def forward(ctx, rgb_feat, meta, *param_tensors):
    names, aa, mask_rgb, views, ray_o, ray_d, pts, ray_diff, mask = meta
    w = {n: (t.detach() if t.dtype == torch.float32 and t.is_contiguous() else t.detach().float().contiguous()) for n, t in zip(names, param_tensors)}
    w = {n: (t.reshape(1) if t.dim() == 0 else t) for n, t in w.items()}
    raw, step = _forward(w, aa, mask_rgb, views, ray_o, ray_d, pts.contiguous(), rgb_feat.detach(), ray_diff, mask)
    ctx.step, ctx.names, ctx.shapes = step, names, [tuple(t.shape) for t in param_tensors]
    ctx.fshape = tuple(rgb_feat.shape)
    return raw

  @staticmethod
  def backward(ctx, draw):
    if ctx.step is None:
      raise RuntimeError('StaticNetFunction: the saved activations were released by the first backward pass; call the renderer again instead of backward(retain_graph=True)')
    g, dF = _backward(ctx.step, draw.float())
    ctx.step = None  # the saved activations are released with the step
    gr = dF[:, :35].reshape(ctx.fshape) if ctx.needs_input_grad[0] else None  # a column-slice view: the gather's backward reads it in place
    gp = tuple(g[n].reshape(shp) if ctx.needs_input_grad[2 + i] else None for i, (n, shp) in enumerate(zip(ctx.names, ctx.shapes)))
    return (gr, None) + gp


class CompositeVanillaFunction(torch.autograd.Function):
  """raw2outputs_vanilla (render_ray.py:134-201): (rgb, depth, weights) differentiable w.r.t. raw; mask / alpha are forward values."""

  @staticmethod
  def forward(ctx, raw, z_vals, pix_mask):
    out = ops.composite(raw.detach(), z_vals, pix_mask)
    ctx.save_for_backward(raw.detach().contiguous(), out['z_vals'], out['alpha'], out['weights'])
    ctx.mark_non_differentiable(out['mask'], out['alpha'])
    return out['rgb'], out['depth'], out['weights'], out['mask'], out['alpha']

  @staticmethod
  def backward(ctx, drgb, ddepth, dweights, _dmask, _dalpha):
    raw, z_vals, alpha, weights = ctx.saved_tensors
    R, S = z_vals.shape
    c = lambda t: None if t is None else _p(t.float().contiguous())
    keep = [t.float().contiguous() if t is not None else None for t in (drgb, ddepth, dweights)]
    draw = torch.empty_like(raw)
    call('dyn_train_composite_bwd', _p(raw), _p(z_vals), _p(alpha), _p(weights), *[None if t is None else _p(t) for t in keep], R, S, _p(draw),
         stream_of(raw))
    return draw, None, None


def static_raw(net, args_flags, views, rgb_feat, ray_o, ray_d, pts, ray_diff, mask):
  """raw_static with an autograd graph.  net: the reference's DynibarStatic (nn.Module, DataParallel-wrapped or not) or a dict of
  parameter tensors; args_flags = (anti_alias_pooling, mask_rgb); rgb_feat: the gathered static features -- from
  train_motion.gather(...) when the feature maps take part in the graph."""
  aa, mask_rgb = args_flags
  names, tensors = _param_list(net)
  if aa and 's' not in names:
    raise KeyError("DynibarStatic has no parameter 's' but anti_alias_pooling is on (mlp_network.py:330-331)")
  if not aa and 's' in names:
    i = names.index('s')
    names, tensors = names[:i] + names[i + 1:], tensors[:i] + tensors[i + 1:]
  meta = (names, bool(aa), bool(mask_rgb), views, ray_o, ray_d, pts, ray_diff, mask)
  return StaticNetFunction.apply(rgb_feat, meta, *tensors)


def composite_vanilla(raw, z_vals, pix_mask):
  """-> dict with the reference's key set (render_ray.py:202-211); rgb / depth / weights carry the graph."""
  rgb, depth, weights, m, alpha = CompositeVanillaFunction.apply(raw, z_vals, pix_mask)
  from collections import OrderedDict
  return OrderedDict([('rgb', rgb), ('depth', depth), ('weights', weights), ('mask', m > 0), ('alpha', alpha), ('z_vals', z_vals)])
