"""Training form of the feature encoder (the reference optimises ``feature_net`` together with the MLPs,
train.py:272-281): forward WITH saved activations and the backward pass of the executed part of ``ResNet.forward``
(ibrnet/feature_network.py:179-311 -- conv1 7x7 / 2 -> InstanceNorm -> ReLU -> layer1 (three BasicBlocks, :44-83) -> 1x1 out_conv),
on the HIP kernels: every convolution is an explicit im2col (``dyn_enc_im2col``: channels-last patches, reflect padding) + the training
GEMM in its three roles (``dyn_train_gemm`` through ``train_static._Lin``), InstanceNorm / residual / ReLU and their backward are row
kernels over fp64 statistics tables (``dyn_enc_in_*``), patch gradients return to the maps through ``dyn_enc_col2im``.  PyTorch carries
the tensors between the kernels and the autograd edge to the optimizer; the weight re-layouts ([oc, ic, ky, kx] <-> [oc, (ky, kx, ic)])
are views / small copies of the parameter tensors.

The inference encoder (csrc/dyn_encoder.hip, ``ops.Encoder``) keeps nothing and fetches its operands straight from the maps; this form
materialises the patch matrices and keeps them for the backward pass (a 3x3 convolution of 15 quarter-resolution maps: 320 MB each) -- it is
built for gradients, and runs once per training iteration on the source views.
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import call, stream_of
from .train_static import NONE, _Lin, _act_bwd, _p

PARAMS = (['conv1.weight', 'bn1.weight', 'bn1.bias'] +
          [f'layer1.{b}.{n}' for b in range(3) for n in ('conv1.weight', 'bn1.weight', 'bn1.bias', 'conv2.weight', 'bn2.weight', 'bn2.bias')] +
          ['layer1.0.downsample.0.weight', 'layer1.0.downsample.1.weight', 'layer1.0.downsample.1.bias', 'out_conv.weight', 'out_conv.bias'])


def _pd(t):
  """device pointer of a contiguous float64 tensor"""
  assert t.dtype == torch.float64 and t.is_contiguous()
  return ctypes.c_void_p(t.data_ptr())


def _out_hw(H, W, k, stride, pad):
  return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


class _Conv:
  """One convolution as im2col + GEMM: weight [64, C, k, k] -> operand [64, k * k * C] in (ky, kx, ic) order."""

  def __init__(self, W, k, stride, pad):
    self.k, self.stride, self.pad, self.C = k, stride, pad, W.shape[1]
    self.Wm = W.detach().permute(0, 2, 3, 1).reshape(W.shape[0], -1).contiguous().float()
    self.lin = _Lin(self.Wm)
    self.K = self.Wm.shape[1]
    self.ldc = (self.K + 3) // 4 * 4

  def patches(self, st, x):
    """x [N,H,W,C] -> (col [rows, ldc], Ho, Wo); a 1x1 / 1 convolution reads the map itself"""
    N, H, W, C = x.shape
    Ho, Wo = _out_hw(H, W, self.k, self.stride, self.pad)
    if self.k == 1 and self.stride == 1:
      return x.reshape(N * H * W, C), Ho, Wo
    col = torch.empty((N * Ho * Wo, self.ldc), dtype=torch.float32, device=x.device)  # (dyn_enc_im2col writes the padding columns as zeros)
    call('dyn_enc_im2col', _p(x), N, H, W, C, self.k, self.k, self.stride, self.pad, Ho, Wo, _p(col), self.ldc, st)
    return col, Ho, Wo

  def fwd(self, st, x):
    N = x.shape[0]
    col, Ho, Wo = self.patches(st, x)
    y = torch.empty((N, Ho, Wo, self.Wm.shape[0]), dtype=torch.float32, device=x.device)
    self.lin.fwd(st, col, 0, col.shape[1], y, 0, y.shape[-1], col.shape[0])
    self.col = (col, Ho, Wo)  # kept for the weight gradient (9x the map for a 3x3 convolution: 2.3 GB for 18 images -- cheaper than forming it again)
    return y

  def bwd(self, st, x, dy, dx=None):
    """dy [N,Ho,Wo,64] (consumed) -> weight gradient in the parameter's layout; dx [N,H,W,C] += the input gradient when given"""
    N, H, W, C = x.shape
    col, Ho, Wo = self.col if getattr(self, 'col', None) is not None else self.patches(st, x)
    self.col = None
    rows = col.shape[0]
    dWm = torch.zeros_like(self.Wm)
    dcol = None
    if dx is not None:
      dcol = dx.reshape(rows, C) if (self.k == 1 and self.stride == 1) else torch.empty((rows, self.ldc), dtype=torch.float32, device=x.device)
    if dcol is not None and self.k == 1 and self.stride == 1:
      # the patch matrix IS the map: the data gradient accumulates straight into it
      self.lin.bwd(st, dy.reshape(rows, -1), 0, dy.shape[-1], col, 0, col.shape[1], dWm, rows, dcol, 0, C, acc_dx=1)
    else:
      self.lin.bwd(st, dy.reshape(rows, -1), 0, dy.shape[-1], col, 0, col.shape[1], dWm, rows, dcol, 0, self.ldc if dcol is not None else 0)
      if dcol is not None:
        call('dyn_enc_col2im', _p(dcol), self.ldc, N, H, W, C, self.k, self.k, self.stride, self.pad, Ho, Wo, _p(dx), st)
    return dWm.reshape(self.Wm.shape[0], self.k, self.k, C).permute(0, 3, 1, 2).contiguous()


class _Norm:
  """InstanceNorm2d(affine=True) (+ residual) (+ ReLU) over a channels-last map, statistics in an fp64 table"""

  def __init__(self, gamma, beta):
    self.gamma, self.beta = gamma.detach().float().contiguous(), beta.detach().float().contiguous()

  def fwd(self, st, x, res=None, relu=True):
    N, H, W, C = x.shape
    assert C == 64
    self.stats = torch.zeros((N, 64, 2), dtype=torch.float64, device=x.device)
    call('dyn_enc_in_stats', _p(x), N, H * W, _pd(self.stats), st)
    y = torch.empty_like(x)
    call('dyn_enc_in_apply', _p(x), _pd(self.stats), _p(self.gamma), _p(self.beta), _p(res) if res is not None else None, int(relu), N, H * W, _p(y), st)
    self.relu = relu
    return y

  def bwd(self, st, dy, y, x, want_res=False):
    """-> (dx, dres or None, dgamma, dbeta)"""
    N, H, W, C = x.shape
    sums2 = torch.empty((N, 64, 2), dtype=torch.float64, device=x.device)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_res else None
    dg, db = torch.zeros(64, dtype=torch.float32, device=x.device), torch.zeros(64, dtype=torch.float32, device=x.device)
    call('dyn_enc_in_bwd', _p(dy), _p(y) if self.relu else None, int(self.relu), _p(x), _pd(self.stats), _p(self.gamma), N, H * W, _pd(sums2), _p(dx),
         _p(dres) if dres is not None else None, _p(dg), _p(db), st)
    return dx, dres, dg, db


def _forward(w, img):
  """w: {name: tensor}; img [N,H,W,3] fp32 contiguous on the device -> (coarse [N,Hf,Wf,32], fine, saved state)"""
  st = stream_of(img)
  s = {}
  c1 = _Conv(w['conv1.weight'], 7, 2, 3)
  n1 = _Norm(w['bn1.weight'], w['bn1.bias'])
  a0 = c1.fwd(st, img)
  x = n1.fwd(st, a0, relu=True)
  s['stem'] = (c1, n1, a0, x)
  blocks = []
  for b in range(3):
    pre = f'layer1.{b}.'
    stride = 2 if b == 0 else 1
    ca, na = _Conv(w[pre + 'conv1.weight'], 3, stride, 1), _Norm(w[pre + 'bn1.weight'], w[pre + 'bn1.bias'])
    cb, nb = _Conv(w[pre + 'conv2.weight'], 3, 1, 1), _Norm(w[pre + 'bn2.weight'], w[pre + 'bn2.bias'])
    a1 = ca.fwd(st, x)
    h1 = na.fwd(st, a1, relu=True)
    a2 = cb.fwd(st, h1)
    if b == 0:
      cd, nd = _Conv(w[pre + 'downsample.0.weight'], 1, 2, 0), _Norm(w[pre + 'downsample.1.weight'], w[pre + 'downsample.1.bias'])
      ad = cd.fwd(st, x)
      idn = nd.fwd(st, ad, relu=False)
      ds = (cd, nd, ad)
    else:
      idn, ds = x, None
    y = nb.fwd(st, a2, res=idn, relu=True)
    blocks.append((x, ca, na, a1, h1, cb, nb, a2, ds, y))
    x = y
  # out_conv (1x1, bias): the coarse and the fine half as two products, so that each is a contiguous [N,Hf,Wf,32] map (what the gather taps)
  Wo = w['out_conv.weight'].detach().reshape(64, 64).float()
  bo = w['out_conv.bias'].detach().float()
  N, Hf, Wf, _ = x.shape
  rows = N * Hf * Wf
  outs, lins = [], []
  for half in range(2):
    lin = _Lin(Wo[32 * half:32 * half + 32].contiguous(), bo[32 * half:32 * half + 32].contiguous())
    o = torch.empty((N, Hf, Wf, 32), dtype=torch.float32, device=img.device)
    lin.fwd(st, x.reshape(rows, 64), 0, 64, o, 0, 32, rows)
    outs.append(o)
    lins.append(lin)
  s['blocks'], s['out'], s['img'], s['x3'] = blocks, lins, img, x
  return outs[0], outs[1], s


def _backward(s, dcoarse, dfine):
  """-> {param name: gradient}"""
  img = s['img']
  st = stream_of(img)
  g = {}
  x3 = s['x3']
  N, Hf, Wf, _ = x3.shape
  rows = N * Hf * Wf
  dx = torch.zeros_like(x3)
  gW, gb = [], []
  for half, d in enumerate((dcoarse, dfine)):
    d = d.reshape(rows, 32)
    db = torch.zeros(32, dtype=torch.float32, device=img.device)
    _act_bwd(st, d, 0, 32, None, 0, 32, rows, 32, NONE, db)  # bias gradient (column sums) and the scale of the GEMM operand
    dW = torch.zeros((32, 64), dtype=torch.float32, device=img.device)
    s['out'][half].bwd(st, d, 0, 32, x3.reshape(rows, 64), 0, 64, dW, rows, dx.reshape(rows, 64), 0, 64, acc_dx=1)
    gW.append(dW)
    gb.append(db)
  g['out_conv.weight'] = torch.cat(gW, 0).reshape(64, 64, 1, 1)
  g['out_conv.bias'] = torch.cat(gb, 0)
  for b in (2, 1, 0):
    pre = f'layer1.{b}.'
    xin, ca, na, a1, h1, cb, nb, a2, ds, y = s['blocks'][b]
    da2, dres, g[pre + 'bn2.weight'], g[pre + 'bn2.bias'] = nb.bwd(st, dx, y, a2, want_res=True)
    dh1 = torch.zeros_like(h1)
    g[pre + 'conv2.weight'] = cb.bwd(st, h1, da2, dh1)
    da1, _, g[pre + 'bn1.weight'], g[pre + 'bn1.bias'] = na.bwd(st, dh1, h1, a1)
    if ds is None:
      dxin = dres  # the identity branch's gradient; the convolution's patch gradients are added onto it
    else:
      dxin = torch.zeros_like(xin)
      cd, nd, ad = ds
      dad, _, g[pre + 'downsample.1.weight'], g[pre + 'downsample.1.bias'] = nd.bwd(st, dres, None, ad)
      g[pre + 'downsample.0.weight'] = cd.bwd(st, xin, dad, dxin)
    g[pre + 'conv1.weight'] = ca.bwd(st, xin, da1, dxin)
    dx = dxin
  c1, n1, a0, x0 = s['stem']
  da0, _, g['bn1.weight'], g['bn1.bias'] = n1.bwd(st, dx, x0, a0)
  g['conv1.weight'] = c1.bwd(st, img, da0, None)
  return g


class EncoderFunction(torch.autograd.Function):
  """(coarse, fine) [N,Hf,Wf,32] channels-last = encoder(img [N,H,W,3]) with gradients to the encoder's parameters (not to the images:
  the reference's source views are data)."""

  @staticmethod
  def forward(ctx, img, *param_tensors):
    w = dict(zip(PARAMS, param_tensors))
    coarse, fine, state = _forward(w, img)
    ctx.state, ctx.shapes = state, [tuple(t.shape) for t in param_tensors]
    return coarse, fine

  @staticmethod
  def backward(ctx, dcoarse, dfine):
    if ctx.state is None:
      raise RuntimeError('EncoderFunction: the saved activations were released by the first backward pass')
    z = lambda d, ref: torch.zeros_like(ref) if d is None else d.float().contiguous()
    g = _backward(ctx.state, z(dcoarse, ctx.state['x3'][..., :32]), z(dfine, ctx.state['x3'][..., :32]))
    ctx.state = None
    return (None,) + tuple(g[n].reshape(shp) if ctx.needs_input_grad[1 + i] else None for i, (n, shp) in enumerate(zip(PARAMS, ctx.shapes)))


def encoder_forward(module_or_params, x):
  """x [N,3,H,W] (any strides) -> (x_coarse, x_fine) [N,32,Hf,Wf] as NCHW views of channels-last maps, with an autograd graph into the
  encoder's parameters.  module_or_params: the reference's ResNet module (DataParallel-wrapped or not) or a {name: tensor} dict."""
  src = module_or_params.module if hasattr(module_or_params, 'module') and not isinstance(module_or_params, dict) else module_or_params
  sd = dict(src.named_parameters()) if hasattr(src, 'named_parameters') else dict(src)
  sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
  missing = [n for n in PARAMS if n not in sd]
  if missing:
    raise KeyError(f'feature encoder parameters missing: {missing[:4]}')
  img = x.permute(0, 2, 3, 1)
  if img.dtype != torch.float32 or not img.is_contiguous():
    img = img.float().contiguous()
  coarse, fine = EncoderFunction.apply(img.detach(), *[sd[n] for n in PARAMS])
  return coarse.permute(0, 3, 1, 2), fine.permute(0, 3, 1, 2)
