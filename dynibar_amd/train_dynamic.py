"""Training of the dynamic branch's network and of the two-branch compositing on the HIP kernels (second
slice): ``DynibarDynamic.forward`` (mlp_network.py:236-316) and ``raw2outputs`` (render_ray.py:214-330) as ``torch.autograd.Function``s
whose forward and backward are sequences of the ``dyn_train_*`` kernels, like ``train_static``.

``raw_dy`` carries a graph to DynibarDynamic's parameters, to the gathered features (and through train_motion.GatherFunction to the
dynamic feature maps and the motion-displaced points) and to the points handed to the net; every colour / depth / weight output of
the two-branch compositing carries one to ``raw_dy`` and ``raw_static``.
"""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import call, stream_of
from . import train_static as TS
from .train_static import ELU, NONE, _Lin, _Step, _act_bwd, _p, _rowscale_act_bwd, _split_act_bwd, _untag, zero_grads

PARAM_NAMES = ops.DYNAMIC_TENSORS


def _param_list(net):
  net = net.module if hasattr(net, 'module') and not isinstance(net, dict) else net
  sd = dict(net.named_parameters()) if hasattr(net, 'named_parameters') else dict(net)
  sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
  missing = [n for n in PARAM_NAMES if n not in sd]
  if missing:
    raise KeyError(f'DynibarDynamic parameters missing: {missing[:4]}')
  return list(PARAM_NAMES), [sd[n] for n in PARAM_NAMES]


def _time_pe(time, dev):
  """PeriodicEmbed(10 octaves) of the scalar time embedding (mlp_network.py:147-150,244): [1,24] = [t | cos 2^f t | sin 2^f t | 0 0 0].
  One scalar per call and no gradient: prepared on the host side of the boundary."""
  t = time.reshape(-1)[:1].to(dev).float()
  f = 2.0 ** torch.arange(10, device=dev, dtype=torch.float32)
  return torch.cat([t, torch.cos(f * t), torch.sin(f * t), torch.zeros(3, device=dev)]).reshape(1, 24).contiguous()


def _forward(w, shift, pos_table, ray_d, pts, rgb_feat, mask, time):
  R, S, V = rgb_feat.shape[:3]
  P, N = R * S, R * S * V
  dev = rgb_feat.device
  st = stream_of(rgb_feat)
  new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
  f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
  ray_d, pts, rgb_feat, mask = f32(ray_d), f32(pts), f32(rgb_feat), f32(mask)
  s = _Step()
  s.R, s.S, s.V, s.P, s.N, s.w = R, S, V, P, N, w
  s.M = mask.reshape(-1)
  L = s.L = {}
  L['rd0'] = _Lin(w['ray_dir_fc.0.weight'], w['ray_dir_fc.0.bias'])
  L['rd2'] = _Lin(w['ray_dir_fc.2.weight'], w['ray_dir_fc.2.bias'])
  L['b0g'] = _Lin(w['base_fc.0.weight'], w['base_fc.0.bias'], 0, 70)
  L['b0f'] = _Lin(w['base_fc.0.weight'], None, 70, 35)
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
  L['p0g'] = _Lin(w['ref_pts_fc.0.weight'], w['ref_pts_fc.0.bias'], 0, 128)
  L['p0p'] = _Lin(w['ref_pts_fc.0.weight'], None, 128, 33)
  L['p2'] = _Lin(w['ref_pts_fc.2.weight'], w['ref_pts_fc.2.bias'])
  L['o0'] = _Lin(w['out_geometry_fc.0.weight'], w['out_geometry_fc.0.bias'])
  L['o2'] = _Lin(w['out_geometry_fc.2.weight'], w['out_geometry_fc.2.bias'])
  L['r0g'] = _Lin(w['rgb_fc.0.weight'], w['rgb_fc.0.bias'], 0, 128)
  L['r0d'] = _Lin(w['rgb_fc.0.weight'], None, 128, 27)
  L['r2'] = _Lin(w['rgb_fc.2.weight'], w['rgb_fc.2.bias'])
  L['r4'] = _Lin(w['rgb_fc.4.weight'], w['rgb_fc.4.bias'])
  # time feature (mlp_network.py:244-249): one 35-vector for the whole batch
  s.TPE = _time_pe(time, dev)
  s.DH1, s.DIRF = new(1, 256), new(1, 36)
  L['rd0'].fwd(st, s.TPE, 0, 24, s.DH1, 0, 256, 1, ELU)
  L['rd2'].fwd(st, s.DH1, 0, 256, s.DIRF, 0, 36, 1, ELU)
  s.F, s.w1, s.G1 = new(N, 36), new(N), new(P, 72)
  call('dyn_train_add_table', _p(rgb_feat), 35, _p(s.DIRF), 36, 1, N, 35, _p(s.F), 36, st)
  call('dyn_train_view_weights', 0, None, 0, _p(s.M), None, P, V, _p(s.w1), None, 0, None, 0, None, st)
  call('dyn_train_meanvar', _p(s.F), 36, _p(s.w1), P, V, 35, _p(s.G1), _p(s.G1, 35), 72, st)
  s.PP1, s.H2, s.X1 = new(P, 256), new(N, 256), new(N, 128)
  L['b0g'].fwd(st, s.G1, 0, 72, s.PP1, 0, 256, P)
  L['b0f'].fwd(st, s.F, 0, 36, s.H2, 0, 256, N, ELU, addend=s.PP1, ld_add=256, add_div=V)
  L['b2'].fwd(st, s.H2, 0, 256, s.X1, 0, 128, N, ELU)
  if TS.RECOMPUTE_HIDDEN:
    s.drop('H2')  # recomputed from f (36 wide) and the per-point part in the backward pass (train_static.RECOMPUTE_HIDDEN)
  s.H3, s.XV = new(N, 128), new(N, 132)
  L['v0'].fwd(st, s.X1, 0, 128, s.H3, 0, 128, N, ELU, rowscale=s.w1)  # vis_fc.0 on x * weight: the scale rides in the epilogue
  L['v2'].fwd(st, s.H3, 0, 128, s.XV, 0, 132, N, ELU)
  s.X2, s.vis0 = new(N, 128), new(N)
  call('dyn_train_vis_split', _p(s.X1), 128, _p(s.XV), 132, _p(s.M), None, N, _p(s.X2), 128, _p(s.vis0), st)
  s.H4, s.VL = new(N, 128), new(N)
  L['w0'].fwd(st, s.X2, 0, 128, s.H4, 0, 128, N, ELU, rowscale=s.vis0)  # vis_fc2.0 on x * vis
  L['w2'].fwd(st, s.H4, 0, 128, s.VL, 0, 1, N)
  s.w2, s.VIS, s.G0, s.nvalid = new(N), new(N), new(P, 260), new(P)
  call('dyn_train_view_weights', 1, _p(s.VL), 1, _p(s.M), None, P, V, _p(s.w2), _p(s.VIS), 1, _p(s.G0, 256), 260, _p(s.nvalid), st)
  call('dyn_train_meanvar', _p(s.X2), 128, _p(s.w2), P, V, 128, _p(s.G0), _p(s.G0, 128), 260, st)
  # geometry_fc, + positional table, ray attention (:271-288)
  s.GH1, s.G2, s.G2P, s.QKV = new(P, 256), new(P, 128), new(P, 128), new(P, 384)
  L['g0'].fwd(st, s.G0, 0, 260, s.GH1, 0, 256, P, ELU)
  L['g2'].fwd(st, s.GH1, 0, 256, s.G2, 0, 128, P, ELU)
  call('dyn_train_add_table', _p(s.G2), 128, _p(pos_table), 128, S, P, 128, _p(s.G2P), 128, st)
  L['qkv'].fwd(st, s.G2P, 0, 128, s.QKV, 0, 384, P)
  s.AO, s.PROB = new(P, 128), new(R * 4, S, S)
  call('dyn_train_attn', _p(s.QKV), _p(s.nvalid), R, S, _p(s.AO), _p(s.PROB), st)
  s.FCO, s.G3, s.XHAT, s.RSTD = new(P, 128), new(P, 128), new(P, 128), new(P)
  L['fc'].fwd(st, s.AO, 0, 128, s.FCO, 0, 128, P)
  call('dyn_train_layernorm', _p(s.FCO), _p(s.G2P), _p(w['ray_attention.layer_norm.weight']), _p(w['ray_attention.layer_norm.bias']), P,
       _p(s.G3), _p(s.XHAT), _p(s.RSTD), st)
  # ref_pts_fc on [attention output | PE(pts)], density and colour heads (:290-315)
  s.PPE, s.DPE = new(P, 36), new(R, 28)
  call('dyn_train_dynamic_embed', _p(pts), _p(ray_d), P, R, _p(s.PPE), _p(s.DPE), st)
  s.TP, s.Q1, s.G4 = new(P, 256), new(P, 256), new(P, 128)
  L['p0p'].fwd(st, s.PPE, 0, 36, s.TP, 0, 256, P)
  L['p0g'].fwd(st, s.G3, 0, 128, s.Q1, 0, 256, P, ELU, addend=s.TP, ld_add=256, add_div=1)
  L['p2'].fwd(st, s.Q1, 0, 256, s.G4, 0, 128, P, ELU)
  s.O1, s.SIG = new(P, 128), new(P)
  L['o0'].fwd(st, s.G4, 0, 128, s.O1, 0, 128, P, ELU)
  L['o2'].fwd(st, s.O1, 0, 128, s.SIG, 0, 1, P)
  s.DP, s.C1, s.C2, s.CL = new(R, 128), new(P, 128), new(P, 64), new(P, 4)
  L['r0d'].fwd(st, s.DPE, 0, 28, s.DP, 0, 128, R)
  L['r0g'].fwd(st, s.G4, 0, 128, s.C1, 0, 128, P, ELU, addend=s.DP, ld_add=128, add_div=S)
  L['r2'].fwd(st, s.C1, 0, 128, s.C2, 0, 64, P, ELU)
  L['r4'].fwd(st, s.C2, 0, 64, s.CL, 0, 4, P)
  raw = new(R, S, 4)
  call('dyn_train_dynamic_head', _p(s.CL), 4, _p(s.SIG), _p(s.nvalid), float(shift), P, _p(raw), st)
  s.raw = raw
  return raw, s


def _backward(s, draw):
  """draw [R,S,4] -> ({param: grad}, d rgb_feat as the [N,36] matrix dF whose columns 3..34 are the feature gradients)"""
  R, S, V, P, N, w, L = s.R, s.S, s.V, s.P, s.N, s.w, s.L
  dev = draw.device
  st = stream_of(draw)
  new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
  g = zero_grads(dict(w, **{'__qkv': s.Wqkv}))
  gqkv = g.pop('__qkv')
  draw = draw.contiguous()
  dCL, dSIG = new(P, 4), new(P)
  call('dyn_train_dynamic_head_bwd', _p(draw), _p(s.raw), _p(s.nvalid), P, _p(dCL), 4, _p(dSIG), st)
  # colour head
  dC2, dC1, dDP, dG4 = new(P, 64), new(P, 128), new(R, 128), new(P, 128)
  _act_bwd(st, dCL, 0, 4, None, 0, 4, P, 3, NONE, g['rgb_fc.4.bias'])
  if not L['r4'].bwd(st, dCL, 0, 4, s.C2, 0, 64, g['rgb_fc.4.weight'], P, dC2, 0, 64, act_y=(s.C2, 0, 64, ELU), dbias=g['rgb_fc.2.bias']):  # dC2 arrives times ELU'(C2)
    _act_bwd(st, dC2, 0, 64, None, 0, 64, P, 64, NONE, g['rgb_fc.2.bias'])
  L['r2'].bwd(st, dC2, 0, 64, s.C1, 0, 128, g['rgb_fc.2.weight'], P, dC1, 0, 128, act_y=(s.C1, 0, 128, ELU))
  _act_bwd(st, dC1, 0, 128, None, 0, 128, P, 128, NONE, g['rgb_fc.0.bias'], S, dDP, 128)
  L['r0g'].bwd(st, dC1, 0, 128, s.G4, 0, 128, g['rgb_fc.0.weight'], P, dG4, 0, 128)
  L['r0d'].bwd(st, dDP, 0, 128, s.DPE, 0, 28, g['rgb_fc.0.weight'], R)
  # density head
  dO1 = new(P, 128)
  _act_bwd(st, dSIG, 0, 1, None, 0, 1, P, 1, NONE, g['out_geometry_fc.2.bias'])
  if not L['o2'].bwd(st, dSIG, 0, 1, s.O1, 0, 128, g['out_geometry_fc.2.weight'], P, dO1, 0, 128, act_y=(s.O1, 0, 128, ELU), dbias=g['out_geometry_fc.0.bias']):
    _act_bwd(st, dO1, 0, 128, None, 0, 128, P, 128, NONE, g['out_geometry_fc.0.bias'])
  L['o0'].bwd(st, dO1, 0, 128, s.G4, 0, 128, g['out_geometry_fc.0.weight'], P, dG4, 0, 128, acc_dx=1)
  # ref_pts_fc
  dQ1, dG3 = new(P, 256), new(P, 128)
  _act_bwd(st, dG4, 0, 128, s.G4, 0, 128, P, 128, ELU, g['ref_pts_fc.2.bias'])
  if not L['p2'].bwd(st, dG4, 0, 128, s.Q1, 0, 256, g['ref_pts_fc.2.weight'], P, dQ1, 0, 256, act_y=(s.Q1, 0, 256, ELU), dbias=g['ref_pts_fc.0.bias']):
    _act_bwd(st, dQ1, 0, 256, None, 0, 256, P, 256, NONE, g['ref_pts_fc.0.bias'])
  L['p0g'].bwd(st, dQ1, 0, 256, s.G3, 0, 128, g['ref_pts_fc.0.weight'], P, dG3, 0, 128)
  s.dPPE = new(P, 36)
  L['p0p'].bwd(st, dQ1, 0, 256, s.PPE, 0, 36, g['ref_pts_fc.0.weight'], P, s.dPPE, 0, 36)  # d PE(pts): on into the points (anchor pass)
  # LayerNorm(fc(attention) + (g2 + pos)), attention, qkv
  dY = new(P, 128)
  call('dyn_train_layernorm_bwd', _p(dG3), _p(s.XHAT), _p(s.RSTD), _p(w['ray_attention.layer_norm.weight']), P, _p(dY),
       _p(g['ray_attention.layer_norm.weight']), _p(g['ray_attention.layer_norm.bias']), st)
  dAO, dQKV, dSC = new(P, 128), new(P, 384), new(R * 4, S, S)
  L['fc'].bwd(st, dY, 0, 128, s.AO, 0, 128, g['ray_attention.fc.weight'], P, dAO, 0, 128)
  call('dyn_train_attn_bwd', _p(s.QKV), _p(s.nvalid), R, S, _p(s.PROB), _p(dAO), _p(dSC), _p(dQKV), st)
  L['qkv'].bwd(st, dQKV, 0, 384, s.G2P, 0, 128, gqkv, P, dY, 0, 128, acc_dx=1)
  g['ray_attention.w_qs.weight'], g['ray_attention.w_ks.weight'], g['ray_attention.w_vs.weight'] = gqkv[0:128], gqkv[128:256], gqkv[256:384]
  # geometry_fc (the positional table is a constant: d g2 = d (g2 + pos))
  dGH1, dG0 = new(P, 256), new(P, 260)
  _act_bwd(st, dY, 0, 128, s.G2, 0, 128, P, 128, ELU, g['geometry_fc.2.bias'])
  if not L['g2'].bwd(st, dY, 0, 128, s.GH1, 0, 256, g['geometry_fc.2.weight'], P, dGH1, 0, 256, act_y=(s.GH1, 0, 256, ELU), dbias=g['geometry_fc.0.bias']):
    _act_bwd(st, dGH1, 0, 256, None, 0, 256, P, 256, NONE, g['geometry_fc.0.bias'])
  L['g0'].bwd(st, dGH1, 0, 256, s.G0, 0, 260, g['geometry_fc.0.weight'], P, dG0, 0, 260)
  dX, dw2, dVL = new(N, 128), new(N), new(N)  # dX: gradient of x2, then of x1
  call('dyn_train_meanvar_bwd', _p(s.X2), 128, _p(s.w2), P, V, 128, _p(s.G0), _p(dG0), _p(dG0, 128), 260, _p(dX), 128, 0, _p(dw2), 0, st)
  call('dyn_train_view_weights_bwd', 1, _p(s.VL), 1, _p(s.M), None, P, V, _p(s.w2), _p(dw2), None, 0, _p(s.VIS), 1, _p(dG0, 256), 260,
       _p(dVL), 1, None, st)
  dH4, dXS = new(N, 128), new(N, 128)
  _act_bwd(st, dVL, 0, 1, None, 0, 1, N, 1, NONE, g['vis_fc2.2.bias'])
  if not L['w2'].bwd(st, dVL, 0, 1, s.H4, 0, 128, g['vis_fc2.2.weight'], N, dH4, 0, 128, act_y=(s.H4, 0, 128, ELU), dbias=g['vis_fc2.0.bias']):
    _act_bwd(st, dH4, 0, 128, None, 0, 128, N, 128, NONE, g['vis_fc2.0.bias'])
  s.drop('H4')
  L['w0'].bwd(st, dH4, 0, 128, s.X2, 0, 128, g['vis_fc2.0.weight'], N, dXS, 0, 128, x_scale=s.vis0)  # the layer ran on x * vis
  del dH4
  dXV, scratch = new(N, 132), new(N)
  # the row-scale backward of x * vis, the split's backward and vis_fc.2's ELU in one pass
  _split_act_bwd(st, dX, 128, None, s.XV, s.M, N, dXV, g['vis_fc.2.bias'], dXS=dXS, X2=s.X2, ldx2=128, vis0=s.vis0)
  del dXS
  s.drop('X2')
  dH3, dXW = new(N, 128), new(N, 128)
  if not L['v2'].bwd(st, dXV, 0, 132, s.H3, 0, 128, g['vis_fc.2.weight'], N, dH3, 0, 128, act_y=(s.H3, 0, 128, ELU), dbias=g['vis_fc.0.bias']):
    _act_bwd(st, dH3, 0, 128, None, 0, 128, N, 128, NONE, g['vis_fc.0.bias'])
  del dXV
  s.drop('XV', 'H3')
  L['v0'].bwd(st, dH3, 0, 128, s.X1, 0, 128, g['vis_fc.0.weight'], N, dXW, 0, 128, x_scale=s.w1)  # the layer ran on x * weight
  del dH3
  # d x1 is complete with this term: its row-scale backward and base_fc.2's ELU in one pass (w1 = mask / sum: no parameter behind it)
  _rowscale_act_bwd(st, dXW, 128, s.X1, 128, s.w1, N, dX, 128, scratch, 0, ELU, g['base_fc.2.bias'])
  # base_fc
  dH2, dPP1, dF, dG1 = new(N, 256), new(P, 256), new(N, 36), new(P, 72)
  H2 = s.H2
  if H2 is None:  # RECOMPUTE_HIDDEN: the hidden layer of base_fc again (the same launch as in the forward pass: bit-identical)
    H2 = new(N, 256)
    L['b0f'].fwd(st, s.F, 0, 36, H2, 0, 256, N, ELU, addend=s.PP1, ld_add=256, add_div=V)
  L['b2'].bwd(st, dX, 0, 128, H2, 0, 256, g['base_fc.2.weight'], N, dH2, 0, 256, act_y=(H2, 0, 256, ELU))
  del dX, dXW, H2
  s.drop('X1', 'H2')
  _act_bwd(st, dH2, 0, 256, None, 0, 256, N, 256, NONE, g['base_fc.0.bias'], V, dPP1, 256)
  L['b0f'].bwd(st, dH2, 0, 256, s.F, 0, 36, g['base_fc.0.weight'], N, dF, 0, 36)
  del dH2
  L['b0g'].bwd(st, dPP1, 0, 256, s.G1, 0, 72, g['base_fc.0.weight'], P, dG1, 0, 72)
  _untag(dF)
  call('dyn_train_meanvar_bwd', _p(s.F), 36, _p(s.w1), P, V, 35, _p(s.G1), _p(dG1), _p(dG1, 35), 72, _p(dF), 36, 1, _p(scratch), 0, st)
  # time feature: its gradient is the column sum of d(rgb_feat + feature) over all rows, then back through ray_dir_fc
  dDIRF, dDH1 = torch.zeros((1, 36), dtype=torch.float32, device=dev), new(1, 256)
  _act_bwd(st, dF, 0, 36, None, 0, 36, N, 35, NONE, dDIRF)
  _act_bwd(st, dDIRF, 0, 36, s.DIRF, 0, 36, 1, 35, ELU, g['ray_dir_fc.2.bias'])
  L['rd2'].bwd(st, dDIRF, 0, 36, s.DH1, 0, 256, g['ray_dir_fc.2.weight'], 1, dDH1, 0, 256)
  _act_bwd(st, dDH1, 0, 256, s.DH1, 0, 256, 1, 256, ELU, g['ray_dir_fc.0.bias'])
  L['rd0'].bwd(st, dDH1, 0, 256, s.TPE, 0, 24, g['ray_dir_fc.0.weight'], 1)
  return g, dF


def _posenc(S, dev):
  """DynibarDynamic.posenc(d_hid=128, n_samples=S) (mlp_network.py:218-234): a constant table, built once per (S, device)"""
  import numpy as np
  key = (S, str(dev))
  tab = _posenc.cache.get(key)
  if tab is None:
    a = np.array([[pos / np.power(10000, 2 * (j // 2) / 128) for j in range(128)] for pos in range(S)])
    a[:, 0::2] = np.sin(a[:, 0::2])
    a[:, 1::2] = np.cos(a[:, 1::2])
    tab = torch.from_numpy(a).float().to(dev).contiguous()
    _posenc.cache[key] = tab
  return tab


_posenc.cache = {}


class DynamicNetFunction(torch.autograd.Function):
  """raw_dy [R,S,4] = DynibarDynamic(features gathered at the motion-displaced points) with gradients to rgb_feat (through the gather:
  feature maps and displaced points), to pts_xyz (the Fourier features of ref_pts_fc; the anchor pass's points depend on the motion
  coefficients) and to the parameters."""

  @staticmethod
  def forward(ctx, rgb_feat, pts, meta, *param_tensors):
    names, shift, ray_d, mask, time = meta
    w = {n: (t.detach() if t.dtype == torch.float32 and t.is_contiguous() else t.detach().float().contiguous()) for n, t in zip(names, param_tensors)}
    S = rgb_feat.shape[1]
    raw, step = _forward(w, shift, _posenc(S, rgb_feat.device), ray_d, pts.detach(), rgb_feat.detach(), mask, time)
    step.pts = pts.detach() if (pts.dtype == torch.float32 and pts.is_contiguous()) else pts.detach().float().contiguous()
    ctx.step, ctx.names, ctx.fshape, ctx.pshape = step, names, tuple(rgb_feat.shape), tuple(pts.shape)
    return raw

  @staticmethod
  def backward(ctx, draw):
    if ctx.step is None:
      raise RuntimeError('DynamicNetFunction: the saved activations were released by the first backward pass; call the renderer again instead of backward(retain_graph=True)')
    s = ctx.step
    g, dF = _backward(s, draw.float())
    gr = dF[:, :35].reshape(ctx.fshape) if ctx.needs_input_grad[0] else None
    gpts = None
    if ctx.needs_input_grad[1]:
      from .train_motion import OCTAVES5, _freqs
      fa, fp = _freqs(OCTAVES5)
      gpts = torch.empty(ctx.pshape, dtype=torch.float32, device=draw.device)
      call('dyn_train_embed_bwd', _p(s.pts), 3, s.P, 3, fp, 5, _p(s.dPPE), 36, _p(gpts), 3, 0, stream_of(draw))
    ctx.step = None
    gp = tuple(g[n] if ctx.needs_input_grad[3 + i] else None for i, n in enumerate(ctx.names))
    return (gr, gpts, None) + gp


def wants_grad(net, featmaps):
  from .train_static import wants_grad as wg
  return wg(net, featmaps)


def dynamic_raw(net, shift, rgb_feat, ray_d, pts, mask, time):
  """raw_dy with an autograd graph.  net: the reference's DynibarDynamic (nn.Module, DataParallel-wrapped or not) or a dict of parameter
  tensors; rgb_feat: the features gathered at the motion-displaced points (train_motion.gather); pts: the points handed to the net
  (pts_ref, or pts_anchor in the cross-time pass)."""
  names, tensors = _param_list(net)
  meta = (names, float(shift), ray_d, mask, time)
  return DynamicNetFunction.apply(rgb_feat, pts, meta, *tensors)


class CompositeDualFunction(torch.autograd.Function):
  """raw2outputs (render_ray.py:214-330): rgb, rgb_static, rgb_dy, depth, weights_dy, weights_st, weights differentiable w.r.t. both raw
  inputs; alpha_dy, alpha and the ray mask are forward values."""

  @staticmethod
  def forward(ctx, raw_dy, raw_st, z_vals, pm_dy, pm_st):
    out = ops.composite(raw_dy.detach(), z_vals, pm_dy, raw_st.detach(), pm_st)
    ctx.save_for_backward(raw_dy.detach().float().contiguous(), raw_st.detach().float().contiguous(), out['z_vals'])
    ctx.mark_non_differentiable(out['mask'], out['alpha'], out['alpha_dy'])
    return (out['rgb'], out['rgb_static'], out['rgb_dy'], out['depth'], out['weights_dy'], out['weights_st'], out['weights'], out['mask'], out['alpha'],
            out['alpha_dy'])

  @staticmethod
  def backward(ctx, g_rgb, g_rgb_st, g_rgb_dy, g_depth, g_wd, g_ws, g_w, _m, _a, _ad):
    raw_dy, raw_st, z_vals = ctx.saved_tensors
    R, S = z_vals.shape
    keep = [t.float().contiguous() if t is not None else None for t in (g_rgb, g_rgb_st, g_rgb_dy, g_depth, g_wd, g_ws, g_w)]
    d_dy, d_st = torch.empty_like(raw_dy), torch.empty_like(raw_st)
    call('dyn_train_composite2_bwd', _p(raw_dy), _p(raw_st), _p(z_vals), *[None if t is None else _p(t) for t in keep], R, S, _p(d_dy), _p(d_st),
         stream_of(raw_dy))
    return d_dy, d_st, None, None, None


def composite_dual(raw_dy, raw_st, z_vals, pm_dy, pm_st):
  """-> dict with the reference's key set and order (render_ray.py:316-328)"""
  from collections import OrderedDict
  rgb, rgb_st, rgb_dy, depth, w_dy, w_st, wts, m, alpha, a_dy = CompositeDualFunction.apply(raw_dy, raw_st, z_vals, pm_dy, pm_st)
  return OrderedDict([('rgb', rgb), ('rgb_static', rgb_st), ('rgb_dy', rgb_dy), ('depth', depth), ('alpha_dy', a_dy), ('weights_dy', w_dy),
                      ('weights_st', w_st), ('alpha', alpha), ('weights', wts), ('mask', m > 0), ('z_vals', z_vals)])
