"""Seeded synthetic scenes + network weights for tests, smoke and bench.

Everything here is numpy with ``default_rng`` (PCG64 streams are stable across
numpy versions), so the golden-vector generator that runs against the real
reference in the build container and the tests that run on the GPU box
regenerate bit-identical inputs without shipping them.

Shapes follow SURVEY.md section 8d: camera vectors are the reference's 34-float
layout ``[h, w, K(4x4 row-major), c2w(4x4 row-major)]``
(reference ibrnet/sample_ray.py:11-16).
"""
from __future__ import annotations

import numpy as np


def make_camera(h, w, focal, c2w):
  K = np.eye(4, dtype=np.float32)
  K[0, 0] = K[1, 1] = focal
  K[0, 2] = (w - 1) * 0.5
  K[1, 2] = (h - 1) * 0.5
  return np.concatenate(
      [np.array([h, w], np.float32), K.reshape(-1), c2w.astype(np.float32).reshape(-1)]
  )


def _rot_xyz(rx, ry, rz):
  cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
  Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
  Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
  Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
  return Rz @ Ry @ Rx


def make_pose(rng, t_scale, r_scale):
  c2w = np.eye(4)
  c2w[:3, :3] = _rot_xyz(*(rng.uniform(-r_scale, r_scale, 3)))
  c2w[:3, 3] = rng.uniform(-t_scale, t_scale, 3) * np.array([1.0, 0.3, 0.15])
  return c2w


def smooth_field(rng, n, h, w, c, n_waves=6):
  """Low-frequency image stack [n,h,w,c]: well-conditioned for bilinear parity."""
  yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing='ij')
  out = np.zeros((n, h, w, c))
  for _ in range(n_waves):
    fx = rng.uniform(0.5, 6.0, (n, 1, 1, c))
    fy = rng.uniform(0.5, 6.0, (n, 1, 1, c))
    ph = rng.uniform(0, 2 * np.pi, (n, 1, 1, c))
    am = rng.uniform(0.2, 1.0, (n, 1, 1, c))
    out += am * np.sin(2 * np.pi * (fx * xx[None, ..., None] + fy * yy[None, ..., None]) + ph)
  return out / n_waves


def make_scene(seed=0, H=288, W=512, V=8, F=32, feat_div=4, smooth=False,
               near=1.0, far=20.0, focal=None, t_scale=0.4, r_scale=0.05,
               n_static=None, tag=0):
  """One target view + V dynamic-branch source views (+ n_static static views).

  Returns a dict of float32 numpy arrays shaped like the reference's per-view
  ``data`` dict after batching (leading 1): camera [1,34], src_rgbs [1,V,H,W,3],
  src_cameras [1,V,34], static_src_rgbs/static_src_cameras, depth_range [1,2],
  plus feature maps ``featmaps [V,F,H/feat_div,W/feat_div]`` (NCHW, as the
  reference's ResNet emits them, ibrnet/feature_network.py:302-311).
  """
  rng = np.random.default_rng([seed, tag, 17])
  focal = focal if focal is not None else 0.78 * W
  tgt_c2w = make_pose(rng, 0.05, 0.01)
  camera = make_camera(H, W, focal, tgt_c2w)[None]
  Hf, Wf = H // feat_div, W // feat_div

  def views(n):
    cams = np.stack([make_camera(H, W, focal * rng.uniform(0.95, 1.05), make_pose(rng, t_scale, r_scale))
                     for _ in range(n)])[None]
    if smooth:
      rgbs = 0.5 + 0.5 * smooth_field(rng, n, H, W, 3)
      feats = smooth_field(rng, n, Hf, Wf, F).transpose(0, 3, 1, 2) * 2.0
    else:
      rgbs = rng.random((n, H, W, 3))
      feats = rng.standard_normal((n, F, Hf, Wf))
    return cams.astype(np.float32), rgbs[None].astype(np.float32), np.ascontiguousarray(feats, dtype=np.float32)

  src_cameras, src_rgbs, featmaps = views(V)
  out = dict(camera=camera.astype(np.float32), src_cameras=src_cameras, src_rgbs=src_rgbs,
             featmaps=featmaps, depth_range=np.array([[near, far]], np.float32))
  n_static = V if n_static is None else n_static
  sc, sr, sf = views(n_static)
  out.update(static_src_cameras=sc, static_src_rgbs=sr, static_featmaps=sf)
  return out


def pixel_rays(camera, pix_idx):
  """Rays through pixel indices, the reference's formula
  d = c2w[:3,:3] . inv(K[:3,:3]) . [u,v,1], o = c2w[:3,3]
  (reference ibrnet/sample_ray.py:143-163)."""
  h, w = int(camera[0, 0]), int(camera[0, 1])
  K = camera[0, 2:18].reshape(4, 4)
  c2w = camera[0, 18:34].reshape(4, 4)
  u = (pix_idx % w).astype(np.float32)
  v = (pix_idx // w).astype(np.float32)
  pix = np.stack([u, v, np.ones_like(u)], 0)
  d = (c2w[:3, :3] @ np.linalg.inv(K[:3, :3]) @ pix).T.astype(np.float32)
  o = np.broadcast_to(c2w[:3, 3], d.shape).astype(np.float32).copy()
  uv = np.stack([u, v], -1)
  return o, d, uv


def sample_pixels(seed, H, W, R):
  rng = np.random.default_rng([seed, 991])
  return np.sort(rng.choice(H * W, size=R, replace=False))


# ----------------------------------------------------------------------------
# Network weights.  Layer tables restate the constructor shapes of the
# reference modules (ibrnet/mlp_network.py:129-234 DynibarDynamic, :319-421
# DynibarStatic, :558-603 MotionMLP, :56-78 MultiHeadAttention) as
# (state_dict key, out_features, in_features, has_bias).
# ----------------------------------------------------------------------------

def static_layer_table(F=32):
  C = F + 3
  return [
      ('ray_dir_fc.0', 256, 4 + 33 + 66, True), ('ray_dir_fc.2', C, 256, True),
      ('ref_feature_fc.0', C, 66, True),
      ('base_fc.0', 256, C * 6, True), ('base_fc.2', 128, 256, True),
      ('vis_fc.0', 128, 128, True), ('vis_fc.2', 129, 128, True),
      ('vis_fc2.0', 128, 128, True), ('vis_fc2.2', 1, 128, True),
      ('geometry_fc.0', 256, 257, True), ('geometry_fc.2', 128, 256, True),
      ('ray_attention.w_qs', 128, 128, False), ('ray_attention.w_ks', 128, 128, False),
      ('ray_attention.w_vs', 128, 128, False), ('ray_attention.fc', 128, 128, False),
      ('out_geometry_fc.0', 128, 128, True), ('out_geometry_fc.2', 1, 128, True),
      ('rgb_fc.0', 128, 128 * 2 + 1 + 4, True), ('rgb_fc.2', 64, 128, True), ('rgb_fc.4', 1, 64, True),
  ]


def dynamic_layer_table(F=32):
  C = F + 3
  return [
      ('ray_dir_fc.0', 256, 21, True), ('ray_dir_fc.2', C, 256, True),
      ('base_fc.0', 256, C * 3, True), ('base_fc.2', 128, 256, True),
      ('vis_fc.0', 128, 128, True), ('vis_fc.2', 129, 128, True),
      ('vis_fc2.0', 128, 128, True), ('vis_fc2.2', 1, 128, True),
      ('geometry_fc.0', 256, 257, True), ('geometry_fc.2', 128, 256, True),
      ('ray_attention.w_qs', 128, 128, False), ('ray_attention.w_ks', 128, 128, False),
      ('ray_attention.w_vs', 128, 128, False), ('ray_attention.fc', 128, 128, False),
      ('ref_pts_fc.0', 256, 33 + 128, True), ('ref_pts_fc.2', 128, 256, True),
      ('out_geometry_fc.0', 128, 128, True), ('out_geometry_fc.2', 1, 128, True),
      ('rgb_fc.0', 128, 128 + 27, True), ('rgb_fc.2', 64, 128, True), ('rgb_fc.4', 3, 64, True),
  ]


def motion_layer_table(num_basis=6, W=256, D=8, input_ch=4, num_freqs=16, skips=(4,)):
  cin = input_ch + input_ch * num_freqs * 2
  tab = [('pts_linears.0', W, cin, True)]
  for i in range(D - 1):
    tab.append((f'pts_linears.{i + 1}', W, W + cin if i in skips else W, True))
  tab.append(('coeff_linear', num_basis * 3, W, True))
  return tab


ENCODER_TENSORS = (
    ('conv1.weight', (64, 3, 7, 7)), ('bn1.weight', (64,)), ('bn1.bias', (64,)),
    ('layer1.0.conv1.weight', (64, 64, 3, 3)), ('layer1.0.bn1.weight', (64,)), ('layer1.0.bn1.bias', (64,)),
    ('layer1.0.conv2.weight', (64, 64, 3, 3)), ('layer1.0.bn2.weight', (64,)), ('layer1.0.bn2.bias', (64,)),
    ('layer1.0.downsample.0.weight', (64, 64, 1, 1)), ('layer1.0.downsample.1.weight', (64,)), ('layer1.0.downsample.1.bias', (64,)),
    ('layer1.1.conv1.weight', (64, 64, 3, 3)), ('layer1.1.bn1.weight', (64,)), ('layer1.1.bn1.bias', (64,)),
    ('layer1.1.conv2.weight', (64, 64, 3, 3)), ('layer1.1.bn2.weight', (64,)), ('layer1.1.bn2.bias', (64,)),
    ('layer1.2.conv1.weight', (64, 64, 3, 3)), ('layer1.2.bn1.weight', (64,)), ('layer1.2.bn1.bias', (64,)),
    ('layer1.2.conv2.weight', (64, 64, 3, 3)), ('layer1.2.bn2.weight', (64,)), ('layer1.2.bn2.bias', (64,)),
    ('out_conv.weight', (64, 64, 1, 1)), ('out_conv.bias', (64,)),
)


def make_encoder_weights(seed=0):
  """Seeded state dict (numpy float32) of the executed part of the reference's ResNet (feature_network.py:179-311; names and shapes
  of ENCODER_TENSORS): Kaiming-uniform-like convolutions, InstanceNorm gains around 1 and non-zero shifts, non-zero output bias."""
  rng = np.random.default_rng([seed, 7])
  sd = {}
  for name, shape in ENCODER_TENSORS:
    if len(shape) == 4:
      bound = np.sqrt(3.0 / (shape[1] * shape[2] * shape[3]))
      sd[name] = rng.uniform(-bound, bound, shape).astype(np.float32)
    elif name.endswith('.weight'):
      sd[name] = (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
    else:
      sd[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
  return sd


def make_weights(kind, seed=0, F=32, num_basis=6, gain=1.0, bias=0.1, head_gain=1.0, ln_gain=1.0):
  """Seeded state-dict (numpy float32) for 'static' | 'dynamic' | 'motion'.

  ``gain`` scales every weight matrix, ``bias`` the bias range, ``head_gain`` the density head, ``ln_gain`` the LayerNorm gains:
  the "trained-scale" test weights use them to reach activations of tens and density logits of +-30..50.
  Kaiming-uniform-like scale so activations stay O(1) through ~20 layers; biases
  are non-zero and the MotionMLP head is non-zero (the reference zero-inits it,
  mlp_network.py:602-603, which would make scene motion a no-op in tests).
  LayerNorm affine parameters are perturbed away from (1,0) so a missing
  gamma/beta is caught.
  """
  table = {'static': static_layer_table(F), 'dynamic': dynamic_layer_table(F),
           'motion': motion_layer_table(num_basis)}[kind]
  rng = np.random.default_rng([seed, {'static': 1, 'dynamic': 2, 'motion': 3}[kind]])
  sd = {}
  for name, nout, nin, has_bias in table:
    bound = gain * np.sqrt(3.0 / nin)
    sd[name + '.weight'] = rng.uniform(-bound, bound, (nout, nin)).astype(np.float32)
    if has_bias:
      sd[name + '.bias'] = rng.uniform(-bias, bias, (nout,)).astype(np.float32)
  if kind in ('static', 'dynamic'):
    sd['ray_attention.layer_norm.weight'] = (ln_gain * (1.0 + 0.1 * rng.standard_normal(128))).astype(np.float32)
    sd['out_geometry_fc.2.weight'] *= np.float32(head_gain)  # the density head: trained nets reach logits of +-30..50
    sd['ray_attention.layer_norm.bias'] = (0.05 * rng.standard_normal(128)).astype(np.float32)
  if kind == 'static':
    sd['s'] = np.array(0.2, np.float32)  # anti-alias pooling temperature (mlp_network.py:331)
  if kind == 'motion':
    sd['coeff_linear.weight'] *= 0.05
    sd['coeff_linear.bias'] *= 0.05
  return sd
