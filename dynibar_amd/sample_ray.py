"""Drop-in for the reference's ``ibrnet/sample_ray.py``: ``parse_camera`` and ``RaySamplerSingleImage`` with the same
constructor / attributes / ``get_all`` / ``random_sample`` contract (reference sample_ray.py:11-331).

Differences in mechanism, not in contract: the H*W rays of the target view are generated on the device by ``k_image_rays``
(the reference builds them with numpy + a CPU bmm and copies 3.5 MB per view); pixel selection for training keeps the
reference's module-level ``np.random.RandomState(234)`` stream on the host so the selected indices are bit-identical.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_of

rng = np.random.RandomState(234)  # same seed and same call order as the reference (sample_ray.py:8, :251-256)


def parse_camera(params):
  H = params[:, 0]
  W = params[:, 1]
  intrinsics = params[:, 2:18].reshape((-1, 4, 4))
  c2w = params[:, 18:34].reshape((-1, 4, 4))
  return W, H, intrinsics, c2w


def image_rays(camera, H, W, render_stride, device):
  """k_image_rays -> rays_o, rays_d [(H/stride)*(W/stride), 3] on ``device`` (reference sample_ray.py:143-163)."""
  cam = camera.reshape(-1)[:34].to(device=device, dtype=torch.float32).contiguous()
  Hs, Ws = (H + render_stride - 1) // render_stride, (W + render_stride - 1) // render_stride
  rays_o = torch.empty((Hs * Ws, 3), dtype=torch.float32, device=device)
  rays_d = torch.empty_like(rays_o)
  call('dyn_image_rays', ptr(cam), H, W, render_stride, ptr(rays_o), ptr(rays_d), stream_of(rays_o))
  return rays_o, rays_d


def _opt(data, key):
  return data[key] if key in data.keys() else None


class RaySamplerSingleImage(object):
  """Rays and per-pixel supervision of one target view (reference sample_ray.py:19-331)."""

  def __init__(self, data, device, resize_factor=1, render_stride=1):
    super().__init__()
    self.render_stride = render_stride
    self.rgb = _opt(data, 'rgb')
    self.disp = _opt(data, 'disp')
    self.motion_mask = _opt(data, 'motion_mask')
    self.static_mask = _opt(data, 'static_mask')
    self.flows = data['flows'].squeeze(0) if 'flows' in data.keys() else None
    self.masks = data['masks'].squeeze(0) if 'masks' in data.keys() else None
    self.camera = data['camera']
    self.render_camera = _opt(data, 'render_camera')
    self.anchor_camera = _opt(data, 'anchor_camera')
    self.rgb_path = data['rgb_path'] if 'rgb_path' in data.keys() else None
    self.depth_range = data['depth_range']
    self.device = device
    W, H, self.intrinsics, self.c2w_mat = parse_camera(self.camera)
    self.batch_size = len(self.camera)
    assert self.batch_size == 1, 'only support batch_size=1 for now'
    self.H = int(H[0])
    self.W = int(W[0])
    # pixel grid (x, y), un-normalised, row-major: what kornia.create_meshgrid(H, W, False) returns (sample_ray.py:83-87)
    ys, xs = torch.meshgrid(torch.arange(self.H, dtype=torch.float32, device=device),
                            torch.arange(self.W, dtype=torch.float32, device=device), indexing='ij')
    self.uv_grid = torch.stack([xs, ys], dim=-1).reshape(-1, 2)
    self.rays_o, self.rays_d = image_rays(self.camera, self.H, self.W, render_stride, device)
    if self.rgb is not None:
      self.rgb = self.rgb.reshape(-1, 3)
    if self.disp is not None:
      self.disp = self.disp.reshape(-1, 1)
    if self.motion_mask is not None:
      self.motion_mask = self.motion_mask.reshape(-1, 1)
    if self.static_mask is not None:
      self.static_mask = self.static_mask.reshape(-1, 1)
    if self.flows is not None:
      self.flows = self.flows.reshape(self.flows.shape[0], -1, 2)
      self.masks = self.masks.reshape(self.masks.shape[0], -1, 1)
    self.src_rgbs = _opt(data, 'src_rgbs')
    self.src_cameras = _opt(data, 'src_cameras')
    self.anchor_src_rgbs = _opt(data, 'anchor_src_rgbs')
    self.anchor_src_cameras = _opt(data, 'anchor_src_cameras')
    self.static_src_rgbs = _opt(data, 'static_src_rgbs')
    self.static_src_cameras = _opt(data, 'static_src_cameras')
    self.static_src_masks = _opt(data, 'static_src_masks')

  def _dev(self, t, squeeze=False):
    if t is None:
      return None
    t = t.to(self.device)
    return t.squeeze() if squeeze else t

  def get_all(self):
    """All rays of the view plus the per-view tensors, on the device (reference sample_ray.py:165-235)."""
    return {
        'ray_o': self.rays_o, 'ray_d': self.rays_d, 'depth_range': self._dev(self.depth_range), 'camera': self._dev(self.camera),
        'render_camera': self._dev(self.render_camera), 'anchor_camera': self._dev(self.anchor_camera), 'rgb': self._dev(self.rgb),
        'src_rgbs': self._dev(self.src_rgbs), 'src_cameras': self._dev(self.src_cameras),
        'anchor_src_rgbs': self._dev(self.anchor_src_rgbs), 'anchor_src_cameras': self._dev(self.anchor_src_cameras),
        'static_src_rgbs': self._dev(self.static_src_rgbs), 'static_src_cameras': self._dev(self.static_src_cameras),
        'static_src_masks': self._dev(self.static_src_masks), 'disp': self._dev(self.disp, True),
        'motion_mask': self._dev(self.motion_mask, True), 'static_mask': self._dev(self.static_mask, True), 'uv_grid': self.uv_grid,
        'flows': self._dev(self.flows), 'masks': self._dev(self.masks),
    }

  def sample_random_pixel(self, N_rand, sample_mode, center_ratio=0.8):
    """Host-side pixel selection, same RNG stream as the reference (sample_ray.py:237-260)."""
    if sample_mode == 'center':
      border_H = int(self.H * (1 - center_ratio) / 2.0)
      border_W = int(self.W * (1 - center_ratio) / 2.0)
      u, v = np.meshgrid(np.arange(border_H, self.H - border_H), np.arange(border_W, self.W - border_W))
      u = u.reshape(-1)
      v = v.reshape(-1)
      select_inds = rng.choice(u.shape[0], size=(N_rand,), replace=False)
      select_inds = v[select_inds] + self.W * u[select_inds]
    elif sample_mode == 'uniform':
      select_inds = rng.choice(self.H * self.W, size=(N_rand,), replace=False)
    else:
      raise NotImplementedError
    return select_inds

  def random_sample(self, N_rand, sample_mode, center_ratio=0.8):
    """Random pixel batch with its supervision (reference sample_ray.py:262-331)."""
    select_inds = self.sample_random_pixel(N_rand, sample_mode, center_ratio)
    if self.rgb is None:
      raise NotImplementedError
    idx = torch.from_numpy(np.asarray(select_inds)).long()
    didx = idx.to(self.device)
    return {
        'ray_o': self.rays_o[didx], 'ray_d': self.rays_d[didx], 'camera': self._dev(self.camera),
        'anchor_camera': self._dev(self.anchor_camera), 'depth_range': self._dev(self.depth_range), 'rgb': self._dev(self.rgb[idx]),
        'disp': self._dev(self.disp[idx].squeeze()), 'motion_mask': self._dev(self.motion_mask[idx].squeeze()),
        'static_mask': self._dev(self.static_mask[idx].squeeze()), 'uv_grid': self.uv_grid[didx],
        'flows': self._dev(self.flows[:, idx, :]), 'masks': self._dev(self.masks[:, idx, :]), 'src_rgbs': self._dev(self.src_rgbs),
        'src_cameras': self._dev(self.src_cameras), 'static_src_rgbs': self._dev(self.static_src_rgbs),
        'static_src_cameras': self._dev(self.static_src_cameras), 'static_src_masks': self._dev(self.static_src_masks),
        'anchor_src_rgbs': self._dev(self.anchor_src_rgbs), 'anchor_src_cameras': self._dev(self.anchor_src_cameras),
        'selected_inds': select_inds,
    }
