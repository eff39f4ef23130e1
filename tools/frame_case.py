"""Synthetic Nvidia-Balloon1-shaped full frame (BASELINE configs[2]): 288x512 target rays, 64 coarse + 64 fine samples, 7 dynamic + 11 static
source views, chunk 8192 (eval_nvidia.py:360-378, configs_nvidia/eval_balloon1_long.txt).  Shared by bench.py and tools/framebench.py."""
import types

import numpy as np
import torch

from dynibar_amd import projection, render_image, sample_ray, synthetic as syn

NUM_FRAMES, NUM_BASIS = 24, 6


def dct_basis(K, T):
  """model.py:18-30 init_dct_basis."""
  b = np.zeros((T, K), np.float32)
  for t in range(T):
    for k in range(1, K + 1):
      b[t, k - 1] = np.sqrt(2.0 / T) * np.cos(np.pi / (2.0 * T) * (2 * t + 1) * k)
  return torch.from_numpy(b)


class FrameCase:
  def __init__(self, dev, H=288, W=512, vdy=7, vst=11, chunk=8192):
    self.dev, self.H, self.W, self.chunk, self.vdy, self.vst = dev, H, W, chunk, vdy, vst
    sc = syn.make_scene(seed=0, H=H, W=W, V=vdy, n_static=vst)
    fine = syn.make_scene(seed=0, H=H, W=W, V=vdy, n_static=vst, tag=1)
    T = lambda x: torch.from_numpy(x).to(dev)
    self.data = dict(camera=torch.from_numpy(sc['camera']), rgb_path='x', depth_range=torch.from_numpy(sc['depth_range']),
                     src_rgbs=torch.from_numpy(sc['src_rgbs']), src_cameras=torch.from_numpy(sc['src_cameras']),
                     static_src_rgbs=torch.from_numpy(sc['static_src_rgbs']), static_src_cameras=torch.from_numpy(sc['static_src_cameras']))
    self.model = types.SimpleNamespace(net_coarse_st=syn.make_weights('static', 0), net_coarse_dy=syn.make_weights('dynamic', 0),
                                       net_fine_st=syn.make_weights('static', 100), net_fine_dy=syn.make_weights('dynamic', 100),
                                       motion_mlp=syn.make_weights('motion', 0), motion_mlp_fine=syn.make_weights('motion', 100),
                                       trajectory_basis=dct_basis(NUM_BASIS, NUM_FRAMES).to(dev), trajectory_basis_fine=dct_basis(NUM_BASIS, NUM_FRAMES).to(dev))
    self.args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
    self.cfeat = (T(sc['featmaps']), None, T(sc['static_featmaps']))
    self.ffeat = (T(fine['featmaps']), None, T(fine['static_featmaps']))
    self.proj = projection.Projector(dev)
    self.fidx, self.temb, self.toff = 11, torch.tensor([11 / 24.0], device=dev), [-3, -2, -1, 0, 1, 2, 3][:vdy]

  def sampler(self):
    smp = sample_ray.RaySamplerSingleImage(self.data, self.dev)
    return smp, smp.get_all()

  def render(self, smp, rb):
    """One render_single_image_nvi call; under torch.distributed every rank renders its ray tile and gets the full frame back."""
    return render_image.render_single_image_nvi((self.fidx, None), (self.temb, None), (self.toff, None), smp, rb, self.model, self.proj, self.chunk, 64,
                                                self.args, inv_uniform=True, N_importance=64, det=True, coarse_featmaps=self.cfeat,
                                                fine_featmaps=self.ffeat, is_train=False)
