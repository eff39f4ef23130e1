"""BASELINE.json configs[3] and configs[4] as timed cases (shared by bench.py's extra legs):
  MonoFrame   -- kid-running monocular full frame: render_single_image_mono, 288x512 rays, 64 samples, 7 + 3 dynamic (scene-flow-warped)
                 and 15 static views, anti_alias_pooling 0 / mask_rgb 1 (configs/test_kid-running.txt, render_monocular_bt.py:309-340);
  StressChunk -- one 8192-ray chunk of render_rays_mv at 128 + 128 samples with 16 dynamic and 16 static views."""
import types

import torch

from dynibar_amd import projection, render_image, render_ray, sample_ray, synthetic as syn
from frame_case import NUM_BASIS, NUM_FRAMES, dct_basis


def _model(dev, fine=False):
  W = lambda kind, seed, **kw: syn.make_weights(kind, seed, **kw)
  m = types.SimpleNamespace(net_coarse_st=W('static', 0), net_coarse_dy=W('dynamic', 0), motion_mlp=W('motion', 0, num_basis=NUM_BASIS),
                            trajectory_basis=dct_basis(NUM_BASIS, NUM_FRAMES).to(dev))
  if fine:
    m.net_fine_st, m.net_fine_dy, m.motion_mlp_fine = W('static', 100), W('dynamic', 100), W('motion', 100, num_basis=NUM_BASIS)
    m.trajectory_basis_fine = dct_basis(NUM_BASIS, NUM_FRAMES).to(dev)
  return m


class MonoFrame:
  def __init__(self, dev, H=288, W=512, vdy=10, vst=15, num_vv=3, chunk=8192):
    self.dev, self.chunk, self.num_vv = dev, chunk, num_vv
    sc = syn.make_scene(seed=31, H=H, W=W, V=vdy, n_static=vst)
    T = lambda x: torch.from_numpy(x).to(dev)
    self.data = dict(camera=torch.from_numpy(sc['camera']), rgb_path='x', depth_range=torch.from_numpy(sc['depth_range']),
                     src_rgbs=torch.from_numpy(sc['src_rgbs']), src_cameras=torch.from_numpy(sc['src_cameras']),
                     static_src_rgbs=torch.from_numpy(sc['static_src_rgbs']), static_src_cameras=torch.from_numpy(sc['static_src_cameras']))
    self.model = _model(dev)
    self.args = types.SimpleNamespace(anti_alias_pooling=0, mask_rgb=1, occ_weights_mode=0)
    self.feat = (T(sc['featmaps']), None, T(sc['static_featmaps']))
    self.proj = projection.Projector(dev)
    self.fidx, self.temb, self.toff = 11, torch.tensor([11 / 24.0], device=dev), [-3, -2, -1, 0, 1, 2, 3][:vdy - num_vv]
    self.rays = H * W

  def render(self):
    smp = sample_ray.RaySamplerSingleImage(self.data, self.dev)
    rb = smp.get_all()
    return render_image.render_single_image_mono((self.fidx, None), (self.temb, None), (self.toff, None), smp, rb, self.model, self.proj, self.chunk, 64,
                                                 self.args, inv_uniform=True, N_importance=0, det=True, featmaps=self.feat, is_train=False,
                                                 num_vv=self.num_vv)


class StressChunk:
  def __init__(self, dev, R=8192, S=128, V=16, H=288, W=512):
    self.dev, self.R, self.S = dev, R, S
    sc = syn.make_scene(seed=41, H=H, W=W, V=V, n_static=V)
    fine = syn.make_scene(seed=41, H=H, W=W, V=V, n_static=V, tag=1)
    T = lambda x: torch.from_numpy(x).to(dev)
    o, d, uv = syn.pixel_rays(sc['camera'], syn.sample_pixels(41, H, W, R))
    self.batch = dict(ray_o=T(o), ray_d=T(d), uv_grid=T(uv), camera=T(sc['camera']), depth_range=T(sc['depth_range']), src_rgbs=T(sc['src_rgbs']),
                      src_cameras=T(sc['src_cameras']), static_src_rgbs=T(sc['static_src_rgbs']), static_src_cameras=T(sc['static_src_cameras']))
    self.model = _model(dev, fine=True)
    self.args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
    self.cfeat = (T(sc['featmaps']), None, T(sc['static_featmaps']))
    self.ffeat = (T(fine['featmaps']), None, T(fine['static_featmaps']))
    self.proj = projection.Projector(dev)
    self.fidx, self.temb = 11, torch.tensor([11 / 24.0], device=dev)
    self.toff = [((i * 5) % 7) - 3 for i in range(V)]

  def render(self):
    return render_ray.render_rays_mv((self.fidx, None), (self.temb, None), (self.toff, None), self.batch, self.model, self.proj, self.cfeat, self.ffeat,
                                     self.S, self.args, inv_uniform=True, N_importance=self.S, det=True, is_train=False)
