"""Synthetic kid-running-shaped training iteration (configs/train_kid-running.txt, train.py:203-467): N_rand rays x 64 samples, 7 time-offset
+ 3 virtual dynamic views at the reference and at the anchor frame, 15 static views, anti_alias_pooling 0, mask_rgb 1;
render_rays_mono(is_train=True) under grad mode, a loss over every differentiable output, loss.backward().  Shared by bench.py and
tools/trainbench.py."""
import types

import numpy as np
import torch

from dynibar_amd import projection, render_ray, synthetic as syn
from frame_case import NUM_BASIS, NUM_FRAMES, dct_basis


class TrainCase:
  def __init__(self, dev, R=3072, S=64, H=288, W=512, vdy=10, vst=15, num_vv=3):
    self.dev, self.R, self.S, self.num_vv = dev, R, S, num_vv
    sc = syn.make_scene(seed=21, H=H, W=W, V=vdy, n_static=vst, smooth=False)
    anc = syn.make_scene(seed=22, H=H, W=W, V=vdy, n_static=vst, smooth=False)
    T = lambda x: torch.from_numpy(x).to(dev)
    o, d, uv = syn.pixel_rays(sc['camera'], syn.sample_pixels(21, H, W, R))
    self.batch = dict(ray_o=T(o), ray_d=T(d), uv_grid=T(uv), camera=T(sc['camera']), depth_range=T(sc['depth_range']), src_rgbs=T(sc['src_rgbs']),
                      src_cameras=T(sc['src_cameras']), static_src_rgbs=T(sc['static_src_rgbs']), static_src_cameras=T(sc['static_src_cameras']),
                      anchor_src_rgbs=T(anc['src_rgbs']), anchor_src_cameras=T(sc['src_cameras']))
    P = lambda kind, **kw: {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in syn.make_weights(kind, 0, **kw).items() if k != 's'}
    self.model = types.SimpleNamespace(net_coarse_st=P('static'), net_coarse_dy=P('dynamic'), motion_mlp=P('motion', num_basis=NUM_BASIS),
                                       trajectory_basis=dct_basis(NUM_BASIS, NUM_FRAMES).to(dev).requires_grad_(True))
    self.args = types.SimpleNamespace(anti_alias_pooling=0, mask_rgb=1, occ_weights_mode=0)
    self.feat = tuple(T(x).requires_grad_(True) for x in (sc['featmaps'], anc['featmaps'], sc['static_featmaps']))
    self.proj = projection.Projector(dev)
    nt = vdy - num_vv
    self.fidx, self.temb = (11, 12), (torch.tensor([11 / 24.0], device=dev), torch.tensor([12 / 24.0], device=dev))
    self.toff = ([-3, -2, -1, 0, 1, 2, 3][:nt], [-2, -1, 1, 2, 3, -3, 0][:nt])
    g = torch.Generator().manual_seed(3)
    self.c_rgb, self.c_w = torch.randn(R, 3, generator=g).to(dev), (0.1 * torch.randn(R, S, generator=g)).to(dev)

  def parameters(self):
    ps = [self.model.trajectory_basis] + list(self.feat)
    for n in ('net_coarse_st', 'net_coarse_dy', 'motion_mlp'):
      ps += list(getattr(self.model, n).values())
    return ps

  def step(self):
    """forward + loss + backward of one iteration; gradients accumulate in the leaves' .grad"""
    ret = render_ray.render_rays_mono(self.fidx, self.temb, self.toff, self.batch, self.model, self.feat, self.proj, self.S, self.args, inv_uniform=True,
                                      det=True, is_train=True, num_vv=self.num_vv)
    ref, anc = ret['outputs_coarse_ref'], ret['outputs_coarse_anchor']
    loss = (ref['rgb'] * self.c_rgb).sum() + (anc['rgb'] * self.c_rgb).sum() + (ret['outputs_coarse_st']['rgb'] * self.c_rgb).sum() + \
        (ret['outputs_coarse_ref_dy']['rgb'] * self.c_rgb).sum() + (ret['outputs_coarse_anchor_dy']['rgb'] * self.c_rgb).sum() + \
        (ref['weights'] * self.c_w).sum() + ref['depth'].sum() * 0.01 + ref['render_flows'].abs().mean() + \
        (anc['pts_traj_ref'] - anc['pts_traj_anchor']).abs().mean() + anc['sf_seq'].abs().mean()
    loss.backward()
    return loss.detach()

  def algorithmic_flops(self):
    """forward FLOPs of SURVEY 8d per sample point x 3 (forward, data gradient, weight gradient): static + 2 x dynamic + 2 x motion"""
    S, pts = self.S, self.R * self.S
    vdy = self.batch['src_rgbs'].shape[1]
    vst = self.batch['static_src_rgbs'].shape[1]
    fwd = (0.361e6 + 0.033e6 * (S / 64) + 0.4305e6 * vst) + 2 * (0.566e6 + 0.033e6 * (S / 64) + 0.2468e6 * vdy) + 2 * 1.062e6
    return 3.0 * fwd * pts
