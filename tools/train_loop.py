"""The reference's two training stages on synthetic data, end to end on this package (train.py:116-199 static bootstrap, :203-467 main loop):
seeded scene and targets, the reference's optimizer setup (Adam over the three MLPs' parameters + the trajectory basis, model.py:339-378),
`render_rays_mono` under grad mode, the script's loss, `loss.backward()` through the HIP training kernels, `optimizer.step()`.
Prints the loss every few iterations; the losses must go down.     python tools/train_loop.py [iterations] [rays]
"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynibar_amd import criterion, render_ray  # noqa: E402
from train_case import TrainCase  # noqa: E402


def main_loss(ret, batch):
  """the colour, disparity, flow, cycle and regularisation terms of train.py:302-378 (config weights of configs/train_kid-running.txt)"""
  crit = criterion.Criterion()
  ref, anc = ret['outputs_coarse_ref'], ret['outputs_coarse_anchor']
  loss = crit(ref, batch) + criterion.compute_temporal_rgb_loss(anc, batch)
  loss = loss + crit(ret['outputs_coarse_ref_dy'], batch, motion_mask=batch['motion_mask'])
  pm = ref['mask'].float()
  loss = loss + 0.1 * torch.sum(torch.abs(1.0 / torch.clamp(ref['depth'], min=1e-2) - batch['disp']) * pm) / (torch.sum(pm) + 1e-8)
  nv = ref['render_flows'].shape[0]
  loss = loss + 0.01 * criterion.compute_flow_loss(ref['render_flows'], batch['flows'][:nv], pm[None, :, None] * batch['masks'][:nv])
  pa, pr = anc['pts_traj_anchor'], anc['pts_traj_ref']
  ow = anc['occ_weights'][None, ..., None].repeat(pa.shape[0], 1, 1, pa.shape[-1])
  loss = loss + 0.1 * torch.sum(torch.abs(pr - pa) * ow) / (torch.sum(ow) + 1e-8)
  return loss + 0.05 * torch.mean(torch.abs(anc['sf_seq']))


def run(dev='cuda:0', iters=40, R=1024, S=64, log_every=10, quiet=False, encoders=False):
  """encoders: also train feature_net / feature_net_st (train.py:272-281: the maps are recomputed from the source images every iteration by
  the encoder's training form, the gradients of the maps flow on into its parameters, which sit in the optimizer with lrate_feature)."""
  tc = TrainCase(dev, R=R, S=S)
  g = torch.Generator().manual_seed(9)
  batch = dict(tc.batch)
  batch.update(rgb=torch.rand(R, 3, generator=g).to(dev), disp=(0.05 + 0.5 * torch.rand(R, generator=g)).to(dev),
               flows=(4.0 * torch.randn(6, R, 2, generator=g)).to(dev), masks=(torch.rand(6, R, 1, generator=g) < 0.8).float().to(dev),
               motion_mask=(torch.rand(R, generator=g) < 0.5).float().to(dev), static_mask=(torch.rand(R, generator=g) < 0.3).float().to(dev))
  m = tc.model
  params = [p for n in ('net_coarse_st', 'net_coarse_dy', 'motion_mlp') for p in getattr(m, n).values()] + [m.trajectory_basis]
  groups = [{'params': params, 'lr': 4e-4}]  # lrate_mlp of the config (model.py:339-378)
  enc = None
  if encoders:
    from dynibar_amd import synthetic as syn, train_encoder
    enc = [{k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in syn.make_encoder_weights(sd).items() if k in train_encoder.PARAMS} for sd in (0, 1)]
    groups.append({'params': [p for e in enc for p in e.values()], 'lr': 1e-3})  # lrate_feature
  opt = torch.optim.Adam(groups)
  hist = {'bootstrap': [], 'main': []}
  for stage in ('bootstrap', 'main'):
    t0 = time.perf_counter()
    for it in range(iters):
      opt.zero_grad()
      feat = tc.feat
      if enc is not None:  # train.py:264-281
        nd = batch['src_rgbs'].shape[1]
        cb = torch.cat([batch['src_rgbs'].squeeze(0).permute(0, 3, 1, 2), batch['anchor_src_rgbs'].squeeze(0).permute(0, 3, 1, 2)], 0)
        cb_maps, _ = train_encoder.encoder_forward(enc[0], cb)
        st_maps, _ = train_encoder.encoder_forward(enc[1], batch['static_src_rgbs'].squeeze(0).permute(0, 3, 1, 2))
        feat = (cb_maps[:nd], cb_maps[nd:], st_maps)
      ret = render_ray.render_rays_mono(tc.fidx, tc.temb, tc.toff, batch, m, feat, tc.proj, S, tc.args, inv_uniform=True, det=True,
                                        is_train=(stage == 'main'), num_vv=tc.num_vv)
      if stage == 'bootstrap':   # train.py:180-190
        w = (1.0 - batch['static_mask']) * ret['outputs_coarse_ref']['mask'].float()
        loss = criterion.compute_rgb_loss(ret['outputs_coarse_st']['rgb'], batch, w)
      else:
        loss = main_loss(ret, batch)
      loss.backward()
      opt.step()
      if it % log_every == 0 or it == iters - 1:
        hist[stage].append(float(loss.detach()))
        if not quiet:
          print(f"{stage:9s} iteration {it:4d}  loss {float(loss.detach()):.5f}", flush=True)
    torch.cuda.synchronize()
    hist[stage + '_ms_per_iteration'] = (time.perf_counter() - t0) / iters * 1e3
  return hist


if __name__ == '__main__':
  it = int(sys.argv[1]) if len(sys.argv) > 1 else 40
  R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
  print(json.dumps(run(iters=it, R=R, encoders='--encoders' in sys.argv)))
