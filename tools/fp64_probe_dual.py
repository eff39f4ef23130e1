"""Error distributions of the two-branch pass (inference kernels) and of the fp32 oracle against the float64 oracle, per output.  GPU box:  python tools/fp64_probe_dual.py"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
import parity
from parity import *
dev = 'cuda:0'
for kw in (dict(name='small', S=48, weights='trained'), dict(name='small', S=48, weights='init')):
  name, S, weights, shift = kw['name'], kw['S'], kw['weights'], 5.0
  di, v32, _, _, keep = train_dual_reference(name, S, None, weights, shift)
  _, v64, _, _, keep64 = train_dual_reference(name, S, None, weights, shift, dtype=torch.float64)
  keep = keep & keep64
  sc = to_dev(di['scene'], dev)
  views_dy = ops.SourceViews(sc['camera'], sc['src_rgbs'], sc['src_cameras'], sc['featmaps'])
  views_st = ops.SourceViews(sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], sc['static_featmaps'])
  od, dd, pts, pts_seq, z = (di[k].to(dev) for k in ('o', 'd', 'pts', 'pts_seq', 'z'))
  Rn = od.shape[0]
  rf, _, mk, pm_dy = ops.project_gather(views_dy, Rn, S, pts_st=pts, xyz=pts_seq, pix_mask_thresh=1.0)
  rfs, rds, mks, pm_st = ops.project_gather(views_st, Rn, S, ray_o=od, ray_d=dd, z_vals=z, pix_mask_thresh=1.0)
  raw_dy = ops.DynamicNet(parity._weights(weights)['net_coarse_dy'], dev, shift=shift)(dd, pts, rf, mk, di['temb'].to(dev))
  raw_st = ops.StaticNet(parity._weights(weights)['net_coarse_st'], dev, True, False)(views_st, od, dd, pts, rfs, rds, mks)
  out = ops.composite(raw_dy, z, pm_dy, raw_static=raw_st, pix_mask_st=pm_st)
  ours = dict(raw_dy=cpu(raw_dy), raw_st=cpu(raw_st), rgb=cpu(out['rgb']), rgb_dy=cpu(out['rgb_dy']), weights=cpu(out['weights']), weights_dy=cpu(out['weights_dy']))
  q = lambda e, p: float(torch.quantile(e, p))
  for k in ('raw_dy', 'raw_st', 'weights_dy', 'weights', 'rgb_dy', 'rgb'):
    t = v64[k][keep]; f = v32[k][keep].double(); r = ours[k][keep].double()
    parts = (('rgb', slice(0, 3)), ('sigma', slice(3, 4))) if k.startswith('raw') else (('', slice(None)),)
    live = (t[..., 3] > -1e8) if k.startswith('raw') else torch.ones(t.shape[:-1] if t.dim() > 1 else t.shape, dtype=torch.bool)
    for nm, sl in parts:
      tt, ff, rr = (x[..., sl] if k.startswith('raw') else x for x in (t, f, r))
      eo = (rr - tt).abs()[live].flatten(); er = (ff - tt).abs()[live].flatten(); eor = (rr - ff).abs()[live].flatten()
      print(f'{name} {weights} {k} {nm}: ours p50 {q(eo,.5):.2e} p90 {q(eo,.9):.2e} p99 {q(eo,.99):.2e} max {float(eo.max()):.2e} | ref32 p50 {q(er,.5):.2e} p90 {q(er,.9):.2e} p99 {q(er,.99):.2e} max {float(er.max()):.2e} | ours-ref32 max {float(eor.max()):.2e}  scale {float(tt[live].abs().max()):.2f}')
  if weights == 'trained':
    e = (ours['rgb'].double() - v64['rgb']).abs().max(dim=1)[0]
    e[~keep] = 0
    r = int(e.argmax())
    print('worst ray', r, 'rgb err ours', (ours['rgb'][r].double() - v64['rgb'][r]).tolist(), 'ref32', (v32['rgb'][r].double() - v64['rgb'][r]).tolist())
    torch.set_printoptions(precision=6, linewidth=250)
    w64 = v64['weights'][r]
    idx = torch.nonzero(w64 > 1e-6).flatten()[:8]
    print('samples with weight', idx.tolist(), w64[idx].tolist())
    for nm in ('raw_dy', 'raw_st'):
      print(nm, 'truth', v64[nm][r][idx]); print(nm, 'ours-truth', ours[nm][r][idx].double() - v64[nm][r][idx]); print(nm, 'ref32-truth', v32[nm][r][idx].double() - v64[nm][r][idx])
    print('z', di['z'][r][:4].tolist())
    for k in ('rgb_static', 'rgb_dy'):
      if k in out: print(k, 'ours', cpu(out[k])[r].tolist())
