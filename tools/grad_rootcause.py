"""Round 5, review item 4: where does the deterministic 0.96-of-the-limit gradient error of `train dual few` (S = 200, 2 rays, 3 dynamic / 4 static views) come
from?  Developer tool for the GPU box:  python tools/grad_rootcause.py [--name few --S 200 --R 2]

(1) every static-branch gradient of the two-branch step: HIP vs the oracle in fp64, next to the oracle's own fp32 vs fp64 (the 'conditioning' the
    test's allowance is built from) -- is the kernel further from the exact gradient than the reference's own fp32 autograd is?
(2) d loss / d raw_st (the cotangent that enters the static net's backward) HIP vs fp64: is the error already there (compositing), or made in the net?
(3) the static net alone, driven by the fp64 cotangent restricted to the points of one class at a time -- classes by the number of valid views of the
    point (0: the -1e9 branch, 1: attention-masked, 2 .. V) -- HIP vs fp64 per class: which rows carry the error?
Run it on the default library and on the exact 6-term build (DYNIBAR_HIP_LIB=.../libdynibar_hip_x6.so): equal errors = not the split products."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity  # noqa: E402
from oracle import ibr_oracle as O  # noqa: E402  (developer tool: the oracle is the checker here)
from dynibar_amd import ops, train_dynamic as TD, train_motion as TM, train_static as TS  # noqa: E402


def capture_oracle(name, S, R, dtype):
  cap = {}
  orig = O.static_net

  def wrapped(*a, **k):
    r = orig(*a, **k)
    r.retain_grad()
    cap['raw_st'] = r
    cap['mask'] = a[7]
    return r

  O.static_net = wrapped
  try:
    di, v_ref, cot, g, keep = parity.train_dual_reference(name, S, R, 'init', 5.0, 0, dtype=dtype)
  finally:
    O.static_net = orig
  return di, cot, g, keep, cap['raw_st'].grad.detach(), cap['raw_st'].detach(), cap['mask'].detach()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--name', default='few')
  ap.add_argument('--S', type=int, default=200)
  ap.add_argument('--R', type=int, default=2)
  a = ap.parse_args()
  dev = 'cuda:0'
  name, S, R = a.name, a.S, a.R
  di, cot, g32, keep, draw32, raw32, mask = capture_oracle(name, S, R, torch.float32)
  _, _, g64, _, draw64, raw64, _ = capture_oracle(name, S, R, torch.float64)
  scene = di['scene']
  sc = parity.to_dev(scene, dev)
  shift = 5.0
  cap = {}

  def hip_step():
    fm_dy = scene['featmaps'].detach().clone().to(dev).requires_grad_(True)
    fm_st = scene['static_featmaps'].detach().clone().to(dev).requires_grad_(True)
    od, dd, pts, pts_seq, z = (di[k].to(dev) for k in ('o', 'd', 'pts', 'pts_seq', 'z'))
    Rn = od.shape[0]
    views_dy = ops.SourceViews(sc['camera'], sc['src_rgbs'], sc['src_cameras'], fm_dy.detach())
    views_st = ops.SourceViews(sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], fm_st.detach())
    rf, _, mk, pm_dy = TM.gather(views_dy, fm_dy, Rn, S, xyz=pts_seq, pts_st=pts, pix_mask_thresh=1.0)
    rfs, rds, mks, pm_st = TM.gather(views_st, fm_st, Rn, S, ray_o=od, ray_d=dd, z_vals=z, pix_mask_thresh=1.0)
    prm_dy = {k: v.detach().to(dev).requires_grad_(True) for k, v in di['W']['net_coarse_dy'].items()}
    prm_st = {k: v.detach().to(dev).requires_grad_(True) for k, v in di['W']['net_coarse_st'].items()}
    raw_dy = TD.dynamic_raw(prm_dy, shift, rf, dd, pts, mk, di['temb'].to(dev))
    raw_st = TS.static_raw(prm_st, (True, False), views_st, rfs, od, dd, pts, rds, mks)
    raw_st.retain_grad()
    out = TD.composite_dual(raw_dy, raw_st, z, pm_dy, pm_st)
    out_dy = TS.composite_vanilla(raw_dy, z, pm_dy)
    loss = sum((out[k] * cot[k].to(dev)).sum() for k in cot if k != 'dy_rgb') + (out_dy['rgb'] * cot['dy_rgb'].to(dev)).sum()
    loss.backward()
    cap.update(raw_st=raw_st.detach().cpu(), draw=raw_st.grad.detach().cpu(), inputs=(views_st, rfs.detach(), od, dd, pts, rds.detach(), mks.detach()))
    return {'st/' + k: v.grad.detach().cpu() for k, v in prm_st.items()}

  got = hip_step()
  from dynibar_amd import _lib
  print(f'library: {_lib.LIB_PATH}   split terms {_lib.lib().dyn_mlp_split_terms()} kind {_lib.lib().dyn_mlp_split_kind()}   case {name} S={S} R={R}')
  gmax = max(float(v.abs().max()) for k, v in g32.items() if not k.startswith('featmaps'))
  print('\n(1) static-branch gradients: error / max|g| of the tensor      [hip vs fp64 | oracle-fp32 vs fp64 | hip vs oracle-fp32]   max|g|')
  rows = []
  for k in sorted(g32):
    if not k.startswith('st/'):
      continue
    r64 = g64[k].double()
    h = got[k].double().reshape(r64.shape)
    r32 = g32[k].double()
    sc_ = float(r64.abs().max())
    if sc_ < 1e-3 * gmax:
      continue
    rows.append((float((h - r64).abs().max()) / sc_, float((r32 - r64).abs().max()) / sc_, float((h - r32).abs().max()) / sc_, sc_, k))
  for e in sorted(rows, reverse=True):
    print(f'   {e[0]:9.2e}   {e[1]:9.2e}   {e[2]:9.2e}   {e[3]:9.2e}  {e[4]}')

  print('\n(2) the cotangent d loss / d raw_st entering the static net, and the forward raw_st:')
  live = raw64[..., 3] > -1e8
  for nm, hv, v32, v64 in (('raw_st rgb', cap['raw_st'][..., :3], raw32[..., :3], raw64[..., :3]), ('raw_st sigma (live)', cap['raw_st'][..., 3][live], raw32[..., 3][live], raw64[..., 3][live]),
                           ('d raw_st rgb', cap['draw'][..., :3], draw32[..., :3], draw64[..., :3]), ('d raw_st sigma (live)', cap['draw'][..., 3][live], draw32[..., 3][live], draw64[..., 3][live])):
    s_ = float(v64.abs().max())
    print(f'   {nm:24s} max|.| {s_:9.3e}   hip-fp64 {float((hv.double() - v64).abs().max()) / s_:9.2e}   fp32-fp64 {float((v32.double() - v64).abs().max()) / s_:9.2e}  (relative to max)')

  print('\n(3) the static net ALONE under the fp64 cotangent restricted to one class of points (by number of valid views); error of out_geometry_fc.0.weight,')
  print('    ray_dir_fc.0.weight, base_fc.0.weight / max|g| of the FULL gradient of that tensor:   [hip vs fp64 | oracle-fp32 vs fp64]  points in class')
  nvalid = mask[..., 0].sum(dim=2).round().long()  # [R,S]
  Vs = mask.shape[2]
  views_st, rfs, od, dd, pts, rds, mks = cap['inputs']
  keys = ('out_geometry_fc.0.weight', 'ray_dir_fc.0.weight', 'base_fc.0.weight', 'geometry_fc.2.weight')
  full = {k: float(g64['st/' + k].abs().max()) for k in keys}

  def oracle_net_grads(dtype, cotan):
    cv = lambda v: v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v
    sd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in di['W']['net_coarse_st'].items()}
    scc = {k: cv(v) for k, v in scene.items()}
    p_, o_, d_ = cv(di['pts']), cv(di['o']), cv(di['d'])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
      rfs_, rds_, mks_ = O.compute_with_motions(p_, p_[None].repeat(Vs, 1, 1, 1), scc['camera'], scc['static_src_rgbs'], scc['static_src_cameras'], scc['static_featmaps'])
      raw = O.static_net(sd, p_, O.ref_plucker(o_, d_), O.src_plucker(p_, scc['static_src_cameras']), rfs_, F.normalize(d_, dim=-1), rds_, mks_, True, False)
      raw.backward(cotan.to(dtype))
    finally:
      torch.set_default_dtype(prev)
    return {k: v.grad.detach().double() for k, v in sd.items()}

  def hip_net_grads(cotan):
    prm = {k: v.detach().to(dev).requires_grad_(True) for k, v in di['W']['net_coarse_st'].items()}
    raw = TS.static_raw(prm, (True, False), views_st, rfs, od, dd, pts, rds, mks)
    raw.backward(cotan.float().to(dev))
    return {k: v.grad.detach().cpu().double() for k, v in prm.items()}

  classes = [('all', torch.ones_like(nvalid, dtype=torch.bool))] + [(f'nvalid={n}', nvalid == n) for n in range(Vs + 1)]
  for tag, sel in classes:
    if int(sel.sum()) == 0:
      continue
    ct = draw64 * sel[..., None].double()
    g_h, g_32, g_64 = hip_net_grads(ct), oracle_net_grads(torch.float32, ct), oracle_net_grads(torch.float64, ct)
    line = f'   {tag:10s} {int(sel.sum()):5d} pts '
    for k in keys:
      line += f' | {k.split(".weight")[0]:18s} {float((g_h[k] - g_64[k]).abs().max()) / full[k]:8.1e} {float((g_32[k] - g_64[k]).abs().max()) / full[k]:8.1e}'
    print(line)
  print('\n(4) the ORACLE\'s fp64 static-net backward driven by the HIP cotangent (what the compositing backward kernel handed over) vs by its own fp64 cotangent;')
  print('    and (5) the HIP net driven by the HIP cotangent vs the oracle fp64 net driven by the same HIP cotangent (the net alone, again):')
  ct_hip = cap['draw'].double()
  g_o_hipct = oracle_net_grads(torch.float64, ct_hip)
  g_o_64ct = oracle_net_grads(torch.float64, draw64)
  g_h_hipct = hip_net_grads(ct_hip)
  for k in keys:
    print(f'   {k:28s} oracle64(ct_hip) - oracle64(ct_64): {float((g_o_hipct[k] - g_o_64ct[k]).abs().max()) / full[k]:8.1e}    hip(ct_hip) - oracle64(ct_hip): '
          f'{float((g_h_hipct[k] - g_o_hipct[k]).abs().max()) / full[k]:8.1e}    full-step hip - g64: {float((got["st/" + k].double() - g64["st/" + k].double()).abs().max()) / full[k]:8.1e}'
          f'    oracle64(ct_64) - g64 (sanity): {float((g_o_64ct[k] - g64["st/" + k].double()).abs().max()) / full[k]:8.1e}')
  dsg_h, dsg_64, dsg_32 = cap['draw'][..., 3].double(), draw64[..., 3].double(), draw32[..., 3].double()
  print('\n(6) d loss / d sigma per sample, ray 0: sample, fp64 value, hip - fp64, fp32 - fp64  (every 10th sample + the 5 largest |hip - fp64|)')
  idx = sorted(set(list(range(0, S, 10)) + [int(i) for i in (dsg_h[0] - dsg_64[0]).abs().topk(5).indices]))
  for i in idx:
    print(f'     {i:4d}  {float(dsg_64[0, i]):+.6e}  {float(dsg_h[0, i] - dsg_64[0, i]):+.3e}  {float(dsg_32[0, i] - dsg_64[0, i]):+.3e}')
  eh, e32 = (dsg_h - dsg_64), (dsg_32 - dsg_64)
  print(f'     sum over samples of (hip - fp64): {[float(v) for v in eh.sum(dim=1)]}   of |hip - fp64|: {[float(v) for v in eh.abs().sum(dim=1)]}')
  print(f'     sum over samples of (fp32 - fp64): {[float(v) for v in e32.sum(dim=1)]}   of |fp32 - fp64|: {[float(v) for v in e32.abs().sum(dim=1)]}')
  for nm in ('rgb',):
    er = (cap['draw'][..., :3].double() - draw64[..., :3].double())
    print(f'     d raw rgb: sum (hip - fp64) {float(er.sum()):+.3e}, sum |.| {float(er.abs().sum()):.3e}')
  # where inside out_geometry_fc.0.weight: rows (output features) or columns (attention output features)?
  k = 'out_geometry_fc.0.weight'
  e = (got['st/' + k].double() - g64['st/' + k].double()).abs()
  print(f'\n   {k}: error by output row (max over columns) top 5: {[round(float(v), 9) for v in e.max(dim=1).values.topk(5).values]}; '
        f'by input column top 5: {[round(float(v), 9) for v in e.max(dim=0).values.topk(5).values]}; median element error {float(e.median()):.2e}, max {float(e.max()):.2e}')


if __name__ == '__main__':
  main()
