"""Developer probe 4: which (co-runner on stream B, victim on stream A) pairs change the victim's output?  Co-runners loop for the whole trial; every victim op runs several
times on fixed inputs and is compared bit for bit with its run alone.  Run with DYN_RAGGED=0.  python tools/concurrency_probe4.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynibar_amd import ops, synthetic as syn
import concurrency_probe as P
import concurrency_probe2 as P2

R, S, V = P2.R, P2.S, 8
dev = torch.device('cuda:0')


def main():
  sc = syn.make_scene(seed=0, H=P2.H, W=P2.W, V=7, n_static=V)
  ca, cb = P.Chunk(sc, 3 * 8192, R), P.Chunk(sc, 11 * 8192, R)
  na, nb = P.Nets(), P.Nets()
  ref = P.run(ca, na); refb = P.run(cb, nb)
  torch.cuda.synchronize()
  o_ref, n_ref = P2.layout(V)['ref']
  sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
  ws_a = lambda: list(na.st._ws.bufs.values())[-1]
  vdy, vst = na.views
  rows = [(11 + o_) % 24 for o_ in (-3, -2, -1, 0, 1, 2, 3)]

  def victims():
    out = {}
    out['sample'] = ops.sample_along_ray(ca.o, ca.d, ca.scene['depth_range'], S, True, want_s=False)[0]
    out['motion'] = na.mo(ref['pts'], ca.time, 6)
    out['traj'] = ops.trajectory_points(ref['coeff'], ca.basis, ref['pts'], rows, 11)
    out['gather_st'] = ops.project_gather(vst, R, S, ray_o=ca.o, ray_d=ca.d, z_vals=ref['z'], pix_mask_thresh=1.0)[0]
    out['gather_dy'] = ops.project_gather(vdy, R, S, xyz=ref['pts_seq'], pts_st=ref['pts'], pix_mask_thresh=1.0)[0]
    out['dynamic_net'] = na.dy(ca.d, ref['pts'], ref['rf_dy'], ref['mk_dy'], ca.time)
    out['static_net'] = na.st(vst, ca.o, ca.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st'])
    out['static_ref_feat'] = ws_a()[o_ref:o_ref + n_ref].clone()
    out['composite'] = ops.composite(ref['raw_dy'], ref['z'], ref['pm_dy'], ref['raw_st'], ref['pm_st'])['rgb']
    return out

  with torch.cuda.stream(sa):
    good = victims()
    torch.cuda.synchronize()
    again = victims()
    torch.cuda.synchronize()
  print('alone twice:', {k: int((good[k] != again[k]).sum()) for k in good})
  co = {
      'motion': lambda: nb.mo(refb['pts'], cb.time, 6),
      'dynamic net': lambda: nb.dy(cb.d, refb['pts'], refb['rf_dy'], refb['mk_dy'], cb.time),
      'static net': lambda: nb.st(nb.views[1], cb.o, cb.d, refb['pts'], refb['rf_st'], refb['rd_st'], refb['mk_st']),
      'gather': lambda: ops.project_gather(nb.views[1], R, S, ray_o=cb.o, ray_d=cb.d, z_vals=refb['z'], pix_mask_thresh=1.0),
  }
  reps = {'motion': 14, 'dynamic net': 8, 'static net': 6, 'gather': 200}
  for name, fn in co.items():
    tot = {k: 0 for k in good}
    for trial in range(4):
      torch.cuda.synchronize()
      with torch.cuda.stream(sb):
        for _ in range(reps[name]):
          fn()
      with torch.cuda.stream(sa):
        for _ in range(3):
          got = victims()
          for k in got:
            tot[k] += int((got[k] != good[k]).sum())
      torch.cuda.synchronize()
    print(f'co-runner "{name}": differing dwords per victim over 12 victim rounds: {tot}', flush=True)


if __name__ == '__main__':
  main()
