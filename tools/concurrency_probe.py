"""Developer probe: which kernel's output changes when another chunk runs on a second HIP stream?  One chunk of the dual-branch coarse stage (motion MLP -> trajectory
points -> gather (7 displaced + 11 static views) -> DynibarDynamic / DynibarStatic -> compositing) is computed ALONE (reference), then again on stream A while a different
chunk loops on stream B; every intermediate tensor is compared bit for bit.  python tools/concurrency_probe.py [R] [--ragged 0|1]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynibar_amd import ops, synthetic as syn
from frame_case import dct_basis

H, W, F, S = 288, 512, 32, 64
dev = torch.device('cuda:0')


class Chunk:
  def __init__(self, sc, first_ray, R):
    T = lambda x: torch.from_numpy(x).to(dev)
    self.R = R
    o, d, _ = syn.pixel_rays(sc['camera'], (np.arange(R) + first_ray) % (H * W))
    self.o, self.d = T(o), T(d)
    self.scene = {k: T(v) for k, v in sc.items()}
    self.time = torch.tensor([11 / 24.0], device=dev)
    self.basis = dct_basis(6, 24).to(dev)


class Nets:
  def __init__(self):
    self.st = ops.StaticNet(syn.make_weights('static', 0, F), dev, True, False)
    self.dy = ops.DynamicNet(syn.make_weights('dynamic', 0, F), dev)
    self.mo = ops.MotionMLP(syn.make_weights('motion', 0), dev, num_basis=6)
    self.views = None


def run(c, n, out=None):
  sc = c.scene
  if n.views is None:
    n.views = (ops.SourceViews(sc['camera'], sc['src_rgbs'], sc['src_cameras'], sc['featmaps']),
               ops.SourceViews(sc['camera'], sc['static_src_rgbs'], sc['static_src_cameras'], sc['static_featmaps']))
  vdy, vst = n.views
  r = {}
  r['pts'], r['z'], _ = ops.sample_along_ray(c.o, c.d, sc['depth_range'], S, True, want_s=False)
  r['coeff'] = n.mo(r['pts'], c.time, 6)
  rows = [(11 + o_) % 24 for o_ in (-3, -2, -1, 0, 1, 2, 3)]
  r['pts_seq'] = ops.trajectory_points(r['coeff'], c.basis, r['pts'], rows, 11)
  r['rf_dy'], _, r['mk_dy'], r['pm_dy'] = ops.project_gather(vdy, c.R, S, xyz=r['pts_seq'], pts_st=r['pts'], pix_mask_thresh=1.0)
  r['rf_st'], r['rd_st'], r['mk_st'], r['pm_st'] = ops.project_gather(vst, c.R, S, ray_o=c.o, ray_d=c.d, z_vals=r['z'], pix_mask_thresh=1.0)
  r['raw_dy'] = n.dy(c.d, r['pts'], r['rf_dy'], r['mk_dy'], c.time)
  r['raw_st'] = n.st(vst, c.o, c.d, r['pts'], r['rf_st'], r['rd_st'], r['mk_st'])
  comp = ops.composite(r['raw_dy'], r['z'], r['pm_dy'], r['raw_st'], r['pm_st'])
  r['rgb'], r['weights'] = comp['rgb'], comp['weights']
  return r


def main():
  R = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8192
  sc = syn.make_scene(seed=0, H=H, W=W, V=7, n_static=11)
  ca, cb = Chunk(sc, 3 * 8192, R), Chunk(sc, 11 * 8192, R)
  na, nb = Nets(), Nets()  # (separate network objects: separate workspaces, as two streams of one frame have)
  ref = run(ca, na); run(cb, nb)
  torch.cuda.synchronize()
  again = run(ca, na)
  torch.cuda.synchronize()
  print('alone, twice: max diff over all stages', max(float((ref[k].float() - again[k].float()).abs().max()) for k in ref))
  sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
  # the networks alone, on fixed inputs, many trials: how often, which network, which points
  vst = na.views[1]
  tally = {'raw_st': [0, 0, 0.0], 'raw_dy': [0, 0, 0.0], 'coeff': [0, 0, 0.0]}
  n_trials = int(os.environ.get('PROBE_TRIALS', '24'))
  for trial in range(n_trials):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
      for _ in range(2):
        run(cb, nb)
    with torch.cuda.stream(sa):
      got = {'raw_st': na.st(vst, ca.o, ca.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st']),
             'raw_dy': na.dy(ca.d, ref['pts'], ref['rf_dy'], ref['mk_dy'], ca.time), 'coeff': na.mo(ref['pts'], ca.time, 6)}
    torch.cuda.synchronize()
    for k, v in got.items():
      dlt = (ref[k] - v).abs().reshape(R, S, -1).amax(dim=2)
      bad = dlt > 0
      if bool(bad.any()):
        tally[k][0] += 1
        tally[k][1] += int(bad.sum())
        tally[k][2] = max(tally[k][2], float(dlt.max()))
        rays = torch.nonzero(bad.any(dim=1)).flatten().tolist()
        print(f'  trial {trial} {k}: {int(bad.sum())} points differ (max {float(dlt.max()):.3e}) in rays {rays[:12]}{"..." if len(rays) > 12 else ""}; samples of the first ray: '
              f'{torch.nonzero(bad[rays[0]]).flatten().tolist()[:70]}', flush=True)
  for k, (nt, npnt, mx) in tally.items():
    print(f'networks under a concurrent chunk, {n_trials} trials: {k}: {nt} trials with differences, {npnt} points in all, max {mx:.3e}', flush=True)
  for trial in range(4):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
      for _ in range(3):
        run(cb, nb)
    with torch.cuda.stream(sa):
      got = run(ca, na)
    torch.cuda.synchronize()
    line = []
    for k in ref:
      dlt = (ref[k].float() - got[k].float()).abs()
      nbad = int((dlt > 0).sum())
      if nbad:
        line.append(f'{k}: {nbad} of {dlt.numel()} differ (max {float(dlt.max()):.3e})')
    print(f'trial {trial}: ' + ('; '.join(line) if line else 'every stage bit-identical to the run alone'), flush=True)


if __name__ == '__main__':
  main()
