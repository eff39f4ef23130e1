"""Developer probe 3: k_static_ref_feat's output [R,36] changes when k_motion_mlp runs on another stream.  What ARE the changed values?
Run with DYN_RAGGED=0.  python tools/concurrency_probe3.py"""
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
  ws_of = lambda n: list(n.st._ws.bufs.values())[-1]
  call_a = lambda: na.st(na.views[1], ca.o, ca.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st'])
  with torch.cuda.stream(sa):
    call_a(); torch.cuda.synchronize()
    good = ws_of(na)[o_ref:o_ref + n_ref].clone().reshape(R, 36)
  with torch.cuda.stream(sb):
    nb.st(nb.views[1], cb.o, cb.d, refb['pts'], refb['rf_st'], refb['rd_st'], refb['mk_st']); torch.cuda.synchronize()
    good_b = ws_of(nb)[o_ref:o_ref + n_ref].clone().reshape(R, 36)
  print('ws pointers: a', hex(ws_of(na).data_ptr()), 'b', hex(ws_of(nb).data_ptr()), ' ray_o a', hex(ca.o.data_ptr()), 'ray_d a', hex(ca.d.data_ptr()))
  for mode in ('motion', 'motion, sync before a', 'points-only pts tensor'):
    for trial in range(6):
      torch.cuda.synchronize()
      with torch.cuda.stream(sb):
        for _ in range(6):
          co = nb.mo(ref['pts'], cb.time, 6)
      if mode == 'motion, sync before a':
        torch.cuda.synchronize()  # (control: the same sequence without overlap)
      with torch.cuda.stream(sa):
        call_a()
      torch.cuda.synchronize()
      with torch.cuda.stream(sa):
        got = ws_of(na)[o_ref:o_ref + n_ref].clone().reshape(R, 36)
      torch.cuda.synchronize()
      bad = (got != good)
      rays = torch.nonzero(bad.any(dim=1)).flatten().tolist()
      print(f'[{mode}] trial {trial}: {int(bad.sum())} dwords in {len(rays)} rays differ: rays {rays[:20]}', flush=True)
      for r in rays[:3]:
        ch = torch.nonzero(bad[r]).flatten().tolist()
        d_all = (good - got[r][None]).abs().amax(dim=1)
        d_b = (good_b - got[r][None]).abs().amax(dim=1)
        print(f'    ray {r}: channels {ch}\n      alone      {good[r][:8].tolist()}\n      concurrent {got[r][:8].tolist()}\n      nearest alone ray of a: {int(d_all.argmin())} (max diff {float(d_all.min()):.3e}); '
              f'nearest ray of b: {int(d_b.argmin())} (max diff {float(d_b.min()):.3e}); finite {bool(torch.isfinite(got[r]).all())}', flush=True)
    if mode == 'motion':
      pass


if __name__ == '__main__':
  main()
