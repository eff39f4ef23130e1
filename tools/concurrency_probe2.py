"""Developer probe 2: DynibarStatic on fixed inputs while another chunk loops on a second stream -- WHICH of its kernels produces different bytes?  After every call the
network's workspace is read back: the view kernel's products (parked x, row records, geometry_fc input records, nvalid) and the point kernel's (hg) are compared with the
run alone.  Run with DYN_RAGGED=0 (the regular dense layout is replicated here).  python tools/concurrency_probe2.py [V]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynibar_amd import ops, synthetic as syn
import concurrency_probe as P

H, W, F, S, R = 288, 512, 32, 64, 8192
dev = torch.device('cuda:0')


def layout(V):
  n_pts = R * S
  dense = (9 <= V <= 12) or (17 <= V <= 26)
  PT = 256 // V if dense else 32 // (4 if V <= 4 else 8 if V <= 8 else 16 if V <= 16 else 32)
  nta = ((n_pts + PT - 1) // PT) * (8 if dense else 1)
  tpr = 2
  ntb = R * tpr
  o = {}
  off = 0
  o['x'] = (off, nta * 4096); off += nta * 4096
  o['rec'] = (off, nta * 128); off += nta * 128
  o['gin'] = (off, ntb * 33 * 256); off += ntb * 33 * 256
  o['nvalid'] = (off, n_pts); off += (n_pts + 3) & ~3
  o['hg'] = (off, ntb * 16 * 256); off += ntb * 16 * 256
  o['ref'] = (off, R * 36)
  return o


def main():
  V = int(sys.argv[1]) if len(sys.argv) > 1 else 11
  assert os.environ.get('DYN_RAGGED') == '0'
  sc = syn.make_scene(seed=0, H=H, W=W, V=7, n_static=V)
  ca, cb = P.Chunk(sc, 3 * 8192, R), P.Chunk(sc, 11 * 8192, R)
  na, nb = P.Nets(), P.Nets()
  ref = P.run(ca, na); P.run(cb, nb)
  torch.cuda.synchronize()
  lay = layout(V)
  vst = na.views[1]
  sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

  def snap():
    (ws,) = [b for b in na.st._ws.bufs.values()] if len(na.st._ws.bufs) == 1 else [list(na.st._ws.bufs.values())[-1]]
    return {k: ws[o:o + n].clone() for k, (o, n) in lay.items()}

  with torch.cuda.stream(sa):
    raw0 = na.st(vst, ca.o, ca.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st'])
    torch.cuda.synchronize()
    s0 = snap()
    raw1 = na.st(vst, ca.o, ca.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st'])
    torch.cuda.synchronize()
    s1 = snap()
  print('alone twice on the side stream: raw diff', float((raw0 - raw1).abs().max()), {k: int((s0[k] != s1[k]).sum()) for k in s0})
  for mode in ('full chunk', 'gather only', 'dynamic net only', 'static net only', 'motion only'):
    tot = {k: 0 for k in lay}
    tot['raw'] = 0
    ntr = 10
    for trial in range(ntr):
      torch.cuda.synchronize()
      with torch.cuda.stream(sb):
        for _ in range(2):
          if mode == 'full chunk':
            P.run(cb, nb)
          elif mode == 'gather only':
            for _i in range(40):
              ops.project_gather(nb.views[1], R, S, ray_o=cb.o, ray_d=cb.d, z_vals=ref['z'], pix_mask_thresh=1.0)
          elif mode == 'dynamic net only':
            for _i in range(3):
              nb.dy(cb.d, ref['pts'], ref['rf_dy'], ref['mk_dy'], cb.time)
          elif mode == 'static net only':
            nb.st(nb.views[1], cb.o, cb.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st'])
          else:
            for _i in range(3):
              nb.mo(ref['pts'], cb.time, 6)
      with torch.cuda.stream(sa):
        raw = na.st(vst, ca.o, ca.d, ref['pts'], ref['rf_st'], ref['rd_st'], ref['mk_st'])
      torch.cuda.synchronize()
      with torch.cuda.stream(sa):
        s = snap()
      torch.cuda.synchronize()
      bad = {k: int((s0[k] != s[k]).sum()) for k in s0}
      bad['raw'] = int((raw0 != raw).sum())
      for k in bad:
        tot[k] += bad[k]
      if any(bad.values()) and trial < 3:
        d = (s0['gin'] != s['gin']).reshape(-1, 33, 64, 4)
        tiles = torch.nonzero(d.reshape(d.shape[0], -1).any(dim=1)).flatten().tolist()
        recs = torch.nonzero(d.any(dim=0).any(dim=1).any(dim=1)).flatten().tolist()
        lanes = torch.nonzero(d.any(dim=0).any(dim=0).any(dim=1)).flatten().tolist()
        dx = (s0['x'] != s['x']).reshape(-1, 4096)
        xt = torch.nonzero(dx.any(dim=1)).flatten().tolist()
        print(f'    [{mode}] trial {trial}: differing dwords {bad}; gin tiles {tiles[:10]} records {recs[:40]} lanes {lanes[:70]}; x tiles {xt[:16]}', flush=True)
    print(f'V={V} concurrent "{mode}", {ntr} trials: differing dwords in all: {tot}', flush=True)


if __name__ == '__main__':
  main()
