import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import parity
from parity import *
dev = 'cuda:0'
for kw in (dict(name='small', S=64), dict(name='noise', S=64), dict(name='small', S=64, weights='trained'), dict(name='harsh', S=64)):
  name, S, weights = kw['name'], kw['S'], kw.get('weights', 'init')
  for aa in (True, False):
    scene, o, d, sd32, v32, _, _, keep = train_static_reference(name, S, None, aa, False, weights)
    _, _, _, _, v64, _, _, keep64 = train_static_reference(name, S, None, aa, False, weights, dtype=torch.float64)
    keep = keep & keep64
    net = ops.StaticNet(parity._weights(weights)['net_coarse_st'], dev, aa, False)
    out, raw = run_static_pass(dev, to_dev(scene, dev), net, o.float().to(dev), d.float().to(dev), S)
    r = cpu(raw)[keep].double(); t = v64['raw'][keep]; f = v32['raw'][keep].double()
    live = t[..., 3] > -1e8
    for nm, sl in (('rgb', slice(0, 3)), ('sigma', slice(3, 4))):
      eo = (r[..., sl] - t[..., sl]).abs()[live].flatten(); er = (f[..., sl] - t[..., sl]).abs()[live].flatten()
      q = lambda e, p: float(torch.quantile(e, p))
      print(f'{name} {weights} aa={int(aa)} raw {nm}: ours p50 {q(eo,.5):.2e} p90 {q(eo,.9):.2e} p99 {q(eo,.99):.2e} max {float(eo.max()):.2e} | ref32 p50 {q(er,.5):.2e} p90 {q(er,.9):.2e} p99 {q(er,.99):.2e} max {float(er.max()):.2e}  scale {float(t[..., sl][live].abs().max()):.2f}')
