"""k_motion_mlp alone (developer tool): python tools/motionbench.py [--rays 8192 --samples 128] tag[=lib.so] ...
Every variant runs in its own process (DYNIBAR_HIP_LIB), rounds alternate; reports microseconds per launch (HIP events around 10 launches),
algorithmic TFLOP/s (1.062 MFLOP per point) and the fraction of the 833 TFLOP/s split-product ceiling.  Variants built with
timing-only knobs (tools/experiments/r05_timing_only_switches.patch: B6D_NO_DMA, B6D_NO_BARRIER, B6D_NO_LDS) compute garbage: only their time is meaningful (the check column says so)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dynibar_amd', 'csrc')
CHILD = r'''
import json, sys, torch
sys.path.insert(0, %r)
import numpy as np
from dynibar_amd import ops, synthetic as syn
R, S = RAYS, SAMPLES
sd = {k: torch.from_numpy(v) for k, v in syn.make_weights('motion', seed=1).items()}
net = ops.MotionMLP(sd, 'cuda:0', num_basis=6)
g = torch.Generator().manual_seed(0)
pts = (torch.rand(R, S, 3, generator=g) * 4 - 2).cuda()
t = torch.tensor([0.37], device='cuda:0')
for _ in range(3): out = net(pts, t, 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = net(pts, t, 2)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100.0
# reference: fp64 torch on 256 points
W = {k: v.double().cuda() for k, v in sd.items()}
x = pts.reshape(-1, 3)[:256].double()
x4 = torch.cat([x, t.double().expand(256, 1)], 1)
fr = torch.linspace(1., 17., 16, dtype=torch.float32, device='cuda:0').double()
e = (x4[..., None] * fr).reshape(256, -1)
emb = torch.cat([x4, torch.sin(e), torch.cos(e)], -1)
h = emb
for i in range(8):
  h = torch.relu(torch.nn.functional.linear(h, W[f'pts_linears.{i}.weight'], W[f'pts_linears.{i}.bias']))
  if i == 4: h = torch.cat([emb, h], -1)
ref = torch.nn.functional.linear(h, W['coeff_linear.weight'], W['coeff_linear.bias'])
smp = torch.arange(256, device='cuda:0') %% S
ref = ref * (smp < S - 2)[:, None]
err = float((out.reshape(-1, 18)[:256].double() - ref).abs().max())
print('MBRESULT ' + json.dumps({'us': us, 'err': err}))
'''


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rays', type=int, default=8192)
  ap.add_argument('--samples', type=int, default=128)
  ap.add_argument('--rounds', type=int, default=2)
  ap.add_argument('variants', nargs='+')
  a = ap.parse_args()
  libs = {}
  for v in a.variants:
    tag, _, path = v.partition('=')
    libs[tag] = path or (os.path.join(CSRC, 'libdynibar_hip.so') if tag == 'base' else os.path.join(CSRC, f'libdynibar_hip_{tag}.so'))
  code = (CHILD % ROOT).replace('RAYS', str(a.rays)).replace('SAMPLES', str(a.samples))
  res = {t: [] for t in libs}
  for r in range(a.rounds):
    for tag, path in libs.items():
      if not os.path.exists(path):
        print(f'{tag}: no such library {path}', flush=True)
        continue
      pr = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DYNIBAR_HIP_LIB=path), capture_output=True, text=True, timeout=300)
      line = [l for l in pr.stdout.splitlines() if l.startswith('MBRESULT ')]
      if not line:
        print(f'{tag}: FAILED rc={pr.returncode}\n{pr.stderr[-800:]}', flush=True)
        continue
      o = json.loads(line[0][9:])
      res[tag].append(o)
      tf = 1.062e6 * a.rays * a.samples / (o['us'] * 1e-6) / 1e12
      print(f'round {r} {tag:10s} {o["us"]:9.1f} us  {tf:6.1f} TFLOP/s  {tf / 833.3:.3f} of 833  max err vs fp64 {o["err"]:.2e}', flush=True)
  print('SUMMARY (min over rounds)')
  for tag, rs in res.items():
    if rs:
      us = min(x['us'] for x in rs)
      print(f'  {tag:10s} {us:9.1f} us  {1.062e6 * a.rays * a.samples / (us * 1e-6) / 1e12 / 833.3:.3f} of 833  err {rs[0]["err"]:.1e}')


if __name__ == '__main__':
  main()
