"""Per-kernel micro-timing on the GPU (developer tool; bench.py is the contract bench)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynibar_amd import ops, synthetic as syn  # noqa: E402


def timeit(fn, iters=20, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e-3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--R', type=int, default=4096)
  ap.add_argument('--S', type=int, default=64)
  ap.add_argument('--V', type=int, default=8)
  a = ap.parse_args()
  dev = 'cuda:0'
  sc = syn.make_scene(seed=0, V=a.V, n_static=a.V)
  T = lambda x: torch.from_numpy(x).to(dev)
  pix = syn.sample_pixels(0, 288, 512, a.R)
  o, d, uv = syn.pixel_rays(sc['camera'], pix)
  o, d = T(o), T(d)
  views = ops.SourceViews(T(sc['camera']), T(sc['static_src_rgbs']), T(sc['static_src_cameras']), T(sc['static_featmaps']))
  dr = T(sc['depth_range'])
  res = {}
  pts, z, s = ops.sample_along_ray(o, d, dr, a.S, True)
  res['sample_along_ray_us'] = timeit(lambda: ops.sample_along_ray(o, d, dr, a.S, True)) * 1e6
  t = timeit(lambda: ops.project_gather(views, a.R, a.S, ray_o=o, ray_d=d, z_vals=z))
  import ctypes
  from dynibar_amd import _lib
  L = _lib.lib(); L.dyn_profile_enable(1)
  for _ in range(20): ops.project_gather(views, a.R, a.S, ray_o=o, ray_d=d, z_vals=z)
  nk = L.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); L.dyn_profile_read(ms, cnt); L.dyn_profile_enable(0)
  t = [ms[i] / cnt[i] for i in range(nk) if cnt[i] > 0][0] * 1e-3
  H, W, Hf, Wf, F = views.H, views.W, views.Hf, views.Wf, views.F
  bytes_alg = a.R * a.S * a.V * 160 + a.V * (Hf * Wf * F + H * W * 3) * 4 + a.R * (24 + 4 * a.S)
  res['project_gather_us'] = t * 1e6
  res['project_gather_GBps'] = bytes_alg / t / 1e9
  res['project_gather_frac_8TBps'] = bytes_alg / t / 8e12
  raw = torch.randn(a.R, a.S, 4, device=dev)
  pm = torch.ones(a.R, a.S, device=dev)
  res['composite_vanilla_us'] = timeit(lambda: ops.composite(raw, z, pm)) * 1e6
  res['composite_dual_us'] = timeit(lambda: ops.composite(raw, z, pm, raw, pm)) * 1e6
  w = torch.rand(a.R, a.S, device=dev)
  res['fine_samples_us'] = timeit(lambda: ops.fine_samples(z, w, a.S, True)) * 1e6
  print(json.dumps(res, indent=1))


if __name__ == '__main__':
  main()
