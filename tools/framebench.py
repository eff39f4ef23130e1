"""Full-frame render_single_image_nvi at the Nvidia Balloon1 eval shape (BASELINE configs[2] on one GPU): 288x512 rays, 64 coarse +
64 fine samples, 7 dynamic + 11 static source views, chunk 8192 (eval_nvidia.py:360-378).  Developer tool: prints wall time,
rays/s and the per-kernel HIP-event breakdown."""
import argparse, ctypes, json, os, sys, time, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynibar_amd import _lib, projection, render_image, sample_ray, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument('--H', type=int, default=288); ap.add_argument('--W', type=int, default=512)
ap.add_argument('--vdy', type=int, default=7); ap.add_argument('--vst', type=int, default=11)
ap.add_argument('--chunk', type=int, default=8192); ap.add_argument('--frames', type=int, default=2)
a = ap.parse_args()
dev = 'cuda:0'
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from frame_case import FrameCase
fc = FrameCase(dev, a.H, a.W, a.vdy, a.vst, a.chunk)
lib = _lib.lib()
print(f'chunk streams: {render_image.CHUNK_STREAMS} (the per-kernel breakdown below is taken on ONE stream: overlapped kernels would be counted twice)', flush=True)
for f in range(2 * a.frames + 1):
  if f == a.frames + 1:
    render_image.CHUNK_STREAMS = 1
    lib.dyn_profile_enable(1)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  smp, rb = fc.sampler()
  torch.cuda.synchronize(); t1 = time.perf_counter()
  ret = fc.render(smp, rb)
  torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f'frame {f}{" (one stream, kernels timed)" if f > a.frames else ""}: sampler {1e3 * (t1 - t0):.1f} ms, render {1e3 * (t2 - t1):.1f} ms, {a.H * a.W / (t2 - t1):.0f} rays/s', flush=True)
lib.dyn_profile_enable(0)
nk = lib.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); lib.dyn_profile_read(ms, cnt)
tot = sum(ms[i] for i in range(nk))
print('kernel ms per frame (sum over chunks), launches:')
for i in sorted(range(nk), key=lambda i: -ms[i]):
  if cnt[i]:
    print(f'  {lib.dyn_profile_name(i).decode():24s} {ms[i] / a.frames:9.2f} ms  {cnt[i] // a.frames:5d}  {100 * ms[i] / tot:5.1f}%')
print(f'  total kernel time {tot / a.frames:.1f} ms per frame')
print('rgb', tuple(ret['outputs_fine_ref']['rgb'].shape), float(ret['outputs_fine_ref']['rgb'].mean()))
