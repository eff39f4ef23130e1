"""Time one static bootstrap training step (forward + backward through the dyn_train_* kernels) at the reference's training shape
(configs/train_kid-running.txt: N_rand 3072, 64 samples, 15 static views, anti_alias_pooling 0, mask_rgb 1).   python tools/trainbench.py [R]"""
import json
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from dynibar_amd import _lib, ops, synthetic as syn, train_motion as TM, train_static as TS  # noqa: E402


def full_iteration(R):
  """the reference's whole main-loop iteration (train.py:203-467) at the kid-running shape"""
  import ctypes
  from train_case import TrainCase
  tc = TrainCase('cuda:0', R=R)
  for _ in range(2):
    tc.step()
  torch.cuda.synchronize()
  _lib.lib().dyn_profile_enable(1)
  n = 3
  t0 = time.perf_counter()
  for _ in range(n):
    tc.step()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / n * 1e3
  cnt = _lib.lib().dyn_profile_count()
  tot = (ctypes.c_float * cnt)()
  lau = (ctypes.c_int * cnt)()
  _lib.lib().dyn_profile_read(tot, lau)
  _lib.lib().dyn_profile_name.restype = ctypes.c_char_p
  kern = {_lib.lib().dyn_profile_name(i).decode(): (round(tot[i] / n, 3), lau[i] // n) for i in range(cnt) if lau[i]}
  fl = tc.algorithmic_flops()
  print(json.dumps(dict(what='full training iteration (render_rays_mono is_train=True + backward)', R=R, S=tc.S, ms_per_step=round(ms, 2),
                        rays_per_s=round(R / ms * 1e3), algorithmic_tflop_per_step=round(fl / 1e12, 3), tflops=round(fl / ms / 1e9, 1),
                        peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2), kernels_ms_launches=kern)))


def main():
  if len(sys.argv) > 2 and sys.argv[2] == 'full':
    return full_iteration(int(sys.argv[1]))
  R = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
  S, V = 64, 15
  dev = 'cuda:0'
  sc = syn.make_scene(seed=21, H=288, W=512, V=7, n_static=V, smooth=False)
  t = lambda x: torch.from_numpy(x).to(dev)
  fm = t(sc['static_featmaps']).requires_grad_(True)
  views = ops.SourceViews(t(sc['camera']), t(sc['static_src_rgbs']), t(sc['static_src_cameras']), fm.detach())
  pix = syn.sample_pixels(21, 288, 512, R)
  o, d, _ = syn.pixel_rays(sc['camera'], pix)
  o, d = t(o), t(d)
  prm = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in syn.make_weights('static', 0).items() if k != 's'}
  dr = t(sc['depth_range'])
  cot = torch.randn(R, 3, device=dev)

  def step():
    pts, z, _ = ops.sample_along_ray(o, d, dr, S, True)
    rgb_feat, ray_diff, mask, pm = TM.gather(views, fm, R, S, ray_o=o, ray_d=d, z_vals=z, pix_mask_thresh=1.0)
    raw = TS.static_raw(prm, (False, True), views, rgb_feat, o, d, pts, ray_diff, mask)
    out = TS.composite_vanilla(raw, z, pm)
    loss = (out['rgb'] * cot).sum()
    loss.backward()

  for _ in range(2):
    step()
  torch.cuda.synchronize()
  _lib.lib().dyn_profile_enable(1)
  n = 5
  t0 = time.perf_counter()
  for _ in range(n):
    step()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / n * 1e3
  import ctypes
  cnt = _lib.lib().dyn_profile_count()
  tot = (ctypes.c_float * cnt)()
  lau = (ctypes.c_int * cnt)()
  _lib.lib().dyn_profile_read(tot, lau)
  _lib.lib().dyn_profile_name.restype = ctypes.c_char_p
  kern = {_lib.lib().dyn_profile_name(i).decode(): (round(tot[i] / n, 3), lau[i] // n) for i in range(cnt) if lau[i]}
  # algorithmic FLOPs of the step: forward static net x 3 (forward, data gradient, weight gradient)
  fwd = (0.361e6 + 0.033e6 * (S / 64) + 0.4305e6 * V) * R * S
  print(json.dumps(dict(what='static bootstrap training step', R=R, S=S, V=V, ms_per_step=round(ms, 2), rays_per_s=round(R / ms * 1e3),
                        algorithmic_tflop_per_step=round(3 * fwd / 1e12, 3), tflops=round(3 * fwd / ms / 1e9, 1),
                        peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2), kernels_ms_launches=kern)))


if __name__ == '__main__':
  main()
