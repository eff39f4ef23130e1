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
sc = syn.make_scene(seed=0, H=a.H, W=a.W, V=a.vdy, n_static=a.vst)
fine = syn.make_scene(seed=0, H=a.H, W=a.W, V=a.vdy, n_static=a.vst, tag=1)
T = lambda x: torch.from_numpy(x).to(dev)
data = dict(camera=torch.from_numpy(sc['camera']), rgb_path='x', depth_range=torch.from_numpy(sc['depth_range']),
            src_rgbs=torch.from_numpy(sc['src_rgbs']), src_cameras=torch.from_numpy(sc['src_cameras']),
            static_src_rgbs=torch.from_numpy(sc['static_src_rgbs']), static_src_cameras=torch.from_numpy(sc['static_src_cameras']))
NF, NB = 24, 6
def dct(K, Tn):
  b = np.zeros((Tn, K), np.float32)
  for t in range(Tn):
    for k in range(1, K + 1):
      b[t, k - 1] = np.sqrt(2.0 / Tn) * np.cos(np.pi / (2.0 * Tn) * (2 * t + 1) * k)
  return torch.from_numpy(b)
model = types.SimpleNamespace(net_coarse_st=syn.make_weights('static', 0), net_coarse_dy=syn.make_weights('dynamic', 0),
                              net_fine_st=syn.make_weights('static', 100), net_fine_dy=syn.make_weights('dynamic', 100),
                              motion_mlp=syn.make_weights('motion', 0), motion_mlp_fine=syn.make_weights('motion', 100),
                              trajectory_basis=dct(NB, NF).to(dev), trajectory_basis_fine=dct(NB, NF).to(dev))
args = types.SimpleNamespace(anti_alias_pooling=True, mask_rgb=False, occ_weights_mode=0)
cfeat = (T(sc['featmaps']), None, T(sc['static_featmaps']))
ffeat = (T(fine['featmaps']), None, T(fine['static_featmaps']))
proj = projection.Projector(dev)
fidx, temb, toff = 11, torch.tensor([11 / 24.0], device=dev), [-3, -2, -1, 0, 1, 2, 3][:a.vdy]
lib = _lib.lib()
for f in range(a.frames + 1):
  if f == 1:
    lib.dyn_profile_enable(1)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  smp = sample_ray.RaySamplerSingleImage(data, dev)
  rb = smp.get_all()
  torch.cuda.synchronize(); t1 = time.perf_counter()
  ret = render_image.render_single_image_nvi((fidx, None), (temb, None), (toff, None), smp, rb, model, proj, a.chunk, 64, args, inv_uniform=True,
                                             N_importance=64, det=True, coarse_featmaps=cfeat, fine_featmaps=ffeat, is_train=False)
  torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f'frame {f}: sampler {1e3 * (t1 - t0):.1f} ms, render {1e3 * (t2 - t1):.1f} ms, {a.H * a.W / (t2 - t1):.0f} rays/s', flush=True)
lib.dyn_profile_enable(0)
nk = lib.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); lib.dyn_profile_read(ms, cnt)
tot = sum(ms[i] for i in range(nk))
print('kernel ms per frame (sum over chunks), launches:')
for i in sorted(range(nk), key=lambda i: -ms[i]):
  if cnt[i]:
    print(f'  {lib.dyn_profile_name(i).decode():24s} {ms[i] / a.frames:9.2f} ms  {cnt[i] // a.frames:5d}  {100 * ms[i] / tot:5.1f}%')
print(f'  total kernel time {tot / a.frames:.1f} ms per frame')
print('rgb', tuple(ret['outputs_fine_ref']['rgb'].shape), float(ret['outputs_fine_ref']['rgb'].mean()))
