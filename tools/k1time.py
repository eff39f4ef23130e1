"""K1-only timing (HIP events from the library's own per-kernel ring): python tools/k1time.py [V] [iters]; A/B builds via DYNIBAR_HIP_LIB."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynibar_amd import _lib, ops, synthetic as syn
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
R, S = 4096, 64
dev = 'cuda:0'
sc = syn.make_scene(seed=0, V=V, n_static=V)
T = lambda x: torch.from_numpy(x).to(dev)
pix = syn.sample_pixels(0, 288, 512, R)
o, d, uv = syn.pixel_rays(sc['camera'], pix)
o, d = T(o), T(d)
views = ops.SourceViews(T(sc['camera']), T(sc['static_src_rgbs']), T(sc['static_src_cameras']), T(sc['static_featmaps']))
pts, z, s = ops.sample_along_ray(o, d, T(sc['depth_range']), S, True)
for _ in range(5):
  out = ops.project_gather(views, R, S, ray_o=o, ray_d=d, z_vals=z, pix_mask_thresh=1.0)
torch.cuda.synchronize()
L = _lib.lib(); L.dyn_profile_enable(1)
trash = os.environ.get('TRASH', 'none')  # what runs between two gathers: nothing | a 1.5 GB fill (dirty lines) | a 1.5 GB read (clean lines)
big = torch.empty(1536 * 1024 * 1024 // 4, device=dev) if trash != 'none' else None
if big is not None: big.zero_()
for _ in range(it):
  if trash == 'fill': big.zero_()
  elif trash == 'read': big.sum()
  if os.environ.get('TOUCH'):  # bring the source maps back into the memory-side cache before the gather
    views.feat_cl.sum(); views.src_rgbs.sum()
  out = ops.project_gather(views, R, S, ray_o=o, ray_d=d, z_vals=z, pix_mask_thresh=1.0)
torch.cuda.synchronize()
nk = L.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); L.dyn_profile_read(ms, cnt)
nbytes = R * S * V * 160 + V * (72 * 128 * 32 + 288 * 512 * 3) * 4 + R * (24 + 4 * S)
for i in range(nk):
  if cnt[i]:
    us = ms[i] / cnt[i] * 1e3
    print(f'{os.environ.get("TAG", "")} trash={trash} V={V} {L.dyn_profile_name(i).decode()} {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s algorithmic')
