"""K1-only loop for rocprofv3 counter passes (developer tool)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynibar_amd import ops, synthetic as syn
R, S, V = 4096, 64, 8
dev = 'cuda:0'
sc = syn.make_scene(seed=0, V=V, n_static=V)
T = lambda x: torch.from_numpy(x).to(dev)
pix = syn.sample_pixels(0, 288, 512, R)
o, d, uv = syn.pixel_rays(sc['camera'], pix)
o, d = T(o), T(d)
views = ops.SourceViews(T(sc['camera']), T(sc['static_src_rgbs']), T(sc['static_src_cameras']), T(sc['static_featmaps']))
pts, z, s = ops.sample_along_ray(o, d, T(sc['depth_range']), S, True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
  out = ops.project_gather(views, R, S, ray_o=o, ray_d=d, z_vals=z)
torch.cuda.synchronize()
