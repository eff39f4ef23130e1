"""Developer check of the ragged dense-rows flavour at frame-chunk scale: the same DynibarStatic call (R rays x S samples x V views of the synthetic frame scene) with
DYN_RAGGED=1 and DYN_RAGGED=0 in two processes; the two `raw` tensors must agree to round-off.  python tools/ragged_ab.py [R S V]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from dynibar_amd import ops, synthetic as syn
R, S, V = RSV
H, W, F = 288, 512, 32
sc = syn.make_scene(seed=0, H=H, W=W, V=7, F=F, n_static=V)
T = lambda x: torch.from_numpy(x).cuda()
scene = {k: T(v) for k, v in sc.items()}
o, d, _ = syn.pixel_rays(sc['camera'], np.arange(R) * 17 %% (H * W))
o, d = T(o), T(d)
net = ops.StaticNet(syn.make_weights('static', 0, F), 'cuda:0', True, False)
views = ops.SourceViews(scene['camera'], scene['static_src_rgbs'], scene['static_src_cameras'], scene['static_featmaps'])
pts, z, _ = ops.sample_along_ray(o, d, scene['depth_range'], S, True, want_s=False)
rf, rd, mk, pm = ops.project_gather(views, R, S, ray_o=o, ray_d=d, z_vals=z, pix_mask_thresh=1.0)
raw = net(views, o, d, pts, rf, rd, mk)
raw2 = net(views, o, d, pts, rf, rd, mk)
torch.cuda.synchronize()
print('run-to-run max diff', float((raw - raw2).abs().max()))
np.save(OUT, raw.cpu().numpy()); np.save(OUT + '.mask.npy', mk.cpu().numpy())
'''
def main():
  rsv = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 64, 11)
  outs = []
  for rag in ('1', '0'):
    out = f'/tmp/ragged_ab_{rag}.npy'
    code = (CHILD % ROOT).replace('RSV', repr(rsv)).replace('OUT', repr(out))
    pr = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DYN_RAGGED=rag), capture_output=True, text=True)
    print(f'DYN_RAGGED={rag}:', pr.stdout.strip()[-200:], pr.stderr.strip()[-600:] if pr.returncode else '')
    outs.append(out)
  a, b = np.load(outs[0]), np.load(outs[1])
  mk = np.load(outs[0] + '.mask.npy')[..., 0]
  sig = (b[..., 3] > -1e8)
  dc = np.abs(a[..., :3] - b[..., :3]).max(-1)
  ds = np.where(sig & (a[..., 3] > -1e8), np.abs(a[..., 3] - b[..., 3]), (a[..., 3] != b[..., 3]) * 1e9)
  print('colour: max |ragged - regular|', dc.max(), ' points > 1e-5:', int((dc > 1e-5).sum()), 'of', dc.size)
  print('sigma : max |ragged - regular|', ds.max(), ' points > 1e-4:', int((ds > 1e-4).sum()))
  bad = np.argwhere((dc > 1e-5) | (ds > 1e-4))
  nv = mk.sum(-1)
  for r, s in bad[:20]:
    print(f'  ray {r} sample {s} point {r * rsv[1] + s}: n_valid {int(nv[r, s])} mask {mk[r, s].astype(int).tolist()} ragged {a[r, s]} regular {b[r, s]}')
  if len(bad):
    pts = bad[:, 0] * rsv[1] + bad[:, 1]
    print('  bad points mod 1024:', sorted(set((pts % 1024).tolist()))[:40], ' n_valid histogram of bad points:', np.bincount(nv[bad[:, 0], bad[:, 1]].astype(int), minlength=rsv[2] + 1).tolist())
if __name__ == '__main__':
  main()
