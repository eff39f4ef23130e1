"""K1 (k_project_gather_tile) over view counts and tile shapes (developer tool): python tools/k1sweep.py
Each (P override) runs in its own process (DYN_PG_P is read once per process); reports us per launch and the fraction of 8 TB/s on the algorithmic bytes
of SURVEY section 8d, for the static form (points from rays) and the dynamic form (per-view displaced points xyz, +12 B per point-view read)."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, json, sys, torch
sys.path.insert(0, %r)
from dynibar_amd import ops, synthetic as syn, _lib
L = _lib.lib()
R, S, dev = 4096, 64, 'cuda:0'
out = {}
for V in VIEWS:
  sc = syn.make_scene(seed=0, V=V, n_static=V)
  T = lambda x: torch.from_numpy(x).to(dev)
  o, d, uv = syn.pixel_rays(sc['camera'], syn.sample_pixels(0, 288, 512, R))
  o, d = T(o), T(d)
  views = ops.SourceViews(T(sc['camera']), T(sc['static_src_rgbs']), T(sc['static_src_cameras']), T(sc['static_featmaps']))
  pts, z, s = ops.sample_along_ray(o, d, T(sc['depth_range']), S, True)
  xyz = (pts[None] + 0.01 * torch.randn(V, R, S, 3, device=dev)).contiguous()
  big = torch.empty(1 << 28, device=dev)  # 1 GiB streamed between launches: the source maps leave the caches like in the pipeline
  for form in ('static', 'dynamic'):
    kw = dict(ray_o=o, ray_d=d, z_vals=z) if form == 'static' else dict(pts_st=pts, xyz=xyz)
    for _ in range(3): ops.project_gather(views, R, S, pix_mask_thresh=1.0, **kw)
    torch.cuda.synchronize()
    L.dyn_profile_enable(1)
    for _ in range(12):
      big.add_(1.0)
      ops.project_gather(views, R, S, pix_mask_thresh=1.0, **kw)
    torch.cuda.synchronize()
    nk = L.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); L.dyn_profile_read(ms, cnt); L.dyn_profile_enable(0)
    t = [ms[i] / cnt[i] for i in range(nk) if cnt[i] > 0 and L.dyn_profile_name(i).decode() == 'k_project_gather'][0] * 1e-3
    b = R * S * V * 160 + V * (views.Hf * views.Wf * views.F + views.H * views.W * 3) * 4 + R * (24 + 4 * S) + (R * S * V * 12 if form == 'dynamic' else 0)
    out[f'V{V}_{form}'] = (t * 1e6, b / t / 8e12)
print('K1RESULT ' + json.dumps(out))
'''
views = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '7,8,11,15'.split(','))]
# optional: argv[2] = comma list of tile overrides ('' = the library's rule), argv[3] = comma list of library tags (libdynibar_hip_<tag>.so; 'base' = default)
plist = sys.argv[2].split(',') if len(sys.argv) > 2 else ['', '32', '64']
libs = sys.argv[3].split(',') if len(sys.argv) > 3 else ['base']
for lib, P in [(l, q) for l in libs for q in plist]:
  env = dict(os.environ)
  if lib != 'base':
    env['DYNIBAR_HIP_LIB'] = os.path.join(ROOT, 'dynibar_amd', 'csrc', f'libdynibar_hip_{lib}.so')
    print(f'[{lib}]', end=' ')
  if P: env['DYN_PG_P'] = P
  pr = subprocess.run([sys.executable, '-c', (CHILD % ROOT).replace('VIEWS', repr(views))], env=env, capture_output=True, text=True, timeout=600)
  line = [l for l in pr.stdout.splitlines() if l.startswith('K1RESULT ')]
  if not line:
    print(f'P={P or "auto"}: FAILED\n{pr.stderr[-1200:]}'); continue
  o = json.loads(line[0][9:])
  print(f'P={P or "auto":4s} ' + '  '.join(f'{k}: {v[0]:6.1f} us {v[1]:.3f}' for k, v in o.items()), flush=True)
