"""Cycle stamps at the layer boundaries of the static view chain (developer tool).

  python tools/phasebench.py --build     (here: cross-compiles csrc/libdynibar_hip_phase.so with -DDYN_PHASE_TIMING)
  python tools/phasebench.py             (on the GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dynibar_amd', 'csrc')
TAG = os.environ.get('PHASE_TAG', '')
LIB = os.path.join(CSRC, 'libdynibar_hip_phase%s.so' % TAG)
NAMES = {0: 'start', 1: 'inputs+embed', 2: 'L1 ray_dir_fc.0 (52 steps x 8 tiles)', 3: 'elu(256)', 4: 'L2 ray_dir_fc.2 + ref mult', 5: 'weights',
         6: 'pool stats -> LDS', 7: 'base_fc.0 pooled tile', 8: 'res exchange', 9: 'base_fc.0 per-view', 10: 'elu(256)', 11: 'L4 base_fc.2',
         12: 'elu(128)+bias', 13: 'L5 vis_fc.0', 14: 'elu + L6 vis_fc.2(x)', 15: 'vis, x+=', 16: 'L7 vis_fc2.0', 17: 'elu, vis2', 18: 'store x',
         20: 'weighted stats + stores'}

if '--build' in sys.argv:
  hipcc = '/opt/rocm/bin/hipcc'
  common = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
  subprocess.check_call([hipcc] + common + ['-DDYN_PHASE_TIMING'] + os.environ.get('PHASE_FLAGS', '').split() + ['-c', os.path.join(CSRC, 'dyn_nets.hip'), '-o', os.path.join(CSRC, 'dyn_nets_phase%s.o' % TAG)])
  subprocess.check_call([hipcc] + common + ['-DDYN_PHASE_TIMING', '-ffp-contract=off'] + os.environ.get('PHASE_FLAGS', '').split() + ['-c', os.path.join(CSRC, 'dyn_geometry.hip'), '-o', os.path.join(CSRC, 'dyn_geometry_phase%s.o' % TAG)])
  subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', os.path.join(CSRC, 'dyn_geometry_phase%s.o' % TAG), os.path.join(CSRC, 'dyn_nets_phase%s.o' % TAG), os.path.join(CSRC, 'dyn_encoder.o'), os.path.join(CSRC, 'dyn_train.o'), os.path.join(CSRC, 'dyn_comm.o'), '-ldl', '-o', LIB])
  sys.exit(0)

os.environ['DYNIBAR_HIP_LIB'] = LIB
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dynibar_amd import ops, synthetic as syn  # noqa: E402

R, S, V, dev = 4096, 64, int(os.environ.get("PB_V", "8")), "cuda:0"
sc = syn.make_scene(seed=0, H=288, W=512, V=V, F=32, n_static=V)
T = lambda x: torch.from_numpy(x).to(dev)
scene = {k: T(v) for k, v in sc.items()}
o_np, d_np, _ = syn.pixel_rays(sc['camera'], syn.sample_pixels(100, 288, 512, R))
ray_o, ray_d = T(o_np), T(d_np)
net = ops.StaticNet(syn.make_weights('static', 0, 32), dev, anti_alias_pooling=True, mask_rgb=False)
views = ops.SourceViews(scene['camera'], scene['static_src_rgbs'], scene['static_src_cameras'], scene['static_featmaps'])
pts, z, _ = ops.sample_along_ray(ray_o, ray_d, scene['depth_range'], S, True, want_s=False)
rgb_feat, ray_diff, mask = ops.project_gather(views, R, S, ray_o=ray_o, ray_d=ray_d, z_vals=z)
for _ in range(3):
  net(views, ray_o, ray_d, pts, rgb_feat, ray_diff, mask)
torch.cuda.synchronize()
raw = ctypes.CDLL(LIB)
if hasattr(raw, 'dyn_debug_skew'):  # built with PHASE_FLAGS=-DDYN_PHASE_SKEW
  raw.dyn_debug_skew_reset()
  net(views, ray_o, ray_d, pts, rgb_feat, ray_diff, mask)
  torch.cuda.synchronize()
  sk = (ctypes.c_ulonglong * 32)()
  assert raw.dyn_debug_skew(sk) == 0
  print('view chain, one workgroup, per wave: cycles waiting for its own DMA pieces | cycles in the chunk barrier | chunks (all launches of the call that used ring kid 0)')
  for w in range(8):
    print('  wave %d: dma %8d   barrier %8d   chunks %d' % (w, sk[w * 4], sk[w * 4 + 1], sk[w * 4 + 2]))
pg = (ctypes.c_ulonglong * 16)()
if raw.dyn_debug_pg_phases(pg) == 0:
  for b in range(2):
    st = [pg[b * 8 + i] for i in range(5)]
    print('project_gather | workgroup', 'early' if b == 0 else 'middle', '| one wave, cycles: setup %d, projection + RGB taps + ray_diff %d, feature taps %d, LDS -> global stores %d, total %d' % (st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[4] - st[0]))
buf = (ctypes.c_ulonglong * (3 * 2 * 160))()
assert raw.dyn_debug_phases(buf) == 0
PNAMES = {1: {0: 'start', 1: 'geometry_fc (2 layers)', 2: 'Q, K, V projections', 3: 'attention (4 heads)', 4: 'fc + LayerNorm', 20: 'out_geometry_fc, rgb point part'},
          2: {0: 'start', 20: 'whole kernel'}}
for kid, kname in enumerate(('view chain (k_static_views)', 'point chain (k_net_points)', 'blend (k_static_blend)')):
  for b in range(2):
    base = (kid * 2 + b) * 160
    st = [buf[base + i] for i in range(32)]
    waits = [buf[base + 32 + i] for i in range(64)]
    print(kname, '| workgroup', 'first' if b == 0 else 'middle', '| total cycles', st[20] - st[0])
    names = NAMES if kid == 0 else PNAMES[kid]
    prev = st[0]
    for i in range(1, 21):
      if i in names and st[i]:
        print('  %-44s %8d' % (names[i], st[i] - prev))
        prev = st[i]
    nz = [w for w in waits if w]
    print('  cycles waited at each ring acquire (s_waitcnt vmcnt(0) + barrier):', nz, 'sum', sum(nz))
    at = [buf[base + 96 + i] for i in range(64) if buf[base + 96 + i]]
    print('  cycles between consecutive ring acquires (= per weight chunk):', [at[0] - st[0]] + [at[i + 1] - at[i] for i in range(len(at) - 1)])
