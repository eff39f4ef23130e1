"""A/B timing of variant builds on the GPU box (developer tool):  python tools/abbench.py [--frame] [--iters N] tag[=lib.so] ...
Every variant runs in its own process (DYNIBAR_HIP_LIB) and reports, from the library's own HIP-event ring, the per-kernel times of the
bench step (BASELINE configs[1]: 4096 rays x 64 samples x 8 views) and -- with --frame -- of one Nvidia-eval frame.  'base' = the default
library.  Rounds alternate over the variants so that clock / thermal drift does not favour one of them."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dynibar_amd', 'csrc')

CHILD = r'''
import ctypes, json, os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tools'))
import bench
from dynibar_amd import _lib
L = _lib.lib()
def kernels():
  nk = L.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); L.dyn_profile_read(ms, cnt)
  return {L.dyn_profile_name(i).decode(): (ms[i], cnt[i]) for i in range(nk) if cnt[i]}
step = bench.StaticStep('cuda:0', 4096, 64, 8)
for _ in range(5): step.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(ITERS): step.step()
e1.record(); torch.cuda.synchronize()
L.dyn_profile_enable(1)
for _ in range(ITERS): step.step()
torch.cuda.synchronize()
L.dyn_profile_enable(0)
out = {'step_ms': e0.elapsed_time(e1) / ITERS, 'kernels': {k: m / c for k, (m, c) in kernels().items()}}
if FRAME:
  from frame_case import FrameCase
  fc = FrameCase('cuda:0', 288, 512, 7, 11, 8192)
  smp, rb = fc.sampler(); fc.render(smp, rb); torch.cuda.synchronize()
  e0.record(); fc.render(smp, rb); e1.record(); torch.cuda.synchronize()
  out['frame_ms'] = e0.elapsed_time(e1)
  from dynibar_amd import render_image
  render_image.CHUNK_STREAMS = 1  # (the per-kernel breakdown on one stream: overlapped kernels would be counted twice)
  L.dyn_profile_enable(1)
  fc.render(smp, rb); torch.cuda.synchronize()
  L.dyn_profile_enable(0)
  out['frame_kernels'] = {k: m for k, (m, c) in kernels().items()}
print('ABRESULT ' + json.dumps(out))
'''


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--frame', action='store_true')
  ap.add_argument('--iters', type=int, default=30)
  ap.add_argument('--rounds', type=int, default=2)
  ap.add_argument('variants', nargs='+')
  a = ap.parse_args()
  libs, envs = {}, {}
  for v in a.variants:
    tag, _, path = v.partition('=')
    if path.startswith('@'):  # tag=@NAME=VALUE[,NAME=VALUE]: the default library under these environment variables (e.g. noragged=@DYN_RAGGED=0)
      envs[tag] = dict(kv.split('=', 1) for kv in path[1:].split(','))
      path = ''
    libs[tag] = path or (os.path.join(CSRC, 'libdynibar_hip.so') if (tag == 'base' or tag in envs) else os.path.join(CSRC, f'libdynibar_hip_{tag}.so'))
  res = {t: [] for t in libs}
  code = (CHILD % (ROOT, ROOT)).replace('ITERS', str(a.iters)).replace('FRAME', 'True' if a.frame else 'False')
  for r in range(a.rounds):
    for tag, path in libs.items():
      env = dict(os.environ, DYNIBAR_HIP_LIB=path, **envs.get(tag, {}))
      pr = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
      line = [l for l in pr.stdout.splitlines() if l.startswith('ABRESULT ')]
      if not line:
        print(f'{tag}: FAILED rc={pr.returncode}\n{pr.stderr[-1500:]}', flush=True)
        continue
      res[tag].append(json.loads(line[0][9:]))
      o = res[tag][-1]
      ks = ' '.join(f'{k[2:] if k.startswith("k_") else k}={v * 1e3:.0f}' for k, v in sorted(o['kernels'].items(), key=lambda kv: -kv[1])[:5])
      fr = ''
      if 'frame_ms' in o:
        fr = f' | frame {o["frame_ms"]:.1f} ms: ' + ' '.join(f'{k[2:]}={v:.1f}' for k, v in sorted(o['frame_kernels'].items(), key=lambda kv: -kv[1])[:7])
      print(f'round {r} {tag:12s} step {o["step_ms"]:.3f} ms  us: {ks}{fr}', flush=True)
  print('SUMMARY (min over rounds)')
  for tag, rs in res.items():
    if rs:
      print(f'  {tag:12s} step {min(x["step_ms"] for x in rs):.3f} ms  views {min(x["kernels"].get("k_static_views", 0) for x in rs) * 1e3:.0f} us' +
            (f'  frame {min(x["frame_ms"] for x in rs):.1f} ms' if 'frame_ms' in rs[0] else ''))


if __name__ == '__main__':
  main()
