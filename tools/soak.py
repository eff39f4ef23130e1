"""Frames back to back for N seconds with the per-kernel times of every frame and rocm-smi readings beside them (developer tool: looks for the
two-state behaviour of the one-wave-per-SIMD kernels)."""
import ctypes, os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from dynibar_amd import _lib
from frame_case import FrameCase
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
L = _lib.lib()
fc = FrameCase('cuda:0')
smp, rb = fc.sampler(); fc.render(smp, rb); torch.cuda.synchronize()
smi = {'txt': ''}
stop = False
def sampler():
  while not stop:
    try:
      out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=10).stdout
      w = re.search(r'Package Power \(W\):\s*([0-9.]+)', out); s = re.search(r'sclk clock level:\s*\d+:\s*\((\d+)Mhz\)', out)
      f = re.search(r'fclk clock level:\s*\d+:\s*\((\d+)Mhz\)', out); t = re.findall(r'Temperature \(Sensor (\w+)\) \(C\):\s*([0-9.]+)', out)
      smi['txt'] = f"{w.group(1) if w else '?'} W sclk {s.group(1) if s else '?'} fclk {f.group(1) if f else '?'} " + ' '.join(f'{a}={b}' for a, b in t)
    except Exception as e:
      smi['txt'] = 'smi failed: ' + str(e)[:60]
    time.sleep(2.0)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
def kernels():
  nk = L.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); L.dyn_profile_read(ms, cnt)
  return {L.dyn_profile_name(i).decode()[2:]: ms[i] for i in range(nk) if cnt[i]}
while time.time() - t0 < secs:
  L.dyn_profile_enable(1)
  a = time.perf_counter(); fc.render(smp, rb); torch.cuda.synchronize(); dt = time.perf_counter() - a
  L.dyn_profile_enable(0)
  k = kernels(); n += 1
  if n % 4 == 1:
    print(f't={time.time() - t0:5.1f}s frame {dt * 1e3:6.1f} ms  static_points {k.get("static_points", 0):5.1f} dynamic_points {k.get("dynamic_points", 0):5.1f} motion {k.get("motion_mlp", 0):5.1f} '
          f'static_views {k.get("static_views", 0):6.1f} blend {k.get("static_blend", 0):5.1f} | {smi["txt"]}', flush=True)
stop = True; th.join()
