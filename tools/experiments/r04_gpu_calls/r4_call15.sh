cd /root/repo; O=gpurun_out
python - <<'PY' > $O/r4c15_eval.txt 2>&1
import sys, ctypes, torch
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import eval_loop
from dynibar_amd import _lib
L = _lib.lib()
loop = eval_loop.EvalLoop('cuda:0')
loop.one_view()
def kernels():
  nk = L.dyn_profile_count(); ms = (ctypes.c_float * nk)(); cnt = (ctypes.c_int * nk)(); L.dyn_profile_read(ms, cnt)
  return {L.dyn_profile_name(i).decode(): round(ms[i], 2) for i in range(nk) if cnt[i]}
for rep in range(2):
  L.dyn_profile_enable(1)
  t, p = loop.one_view()
  torch.cuda.synchronize(); L.dyn_profile_enable(0)
  print('eval view', {k: round(v, 1) for k, v in t.items()}, kernels())
fc = loop.fc
from frame_case import FrameCase
fc2 = FrameCase('cuda:0')
smp, rb = fc2.sampler(); fc2.render(smp, rb); torch.cuda.synchronize()
import time
for rep in range(2):
  L.dyn_profile_enable(1)
  t0 = time.perf_counter(); fc2.render(smp, rb); torch.cuda.synchronize(); dt = time.perf_counter() - t0
  L.dyn_profile_enable(0)
  print('synthetic-map frame', round(dt * 1e3, 1), kernels())
PY
cat $O/r4c15_eval.txt | cut -c1-900
