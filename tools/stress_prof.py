"""Per-kernel HIP-event breakdown of ONE BASELINE configs[4] chunk (8192 rays, 128 + 128 samples, 16 + 16 views).  Developer tool for the GPU box."""
import ctypes, sys, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tools'))
import config_cases
from dynibar_amd import _lib
L=_lib.lib()
dev=torch.device('cuda',0)
with torch.no_grad():
  stc=config_cases.StressChunk(dev)
  stc.render(); torch.cuda.synchronize()
  L.dyn_profile_enable(1)
  for _ in range(3): stc.render()
  torch.cuda.synchronize()
  L.dyn_profile_enable(0)
nk=L.dyn_profile_count(); ms=(ctypes.c_float*nk)(); cnt=(ctypes.c_int*nk)(); L.dyn_profile_read(ms,cnt)
tot=sum(ms[i] for i in range(nk))
for i in sorted(range(nk), key=lambda i:-ms[i]):
  if cnt[i]: print(f'{L.dyn_profile_name(i).decode():28s} {ms[i]/3:8.2f} ms  {cnt[i]//3:4d} launches  {100*ms[i]/tot:5.1f}%')
print('total', tot/3)
