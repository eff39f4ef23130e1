"""The bench step (BASELINE configs[1]: 4096 rays x 64 samples x 8 views, static branch) back to back for N seconds (developer tool: power / clock
sampling with rocm-smi beside it); the library is $DYNIBAR_HIP_LIB."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
step = bench.StaticStep('cuda:0', 4096, 64, 8)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
t0 = time.time(); n = 0
while time.time() - t0 < secs:
  for _ in range(50): step.step()
  torch.cuda.synchronize(); n += 50
print(f'{n} steps in {time.time() - t0:.2f} s: {(time.time() - t0) / n * 1e3:.3f} ms per step')
