"""k_motion_mlp back to back for N seconds (developer tool: power / clock sampling with rocm-smi beside it); the library is $DYNIBAR_HIP_LIB."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynibar_amd import ops, synthetic as syn
sd = {k: torch.from_numpy(v) for k, v in syn.make_weights('motion', seed=1).items()}
net = ops.MotionMLP(sd, 'cuda:0', num_basis=6)
pts = (torch.rand(8192, 128, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).cuda()
t = torch.tensor([0.37], device='cuda:0')
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
t0 = time.time(); n = 0
while time.time() - t0 < secs:
  for _ in range(20): out = net(pts, t, 2)
  torch.cuda.synchronize(); n += 20
print(f'{n} launches in {time.time() - t0:.2f} s: {(time.time() - t0) / n * 1e6:.1f} us per launch')
