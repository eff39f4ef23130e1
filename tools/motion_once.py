"""A few launches of k_motion_mlp (8192 rays x 128 samples) for rocprofv3 --pmc runs (developer tool); the library is $DYNIBAR_HIP_LIB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynibar_amd import ops, synthetic as syn
sd = {k: torch.from_numpy(v) for k, v in syn.make_weights('motion', seed=1).items()}
net = ops.MotionMLP(sd, 'cuda:0', num_basis=6)
pts = (torch.rand(8192, 128, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).cuda()
t = torch.tensor([0.37], device='cuda:0')
for _ in range(6): out = net(pts, t, 2)
torch.cuda.synchronize()
