import torch, time
dev='cuda:0'
def timeit(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1)/iters*1e-3
n=335*1024*1024//4
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
t=timeit(lambda: a.zero_()); print('fill 335MB: %.1f us  %.2f TB/s'%(t*1e6, n*4/t/1e12))
t=timeit(lambda: b.copy_(a)); print('copy 335MB: %.1f us  %.2f TB/s (r+w)'%(t*1e6, 2*n*4/t/1e12))
t=timeit(lambda: a.sum()); print('read 335MB: %.1f us  %.2f TB/s'%(t*1e6, n*4/t/1e12))
n2=n*4
a=torch.empty(n2,device=dev)
t=timeit(lambda: a.zero_()); print('fill 1340MB: %.1f us  %.2f TB/s'%(t*1e6, n2*4/t/1e12))
