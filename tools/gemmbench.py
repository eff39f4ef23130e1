"""Time dyn_train_gemm on the training step's layer shapes.   python tools/gemmbench.py   (DYNIBAR_HIP_LIB selects a variant build)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynibar_amd import train_static as TS  # noqa: E402


def timeit(fn, n=5):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def main():
  dev = 'cuda:0'
  M = int(os.environ.get('GB_M', 3072 * 64 * 15))
  for K, N in ((256, 128), (128, 128), (104, 256), (128, 64), (256, 256)):
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.zeros(N, device=dev)
    Y = torch.empty(M, N, device=dev)
    dX = torch.empty(M, K, device=dev)
    dW = torch.zeros(N, K, device=dev)
    lin = TS._Lin(W, b)
    st = TS.stream_of(X)
    t_f = timeit(lambda: lin.fwd(st, X, 0, K, Y, 0, N, M, TS.ELU))
    t_d = timeit(lambda: TS._gemm(st, TS._p(Y), N, 1, TS._p(W), 1, K, TS._p(dX), K, M, K, N))
    t_w = timeit(lambda: TS._gemm(st, TS._p(Y), 1, N, TS._p(X), 1, K, TS._p(dW), K, N, K, M, accumulate=2, k_split=512))
    fl = 2.0 * M * K * N
    by_f = 4.0 * M * (K + N)
    print(f'K {K:4d} N {N:4d}: fwd {t_f:8.1f} us {fl / t_f / 1e6:6.1f} TF {by_f / t_f / 1e6:5.2f} TB/s | dgrad {t_d:8.1f} us {fl / t_d / 1e6:6.1f} TF {by_f / t_d / 1e6:5.2f} TB/s | '
          f'wgrad {t_w:8.1f} us {fl / t_w / 1e6:6.1f} TF {by_f / t_w / 1e6:5.2f} TB/s', flush=True)
    del X, Y, dX


if __name__ == '__main__':
  main()
