"""Child of tests/test_gpu_dist.py (not a test module): ONE rank under torch.distributed.run on cuda:0 with the RCCL backend.  The frame functions take
their multi-rank code path (ray tile, packed pixel rows, the all-gather, deferred entries resolved through collectives) with a process group of one
rank, once through torch.distributed.all_gather_into_tensor and once through the C-ABI's dyn_gather_tiles on a communicator made by dyn_comm_init_rank,
and must return the real reference's frames (tests/golden/image_nvi.npz, image_mono.npz) both ways.  Also: the ABI gather against torch's on a raw buffer."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import parity  # noqa: E402
from dynibar_amd import _lib, render_image  # noqa: E402


def main():
  dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
  torch.cuda.set_device(dev)
  dist.init_process_group('nccl', device_id=dev)
  world, rank = dist.get_world_size(), dist.get_rank()
  assert dist.get_backend() == 'nccl'
  L = _lib.lib()
  assert L.dyn_comm_available() == 1, 'RCCL must be resolvable from the process image (torch has loaded it)'
  # ---- the collective alone: dyn_gather_tiles == all_gather_into_tensor, on the compute stream
  comm = render_image.abi_communicator(dist, world, rank, dev)
  n, r = ctypes.c_int(-1), ctypes.c_int(-1)
  _lib.call('dyn_comm_size_rank', comm, ctypes.byref(n), ctypes.byref(r))
  assert (n.value, r.value) == (world, rank)
  send = torch.randn(1237, 5, device=dev)
  a = torch.empty(world * 1237, 5, device=dev)
  b = torch.full_like(a, float('nan'))
  dist.all_gather_into_tensor(a, send)
  _lib.call('dyn_gather_tiles', _lib.ptr(send), _lib.ptr(b), 1237, 5, comm, _lib.stream_of(send))
  torch.cuda.synchronize()
  assert torch.equal(a, b), 'dyn_gather_tiles differs from all_gather_into_tensor'
  dist.barrier()
  torch.cuda.synchronize()
  # ---- the frame functions on the multi-rank path
  render_image.FORCE_DIST = True
  gdir = os.path.join(HERE, 'golden')
  for mode in ('torch', 'abi'):
    render_image.GATHER = mode
    d, w, rk = render_image._dist()
    assert d is not None and w == world and rk == rank, 'the frame functions must see the process group'
    parity.check_render_image_nvi(str(dev), dict(np.load(os.path.join(gdir, 'image_nvi.npz'))))
    parity.check_render_image_mono(str(dev), dict(np.load(os.path.join(gdir, 'image_mono.npz'))))
    torch.cuda.synchronize()
    print(f'DIST_ONE_RANK frame functions ok through gather={mode}', flush=True)
  dist.barrier()
  torch.cuda.synchronize()
  dist.destroy_process_group()
  print('DIST_ONE_RANK_OK', flush=True)


if __name__ == '__main__':
  main()
