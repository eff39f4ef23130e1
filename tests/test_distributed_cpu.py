"""world_size-2 gloo tests (CPU) of the N > 1 path: ray-tile partitioning, the reference's chunk slicing rules and the
all-gather that reassembles a frame must reproduce the single-process result exactly (tiling must not change any per-ray value)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_render(chunk):
  """A per-ray function with the output structure of render_rays_mv: 1-D, 2-D, bool, [V,R,2] and 4-D (kept as per-chunk lists) tensors."""
  o = chunk['ray_o']
  R = o.shape[0]
  s = torch.arange(5, dtype=torch.float32)[None, :]
  return {'outputs_coarse_ref': {'rgb': o * 2.0 + 1.0, 'depth': o.sum(dim=1), 'weights': o[:, :1] * s, 'mask': o[:, 0] > 0.3,
                                 'render_flows': torch.stack([o[:, :2], -o[:, :2]], dim=0), 'dropme': torch.zeros(2, R, 3, 1)},
          'outputs_fine_ref': {'rgb': o * 3.0, 'mask': o[:, 1] > 0.5, 'depth': o[:, 2]}}


def run_frame(n_rays, chunk_size):
  from dynibar_amd import render_image as RI
  g = torch.Generator().manual_seed(5)
  batch = {'ray_o': torch.rand(n_rays, 3, generator=g), 'camera': torch.zeros(1, 34), 'flows': torch.rand(6, n_rays, 2, generator=g), 'rgb': None}
  seen = []

  def render_chunk(chunk):
    assert chunk['camera'].shape == (1, 34) and chunk['flows'].shape[1] == chunk['ray_o'].shape[0] and chunk['rgb'] is None
    seen.append(chunk['ray_o'].shape[0])
    return fake_render(chunk)

  chunks, n, d, world, rank = RI._render_tiles(batch, chunk_size, render_chunk, ('outputs_coarse_ref', 'outputs_fine_ref'))
  H, W = 7, n_rays // 7
  out = {g_: RI._assemble(chunks[g_], n, H, W, d, world, rank) for g_ in chunks}
  return out, seen


def worker(rank, world, port, n_rays, chunk_size, q):
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    out, seen = run_frame(n_rays, chunk_size)
    q.put((rank, {g: {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in d.items()} for g, d in out.items()}, seen))
  finally:
    dist.destroy_process_group()


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.parametrize('n_rays,chunk', [(7 * 9, 10), (7 * 13, 64)])
def test_two_rank_frame_equals_single_process(n_rays, chunk):
  sys.path.insert(0, ROOT)
  ref, seen1 = run_frame(n_rays, chunk)
  assert sum(seen1) == n_rays
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = free_port()
  procs = [ctx.Process(target=worker, args=(r, 2, port, n_rays, chunk, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert sum(sum(s) for _, _, s in got) == n_rays, 'every ray is rendered exactly once across the ranks'
  for rank, out, _ in got:
    for grp in ref:
      assert list(out[grp].keys()) == list(ref[grp].keys())
      for k in ref[grp]:
        if isinstance(ref[grp][k], list):
          # 4-D entries stay per-chunk lists (as the reference leaves them); under tiling a rank holds the chunks of its own tile
          assert isinstance(out[grp][k], list) and all(t.dim() == 4 for t in out[grp][k])
          continue
        assert torch.equal(out[grp][k], ref[grp][k]), f'rank {rank} {grp}/{k} differs from the single-process frame'
  assert isinstance(ref['outputs_coarse_ref']['dropme'], list), '4-D entries are kept as per-chunk lists, never assembled'
  for grp in ('outputs_coarse_ref',):  # the two ranks' chunk lists together are the single-process list
    both = [t for rank, out, _ in sorted(got, key=lambda g: g[0]) for t in out[grp]['dropme']]
    assert torch.equal(torch.cat(both, dim=1), torch.cat(ref[grp]['dropme'], dim=1))
  # masked pixels are blanked after the gather, like render_image.py:186-188
  m = ref['outputs_coarse_ref']['mask']
  assert bool((ref['outputs_coarse_ref']['rgb'][m == 0] == 0).all())


def test_ray_tiles_partition():
  from dynibar_amd.render_image import ray_tile
  for n in (1, 7, 64, 147456, 147457):
    for world in (1, 2, 3, 8):
      spans = [ray_tile(n, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
      assert all(hi - lo <= tile for lo, hi, tile in spans)
