"""world_size-2 and -3 gloo tests (CPU) of the N > 1 path.  The REAL ``render_single_image_nvi`` / ``_mono`` run in every rank -- ray-tile
partitioning, the reference's chunk slicing rules, the packed all-gather, the deferred per-sample entries, empty tiles -- over a stub
of the kernel layer (``render_rays_mv`` replaced by a cheap per-ray function with the same output structure); the tiled frame must equal
the single-process frame bit for bit (tiling must not change any per-ray value).  Results travel back as numpy arrays: a torch tensor on
an mp.Queue is a file-descriptor handle that dies with its producer."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = []


def stub_render_rays_mv(frame_idx, time_embedding, time_offset, ray_batch, model, projector, coarse_featmaps, fine_featmaps, N_samples, args,
                        inv_uniform=False, N_importance=0, raw_noise_std=0.0, det=False, white_bkgd=False, is_train=True):
  """A per-ray function with the output structure of render_rays_mv: 1-D, 2-D, bool, [V,R,2] and 4-D (kept as per-chunk lists) tensors."""
  chunk = ray_batch
  assert chunk['camera'].shape == (1, 34) and chunk['flows'].shape[1] == chunk['ray_o'].shape[0] and chunk['rgb'] is None
  o = chunk['ray_o']
  R = o.shape[0]
  CALLS.append(R)
  s = torch.arange(5, dtype=torch.float32)[None, :]
  coarse = {'rgb': o * 2.0 + 1.0, 'depth': o.sum(dim=1), 'weights': o[:, :1] * s, 'mask': o[:, 0] > 0.3,
            'render_flows': torch.stack([o[:, :2], -o[:, :2]], dim=0), 'dropme': torch.zeros(2, R, 3, 1) + o[None, :, :, None]}
  fine = {'rgb': o * 3.0, 'mask': o[:, 1] > 0.5, 'depth': o[:, 2], 'alpha': o[:, 1:2] * s}
  return {'outputs_coarse_ref': coarse, 'outputs_fine_ref': fine, 'outputs_fine_anchor': None, 'outputs_fine_anchor_dy': None}


def run_frame(H, W, chunk_size, mode, read=('weights', 'render_flows', 'rgb', 'dropme')):
  from dynibar_amd import render_image as RI
  RI.render_rays_mv = stub_render_rays_mv
  n_rays = H * W
  g = torch.Generator().manual_seed(5)
  batch = {'ray_o': torch.rand(n_rays, 3, generator=g), 'camera': torch.zeros(1, 34), 'flows': torch.rand(6, n_rays, 2, generator=g), 'rgb': None}
  del CALLS[:]
  args = types.SimpleNamespace(frame_outputs=mode)
  ret = RI.render_single_image_nvi((0, None), (None, None), ([0], None), types.SimpleNamespace(H=H, W=W), batch, None, None, chunk_size, 4, args,
                                   N_importance=4, det=True, is_train=False)
  assert ret['outputs_fine'] is None and len(ret['outputs_fine_anchor']) == 0
  pend = {g_: list(ret[g_].pending()) for g_ in ('outputs_coarse_ref', 'outputs_fine_ref')}
  out = {}
  for g_ in ('outputs_coarse_ref', 'outputs_fine_ref'):
    keys = list(ret[g_].keys())
    # lazy mode: read a subset (SPMD: the same entries in the same order on every rank); 'all': everything is already there
    want = keys if mode == 'all' else [k for k in keys if k in read or k in ('rgb', 'depth', 'mask')]
    out[g_] = {'__keys__': keys}
    for k in want:
      v = ret[g_][k]
      out[g_][k] = [t.numpy().copy() for t in v] if isinstance(v, list) else v.numpy().copy()
  return out, list(CALLS), pend


def worker(rank, world, port, H, W, chunk_size, mode, q):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    out, seen, pend = run_frame(H, W, chunk_size, mode)
    q.put((rank, out, seen, pend))
  finally:
    dist.barrier()
    dist.destroy_process_group()


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def run_ranks(world, H, W, chunk, mode):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = free_port()
  procs = [ctx.Process(target=worker, args=(r, world, port, H, W, chunk, mode, q)) for r in range(world)]
  for p in procs:
    p.start()
  got = [q.get(timeout=180) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  return sorted(got, key=lambda g: g[0])


@pytest.mark.parametrize('H,W,chunk,mode', [(7, 9, 10, 'lazy'), (7, 13, 64, 'all'), (7, 9, 10, 'all'), (1, 1, 8, 'lazy'), (1, 3, 2, 'all')])
def test_two_rank_frame_equals_single_process(H, W, chunk, mode):
  frame_equals_single_process(2, H, W, chunk, mode)


@pytest.mark.parametrize('H,W,chunk,mode', [(1, 8, 2, 'lazy'), (1, 2, 4, 'lazy'), (5, 5, 4, 'all')])
def test_three_rank_frame_with_unequal_and_empty_tiles(H, W, chunk, mode):
  """world_size 3: 8 rays -> tiles of 2 / 3 / 3 rays (padded to 3 for the equal-count all-gather), 2 rays -> rank 0's tile is EMPTY (it renders a
  placeholder ray and contributes none, yet joins every collective), 25 rays -> 8 / 8 / 9 with several chunks per tile."""
  from dynibar_amd import render_image as RI
  sizes = [RI.ray_tile(H * W, 3, r)[1] - RI.ray_tile(H * W, 3, r)[0] for r in range(3)]
  assert len(set(sizes)) > 1, 'the case is meant to have unequal tiles'
  frame_equals_single_process(3, H, W, chunk, mode)


def frame_equals_single_process(world, H, W, chunk, mode):
  sys.path.insert(0, ROOT)
  n_rays = H * W
  ref, seen1, pend1 = run_frame(H, W, chunk, mode)
  assert sum(seen1) == n_rays
  got = run_ranks(world, H, W, chunk, mode)
  rendered = sum(sum(s) for _, _, s, _ in got)
  from dynibar_amd import render_image as RI_
  n_empty = sum(1 for r in range(world) if RI_.ray_tile(n_rays, world, r)[1] == RI_.ray_tile(n_rays, world, r)[0])
  assert rendered == n_rays + n_empty, 'every ray is rendered exactly once across the ranks (+ one placeholder per empty tile)'
  for rank, out, _, pend in got:
    for grp in ref:
      assert out[grp]['__keys__'] == ref[grp]['__keys__'], 'key set / order of the tiled frame differs'
      for k in ref[grp]:
        if k == '__keys__':
          continue
        if isinstance(ref[grp][k], list):
          # 4-D entries stay per-chunk lists (as the reference leaves them); under tiling a rank holds the chunks of its own tile
          assert isinstance(out[grp][k], list) and all(t.ndim == 4 for t in out[grp][k])
          continue
        a, b = out[grp][k], ref[grp][k]
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), f'rank {rank} {grp}/{k} differs from the single-process frame'
    if mode == 'lazy':
      # only the primary group's rgb / depth / mask were assembled by the call itself: one packed [rays, 5] all-gather
      assert sorted(set(out['outputs_fine_ref']['__keys__']) - set(pend['outputs_fine_ref'])) == ['depth', 'mask', 'rgb']
      assert sorted(pend['outputs_coarse_ref']) == sorted(out['outputs_coarse_ref']['__keys__'])
    else:
      assert pend['outputs_fine_ref'] == [] and pend['outputs_coarse_ref'] == []
  assert isinstance(ref['outputs_coarse_ref']['dropme'], list), '4-D entries are kept as per-chunk lists, never assembled'
  both = [t for _, out, _, _ in got for t in out['outputs_coarse_ref']['dropme']]
  assert np.array_equal(np.concatenate(both, axis=1), np.concatenate(ref['outputs_coarse_ref']['dropme'], axis=1))
  # masked pixels are blanked after the gather, like render_image.py:186-188 -- also for a group whose rgb was resolved late
  for grp in ref:
    m = ref[grp]['mask']
    assert bool((ref[grp]['rgb'].reshape(-1, 3)[m.reshape(-1) == 0] == 0).all())


def test_ray_tiles_partition():
  from dynibar_amd.render_image import ray_tile
  for n in (1, 5, 7, 64, 147456, 147457):
    for world in (1, 2, 3, 4, 8):
      spans = [ray_tile(n, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
      sizes = [hi - lo for lo, hi, _ in spans]
      assert max(sizes) - min(sizes) <= 1 and all(sz <= tile for sz, (_, _, tile) in zip(sizes, spans)), 'balanced tiles'


def test_collective_payload_of_a_frame():
  """What the call itself moves between ranks by default: rgb + depth + mask of the primary group, 20 B per ray (2.9 MB at 288x512)."""
  from dynibar_amd import render_image as RI
  sent = []

  class FakeDist:
    @staticmethod
    def all_gather_into_tensor(recv, send):
      sent.append(send.numel() * 4)
      recv.view(2, -1)[:] = send.reshape(1, -1)

  n = 288 * 512
  local = {'rgb': torch.zeros(n // 2, 3), 'depth': torch.zeros(n // 2), 'mask': torch.zeros(n // 2, dtype=torch.bool)}
  out = RI.gather_rows(local, n, FakeDist, 2, 0)
  assert len(sent) == 1 and sent[0] * 2 == n * 20 and n * 20 <= 3 * 1024 * 1024
  assert out['mask'].dtype == torch.bool and out['rgb'].shape == (n, 3) and out['depth'].shape == (n,)


def test_frame_outputs_mapping_semantics():
  """FrameOutputs looks like the reference's OrderedDict of host tensors: same keys and order whether or not an entry has been
  resolved; deferred entries are produced once, on first read; dict-style access paths all resolve."""
  import pickle
  from dynibar_amd.render_image import FrameOutputs
  calls = []
  f = FrameOutputs()
  f['rgb'] = torch.ones(2, 3)
  f._defer('weights', lambda: calls.append('weights') or torch.zeros(2, 5))
  f._defer('alpha', lambda: calls.append('alpha') or torch.full((2, 5), 2.0))
  assert list(f.keys()) == ['rgb', 'weights', 'alpha'] and len(f) == 3 and 'alpha' in f and f.pending() == ['weights', 'alpha']
  assert calls == []
  assert float(f['weights'].sum()) == 0.0 and calls == ['weights']
  assert f['weights'] is f['weights'] and calls == ['weights'], 'resolved once, cached'
  assert float(f.get('alpha')[0, 0]) == 2.0 and f.get('nope', 7) == 7 and calls == ['weights', 'alpha']
  g = FrameOutputs()
  g._defer('a', lambda: torch.zeros(1))
  g._defer('b', lambda: torch.ones(1))
  assert [k for k, v in g.items()] == ['a', 'b'] and all(isinstance(v, torch.Tensor) for v in g.values()) and g.pending() == []
  h = FrameOutputs()
  h._defer('a', lambda: torch.arange(3.0))
  back = pickle.loads(pickle.dumps(h))
  assert list(back.keys()) == ['a'] and torch.equal(back['a'], torch.arange(3.0))
  k = FrameOutputs()
  k._defer('a', lambda: torch.ones(1))
  assert float(k.pop('a')) == 1.0 and 'a' not in k


# ---- data-parallel training: the gradient all-reduce (section 8e / 8(f)3) on two gloo ranks ------------------------------------------------
def _grad_worker(rank, world, port, q):
  import torch.distributed as dist
  import torch
  from dynibar_amd import train_dist
  dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
  try:
    torch.manual_seed(0)
    model = types.SimpleNamespace(net_coarse_st=torch.nn.Linear(7, 5), net_coarse_dy=torch.nn.DataParallel(torch.nn.Linear(3, 4)),
                                  motion_mlp=torch.nn.Linear(2, 2), trajectory_basis=torch.nn.Parameter(torch.zeros(6, 3)))
    params = train_dist.trainable_parameters(model)
    g = torch.Generator().manual_seed(100 + rank)
    for i, p in enumerate(params):
      if not (rank == 1 and i == 2):   # rank 1 has no gradient for one parameter: it must still join the collective with zeros
        p.grad = torch.randn(p.shape, generator=g)
    n = train_dist.allreduce_gradients(params, bucket_bytes=64)   # tiny buckets: several collectives
    q.put((rank, n, [p.grad.numpy().copy() for p in params]))
  finally:
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks():
  import torch
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29731
  procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  # expected: the mean over ranks of the per-rank seeded gradients (zeros where a rank had none)
  shapes = [(5, 7), (5,), (4, 3), (4,), (2, 2), (2,), (6, 3)]
  per_rank = []
  for rank in range(2):
    g = torch.Generator().manual_seed(100 + rank)
    gs = []
    for i, shp in enumerate(shapes):
      gs.append(torch.zeros(shp) if (rank == 1 and i == 2) else torch.randn(shp, generator=g))
    per_rank.append(gs)
  for (rank, n, grads) in res:
    assert n > 1
    for i, got in enumerate(grads):
      np.testing.assert_allclose(got, ((per_rank[0][i] + per_rank[1][i]) / 2).numpy(), rtol=1e-6, atol=1e-7)


def test_balanced_chunks_cover_the_tile_without_a_short_tail():
  """render_image.chunk_bounds: the reference's range(lo, hi, chunk_size) leaves a short tail chunk (18 432 rays -> 8192 + 8192 + 2048); the balanced
  form keeps the number of chunks and evens them out (3 x 6144), covers [lo, hi) exactly once and never exceeds chunk_size"""
  from dynibar_amd.render_image import chunk_bounds
  assert chunk_bounds(0, 18432, 8192, False) == [(0, 8192), (8192, 16384), (16384, 18432)]
  assert chunk_bounds(0, 18432, 8192, True) == [(0, 6144), (6144, 12288), (12288, 18432)]
  assert chunk_bounds(5, 5, 8192, True) == []
  assert chunk_bounds(0, 100, 8192, True) == [(0, 100)]
  for lo, hi, cs in ((0, 147456, 8192), (36864, 55296, 8192), (7, 20011, 4096), (0, 8193, 8192), (3, 130, 64)):
    for bal in (False, True):
      b = chunk_bounds(lo, hi, cs, bal)
      assert b[0][0] == lo and b[-1][1] == hi and all(x[1] == y[0] for x, y in zip(b, b[1:])) and all(0 < y - x <= cs for x, y in b)
      assert len(b) == -(-(hi - lo) // cs)
    sizes = [y - x for x, y in chunk_bounds(lo, hi, cs, True)]
    assert max(sizes) - min(sizes) <= 64 + (max(sizes) - sizes[-1])  # equal up to the 64-ray rounding; only the last chunk may be shorter


def test_chunk_walks_that_differ_from_the_reference_keep_its_three_ray_tail_and_make_no_other():
  """A chunk of exactly 3 rays is the one place where the reference's result depends on the chunking (torch.cross without dim crosses the Pluecker
  moments over the rays).  A rank's tile / balanced chunks keep the reference's own 3-ray tail as a chunk and never form another chunk of 3 rays."""
  from dynibar_amd.render_image import chunk_bounds, ray_tile
  # the reference's own walk is left alone, 3-ray tail included
  assert chunk_bounds(0, 8195, 8192, False, n_rays=8195) == [(0, 8192), (8192, 8195)]
  # two ranks: the second rank's tile ends in the reference's tail -- kept as its own chunk, also with balanced chunks
  lo, hi, _ = ray_tile(24579, 2, 1)
  for bal in (False, True):
    b = chunk_bounds(lo, hi, 8192, bal, n_rays=24579)
    assert (24579) % 8192 == 3 and b[-1] == (24579 - 3, 24579) and b[0][0] == lo and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    assert [y - x for x, y in b].count(3) == 1
  # a tile whose own walk would end in 3 rays that are NOT the reference's tail: they join the previous chunk
  assert chunk_bounds(1000, 1131, 64, False, n_rays=5000) == [(1000, 1064), (1064, 1131)]
  assert chunk_bounds(1000, 1131, 64, True, n_rays=5000) == [(1000, 1064), (1064, 1131)]
  # a tile of exactly 3 rays that are not the reference's tail: 2 + 1
  assert chunk_bounds(10, 13, 64, True, n_rays=5000) == [(10, 12), (12, 13)]
  # the whole frame is 3 rays: the reference's only chunk
  assert chunk_bounds(0, 3, 64, True, n_rays=3) == [(0, 3)]
  for n_rays, world, cs in ((8195, 3, 512), (1027, 4, 256), (147456, 8, 8192), (67, 5, 16), (35, 8, 8)):
    for rank in range(world):
      lo, hi, _ = ray_tile(n_rays, world, rank)
      for bal in (False, True):
        b = chunk_bounds(lo, hi, cs, bal, n_rays=n_rays)
        assert (not b and hi == lo) or (b[0][0] == lo and b[-1][1] == hi and all(x[1] == y[0] for x, y in zip(b, b[1:])))
        threes = [x for x in b if x[1] - x[0] == 3]
        assert all(x == (n_rays - 3, n_rays) and n_rays % cs == 3 for x in threes), (n_rays, world, cs, rank, bal, b)
