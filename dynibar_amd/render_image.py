"""Drop-in for the reference's ``ibrnet/render_image.py``: ``render_single_image_nvi`` / ``render_single_image_mono`` with the
reference signatures and return structure (reference render_image.py:9-217, :220-439).

Multi-GPU.  Rays are independent, so when ``torch.distributed`` is initialised (one process per GPU, RCCL) the H*W rays of the target
view are split into ``world_size`` contiguous tiles of equal size (+-1 ray); every rank renders its own tile chunk by chunk and every
rank returns the full frame (the callers are SPMD scripts).  The reference's counterpart is ``nn.DataParallel`` scatter / gather inside
every module call (model.py:134-159).

Output contract.  The reference copies EVERY tensor of EVERY chunk to the host (render_image.py:123-135): per 288x512 frame about
1 GB of per-sample arrays (weights, alpha, z_vals ... [H,W,S]) that ``eval_nvidia.py:380-381`` never reads.  Here a frame group is a
``FrameOutputs`` mapping with the reference's keys, order, shapes and dtypes, in which
  * the pixels every caller reads -- ``rgb``, ``depth``, ``mask`` of the frame's primary group -- are assembled eagerly: ONE packed
    ``[rays, 5]`` all-gather per frame (<= 3 MB at 288x512) and one device-to-host copy;
  * every other entry is resolved on first access (``ret['outputs_coarse_ref']['weights']`` gathers and copies that tensor then, and
    caches it).  A caller that reads an entry gets exactly what the reference returns; a caller that does not pays nothing.
    With world_size > 1 the access issues a collective, so ranks must read the same entries in the same order (SPMD callers do).
``FRAME_OUTPUTS = 'all'`` (or ``args.frame_outputs`` / env ``DYNIBAR_FRAME_OUTPUTS``) restores the eager behaviour: everything is
assembled before the call returns, with one packed collective per group.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

from .render_ray import render_rays_mono, render_rays_mv

_PER_VIEW_KEYS = ('camera', 'anchor_camera', 'render_camera', 'depth_range', 'src_rgbs', 'src_cameras', 'anchor_src_rgbs', 'anchor_src_cameras', 'static_src_rgbs',
                  'static_src_cameras')

FRAME_OUTPUTS = 'lazy'       # 'lazy' | 'all'
TILE_ACROSS_RANKS = True     # False: every rank renders the whole frame by itself (no collective), e.g. when only one rank calls in
_EAGER_KEYS = ('rgb', 'depth', 'mask')
GATHER = os.environ.get('DYNIBAR_GATHER', 'torch')  # the frame's pixel all-gather: 'torch' = torch.distributed.all_gather_into_tensor (RCCL through
                             # PyTorch's process group), 'abi' = the C-ABI's dyn_gather_tiles on a communicator of this package's own (what a
                             # host without PyTorch calls; include/dynibar_hip.h)
FORCE_DIST = os.environ.get('DYNIBAR_FORCE_DIST', '0') == '1'  # take the multi-rank code path (tiles, collective) with a process group of ONE rank too
FRAME_STATS = None           # bench.py sets this to a dict for ONE frame: per-stage clocks (tile_rays, render_ms, gather_ms, gather_bytes); the device is
                             # synchronised between the stages while it is set, never otherwise


def _clock():
  import time
  if torch.cuda.is_available():
    torch.cuda.synchronize()
  return time.perf_counter()


def _dist():
  import torch.distributed as dist
  if TILE_ACROSS_RANKS and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_DIST):
    return dist, dist.get_world_size(), dist.get_rank()
  return None, 1, 0


_ABI_COMMS = {}


def _destroy_abi_comms(keep_group=None):
  """ncclCommDestroy for the communicators this module made (atexit: all of them; when the default process group has changed: those of OTHER groups,
  the entries of `keep_group` -- other devices of the same group included -- stay): RCCL warns or hangs at teardown with live communicators."""
  from ._lib import call
  for key in list(_ABI_COMMS):
    group_ref, comm = _ABI_COMMS[key]
    if keep_group is not None and group_ref() is keep_group:
      continue
    del _ABI_COMMS[key]
    try:
      call('dyn_comm_destroy', comm)
    except Exception:  # teardown: the device or RCCL may already be gone
      pass


def abi_communicator(dist, world, rank, device):
  """ncclComm_t (a ctypes.c_void_p) of this package's own for dyn_gather_tiles on `device`: rank 0 draws the 128-byte id (dyn_comm_unique_id),
  torch.distributed carries it to the other ranks, every rank joins (dyn_comm_init_rank).  One per (world, rank, device) of the CURRENT default process
  group.  An entry holds a weak reference to the group OBJECT and is only reused while that very object is still the default group (`is`, not `id()`:
  after destroy_process_group() + re-init CPython may hand the new group the old address); entries of a group that is gone are destroyed before a new
  communicator is made -- every rank takes that path at the same call, the re-init being collective -- and entries of other devices of the live group
  are kept.  All of them are destroyed at interpreter exit."""
  import ctypes
  import weakref
  from ._lib import call
  group = getattr(getattr(dist, 'group', None), 'WORLD', None)
  key = (world, rank, device.index)
  hit = _ABI_COMMS.get(key)
  if hit is not None and hit[0]() is group:
    return hit[1]
  _destroy_abi_comms(keep_group=group)  # (drops a stale entry under `key` too: its group is not `group`)
  if not getattr(abi_communicator, '_hooked', False):
    import atexit
    atexit.register(_destroy_abi_comms)
    abi_communicator._hooked = True
  idbuf = (ctypes.c_char * 128)()
  if rank == 0:
    call('dyn_comm_unique_id', idbuf)
  if world > 1:
    on = device if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(idbuf.raw), dtype=torch.uint8, device=on)
    dist.broadcast(t, 0)
    idbuf = (ctypes.c_char * 128)(*bytes(t.cpu().tolist()))
  comm = ctypes.c_void_p()
  with torch.cuda.device(device):
    call('dyn_comm_init_rank', ctypes.byref(comm), world, idbuf, rank)
  try:
    ref = weakref.ref(group)
  except TypeError:  # (a group object that cannot be weakly referenced: hold it -- identity still decides)
    ref = (lambda g: (lambda: g))(group)
  _ABI_COMMS[key] = (ref, comm)
  return comm


def _record_stream(obj, stream):
  """Tell the caching allocator that every device tensor inside `obj` (nested dicts / lists / tuples included) is also used on `stream`."""
  if isinstance(obj, torch.Tensor):
    if obj.is_cuda:
      obj.record_stream(stream)
  elif isinstance(obj, dict):
    for v in obj.values():
      _record_stream(v, stream)
  elif isinstance(obj, (list, tuple)):
    for v in obj:
      _record_stream(v, stream)


def ray_tile(n_rays, world, rank):
  """Contiguous tile [lo, hi) of rank ``rank`` and the padded tile size: a balanced split, tile sizes differ by at most one ray
  (ranks beyond n_rays get an empty tile and still take part in the collectives)."""
  lo = (n_rays * rank) // world
  hi = (n_rays * (rank + 1)) // world
  return lo, hi, (n_rays + world - 1) // world


def slice_ray_batch(ray_batch, lo, hi, per_view_keys=_PER_VIEW_KEYS):
  """The reference's chunk slicing rules (render_image.py:70-89): per-view tensors whole, [V,N,...] tensors on dim 1, others on dim 0."""
  chunk = OrderedDict()
  for k in ray_batch:
    v = ray_batch[k]
    if v is None:
      chunk[k] = None
    elif k in per_view_keys:
      chunk[k] = v
    elif len(v.shape) == 3:  # flows and masks
      chunk[k] = v[:, lo:hi, ...]
    else:
      chunk[k] = v[lo:hi]
  return chunk


# ----------------------------------------------------------------------------------------------------------------------
# tile -> frame
# ----------------------------------------------------------------------------------------------------------------------
def _rows(t):
  """[N, ...] or [V, N, c] tensor -> ([N, cols] float32 rows, restore(rows) -> original layout)."""
  dtype = t.dtype
  if t.dim() == 3:
    V, N, c = t.shape
    rows = t.permute(1, 0, 2).reshape(N, V * c)
    back = lambda r: r.reshape(r.shape[0], V, c).permute(1, 0, 2).contiguous()
  else:
    shape = tuple(t.shape[1:])
    rows = t.reshape(t.shape[0], -1)
    back = lambda r: r.reshape((r.shape[0],) + shape)
  if dtype != torch.float32:
    rows = rows.float()  # the ray masks (bool) travel as 0 / 1
    inner = back
    back = lambda r: inner(r).to(dtype)
  return rows, back


def gather_rows(local, n_rays, dist, world, rank, count=None):
  """local: {key: tensor over this rank's tile (ray axis 0; axis 1 for 3-D [V,N,c] tensors)} -> {key: tensor over all n_rays} on every
  rank: the entries are packed side by side into ONE [tile, C] buffer and reassembled with ONE all_gather_into_tensor.
  ``count``: valid rays of the local tile (default: all rows; 0 for a rank whose tile is empty and that rendered a placeholder ray)."""
  if not local:
    return OrderedDict()
  packed, restore, widths = [], [], []
  for k, t in local.items():
    rows, back = _rows(t)
    packed.append(rows)
    restore.append(back)
    widths.append(rows.shape[1])
  n_local = packed[0].shape[0] if count is None else count
  buf = packed[0] if len(packed) == 1 else torch.cat(packed, dim=1)
  if dist is not None:
    lo, hi, tile = ray_tile(n_rays, world, rank)
    assert hi - lo == n_local, 'a rank renders exactly its own tile'
    send = torch.zeros((tile, buf.shape[1]), dtype=torch.float32, device=buf.device)
    send[:n_local] = buf[:n_local]
    recv = torch.empty((world * tile, buf.shape[1]), dtype=torch.float32, device=buf.device)
    if GATHER == 'abi' and buf.is_cuda:
      from ._lib import call, ptr, stream_of
      call('dyn_gather_tiles', ptr(send), ptr(recv), tile, buf.shape[1], abi_communicator(dist, world, rank, buf.device), stream_of(send))
    else:
      dist.all_gather_into_tensor(recv, send)
    spans = [ray_tile(n_rays, world, r) for r in range(world)]
    if all(h - l == tile for l, h, _ in spans):
      buf = recv
    else:
      buf = torch.cat([recv[r * tile: r * tile + (h - l)] for r, (l, h, _) in enumerate(spans)], dim=0)
  else:
    buf = buf[:n_local]
  out = OrderedDict()
  c0 = 0
  for (k, _), back, w in zip(local.items(), restore, widths):
    out[k] = back(buf[:, c0:c0 + w])
    c0 += w
  return out


def _to_host(tensors):
  """{key: device tensor} -> {key: host tensor}: one batch of asynchronous copies into pinned memory and a single synchronisation
  (the reference: one blocking .cpu() per tensor per chunk, render_image.py:113-118)."""
  host = OrderedDict()
  try:
    for k, t in tensors.items():
      host[k] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True) if t.is_cuda else t
    if any(t.is_cuda for t in tensors.values()):
      torch.cuda.current_stream().synchronize()
  except RuntimeError:  # no pinned memory to be had: plain synchronous copies
    host = OrderedDict((k, t.cpu()) for k, t in tensors.items())
  return host


class FrameOutputs(OrderedDict):
  """One output group of a frame: the reference's keys in the reference's order.  Entries not yet assembled are resolved (gathered
  across ranks, copied to the host, reshaped to [H, W, ...]) by the first read and cached."""

  def __init__(self):
    super().__init__()
    self._pending = {}

  def _defer(self, key, thunk):
    super().__setitem__(key, None)
    self._pending[key] = thunk

  def _resolve(self, key):
    thunk = self._pending.pop(key, None)
    if thunk is not None:
      super().__setitem__(key, thunk())

  def pending(self):
    return list(self._pending)

  def resolve_all(self):
    for k in list(self._pending):
      self._resolve(k)
    return self

  def __getitem__(self, key):
    self._resolve(key)
    return super().__getitem__(key)

  def get(self, key, default=None):
    return self[key] if key in self else default

  def __setitem__(self, key, value):
    self._pending.pop(key, None)
    super().__setitem__(key, value)

  def items(self):
    self.resolve_all()
    return super().items()

  def values(self):
    self.resolve_all()
    return super().values()

  def pop(self, key, *default):
    if key in self:
      self._resolve(key)
    return super().pop(key, *default)

  def __reduce__(self):  # pickling / copying a frame materialises it
    self.resolve_all()
    return (OrderedDict, (list(super().items()),))


def _frame_mode(args):
  mode = getattr(args, 'frame_outputs', None) or os.environ.get('DYNIBAR_FRAME_OUTPUTS') or FRAME_OUTPUTS
  if mode not in ('lazy', 'all'):
    raise ValueError(f"frame_outputs must be 'lazy' or 'all', got {mode!r}")
  return mode


def _shape_frame(t, Hs, Ws):
  """render_image.py:137-188: 3-D [V,N,c] entries -> [V,H,W,c], others -> [H,W,...], singleton axes squeezed."""
  if t.dim() == 3:
    return t.reshape((t.shape[0], Hs, Ws, -1)).squeeze()
  return t.reshape((Hs, Ws, -1)).squeeze()


def _assemble(per_chunk, n_rays, Hs, Ws, dist, world, rank, count=None, eager=None):
  """list of per-chunk output dicts of this rank's tile -> FrameOutputs of the full frame.  eager: keys assembled now (None: all)."""
  frame = FrameOutputs()
  keys = list(per_chunk[0].keys()) if per_chunk else []
  local = OrderedDict()
  for k in keys:
    parts = [c[k] for c in per_chunk]
    if parts[0] is None or k == 'random_sigma':
      continue
    if parts[0].dim() == 4:
      # the reference leaves 4-D entries as its per-chunk list of host tensors (render_image.py:368-369); under tiling: this rank's chunks
      n_keep = len(parts) if count is None or count > 0 else 0
      frame._defer(k, (lambda ps=parts[:n_keep]: [t.cpu() for t in ps]))
      continue
    local[k] = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1 if parts[0].dim() == 3 else 0)
    frame._defer(k, None)
  now = [k for k in local if eager is None or k in eager]

  def fetch(ks):
    t0 = _clock() if FRAME_STATS is not None else 0.0
    full = gather_rows(OrderedDict((k, local[k]) for k in ks), n_rays, dist, world, rank, count)
    host = OrderedDict((k, _shape_frame(t, Hs, Ws)) for k, t in _to_host(full).items())
    if FRAME_STATS is not None and ks:
      tile = ray_tile(n_rays, world, rank)[2]
      FRAME_STATS['gather_ms'] = FRAME_STATS.get('gather_ms', 0.0) + (_clock() - t0) * 1e3
      FRAME_STATS['gather_bytes'] = FRAME_STATS.get('gather_bytes', 0) + 4 * tile * sum(_rows(local[k])[0].shape[1] for k in ks)
    return host

  def blank(rgb, mask):
    rgb[mask == 0] = 0.0  # render_image.py:162-164, :186-188: pixels whose ray mask is off are zeroed, in every group
    return rgb

  for k, t in fetch(now).items():
    frame[k] = t
  if 'rgb' in now and 'mask' in now:
    blank(OrderedDict.__getitem__(frame, 'rgb'), OrderedDict.__getitem__(frame, 'mask'))

  def late(k):
    if k == 'rgb' and 'mask' in local:  # the blanking needs the ray mask: both travel together
      got = fetch(['rgb', 'mask'] if 'mask' in frame._pending else ['rgb'])
      if 'mask' in got:
        frame['mask'] = got['mask']
      return blank(got['rgb'], frame['mask'])
    return fetch([k])[k]

  for k in local:
    if k not in now:
      frame._defer(k, (lambda kk=k: late(kk)))
  if eager is None:
    frame.resolve_all()
  return frame


CHUNK_STREAMS = int(os.environ.get('DYNIBAR_CHUNK_STREAMS', '2'))  # HIP streams the chunks of a frame alternate over (1: the caller's stream only)
_SIDE_STREAMS = {}


def _side_streams(dev, n):
  key = (dev.index, n)
  if key not in _SIDE_STREAMS:
    _SIDE_STREAMS[key] = [torch.cuda.Stream(dev) for _ in range(n)]
  return _SIDE_STREAMS[key]


BALANCED_CHUNKS = os.environ.get('DYNIBAR_BALANCED_CHUNKS', 'tiled')  # 'tiled': when the frame is tiled across ranks; 'always'; 'never'


def chunk_bounds(lo, hi, chunk_size, balanced, n_rays=None):
  """[(a, b)] covering [lo, hi).  The reference walks range(lo, hi, chunk_size) (render_image.py:68): a rank's tile of 18 432 rays at chunk_size 8192
  would render as 8192 + 8192 + 2048, and the 2048-ray tail pays the full start-up of the one-workgroup-per-CU kernels for a quarter of the rays.
  Balanced: the same NUMBER of chunks, equal sizes rounded up to 64 rays (18 432 -> 3 x 6144); only the per-chunk lists of 4-D entries change their
  split points, and under tiling those are rank-local already.
  Results do not depend on the chunking -- every ray is independent (tests/test_distributed_cpu.py, the chunk-invariance GPU test) -- with ONE exception
  that is the reference's own: a chunk of exactly 3 rays has its Pluecker moments crossed over the rays (torch.cross without dim, render_ray.py:375 /
  :392; csrc/dyn_device.h).  With ``n_rays`` (the frame's ray count) a walk that differs from the reference's -- a rank's tile, balanced chunks --
  therefore keeps the reference's 3-ray tail (n_rays = 3 mod chunk_size) as a chunk of its own when the tile holds it, and never forms another chunk
  of exactly 3 rays (it joins its neighbour: 3 rays above chunk_size, or is rendered as 2 + 1)."""
  n = hi - lo
  if n <= 0:
    return []
  if not balanced or n <= chunk_size:
    bounds = [(i, min(i + chunk_size, hi)) for i in range(lo, hi, chunk_size)]
  else:
    k = (n + chunk_size - 1) // chunk_size
    size = min(chunk_size, ((n + k - 1) // k + 63) // 64 * 64)
    bounds = [(i, min(i + size, hi)) for i in range(lo, hi, size)]
  if n_rays is None or chunk_size == 3 or (not balanced and lo == 0 and hi == n_rays):
    return bounds  # the reference's own walk (or a caller that does not say which frame this is)
  tail = (n_rays - 3, n_rays) if n_rays % chunk_size == 3 else None
  if tail is not None and hi == n_rays and lo <= tail[0]:
    bounds = (chunk_bounds(lo, tail[0], chunk_size, balanced) if lo < tail[0] else []) + [tail]
  else:
    tail = None  # (not in this tile, or cut by the tile's edge: its rays are then rendered like any others)
  out = []
  for a, b in bounds:
    if b - a == 3 and (a, b) != tail:
      if out and out[-1] != tail:
        out[-1] = (out[-1][0], b)
      else:
        out += [(a, a + 2), (a + 2, b)]
    else:
      out.append((a, b))
  return out


def _render_tiles(ray_batch, chunk_size, render_chunk, group_names):
  dist, world, rank = _dist()
  n_rays = ray_batch['ray_o'].shape[0]
  lo, hi, _ = ray_tile(n_rays, world, rank)
  count = hi - lo
  if count == 0:
    lo, hi = 0, 1  # an empty tile still joins the collectives: render one placeholder ray for the key / shape structure, contribute none
  chunks = {g: [] for g in group_names}
  balanced = BALANCED_CHUNKS == 'always' or (BALANCED_CHUNKS == 'tiled' and world > 1)
  bounds = chunk_bounds(lo, hi, chunk_size, balanced, n_rays=n_rays)
  t0 = _clock() if FRAME_STATS is not None else 0.0
  # per-chunk device times from stream events (no synchronisation inside the loop: a host clock per chunk exposes the launch latency of every
  # chunk's first kernels -- 18 x 2.4 ms of a 730 ms frame when it was tried)
  marks = []
  timed = FRAME_STATS is not None and torch.cuda.is_available() and ray_batch['ray_o'].is_cuda

  def mark():
    if timed:
      e = torch.cuda.Event(enable_timing=True)
      e.record()
      marks.append(e)

  mark()
  spans = []
  dev = ray_batch['ray_o'].device
  n_streams = CHUNK_STREAMS if (dev.type == 'cuda' and len(bounds) > 2) else 1
  if n_streams > 1:
    # Chunks are independent, so consecutive chunks go to alternating side streams: the device then fills the tail of one chunk's kernels (and the idle
    # CUs under its one-workgroup-per-CU kernels, which leave 27 KB of LDS and 56 registers per lane free) with the next chunk's small and HBM-bound
    # kernels -- sampling, the projection / gather, compositing.  The first chunk fills the per-view caches (projection matrices, channels-last maps,
    # packed networks): the second stream starts behind it; after that the two streams run free.  Outputs are handed to the caller's stream at the end.
    cur = torch.cuda.current_stream(dev)
    side = _side_streams(dev, n_streams)
    for st in side:
      st.wait_stream(cur)
    for i, (a, b) in enumerate(bounds):
      st = side[i % n_streams]
      if i == 1:
        for other in side[1:]:
          other.wait_stream(side[0])
      with torch.cuda.stream(st):
        if timed:  # the chunk's own span on its stream (spans of chunks on different streams overlap: their sum exceeds the frame)
          spans.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
          spans[-1][0].record()
        ret = render_chunk(slice_ray_batch(ray_batch, a, b))
        if timed:
          spans[-1][1].record()
        for g in group_names:
          if ret.get(g) is not None:
            _record_stream(ret[g], cur)
            chunks[g].append(ret[g])
    for st in side:
      cur.wait_stream(st)
    mark()
  else:
    for a, b in bounds:
      ret = render_chunk(slice_ray_batch(ray_batch, a, b))
      for g in group_names:
        if ret.get(g) is not None:
          chunks[g].append(ret[g])
      mark()
  if FRAME_STATS is not None:
    FRAME_STATS.update(tile_rays=count, render_ms=(_clock() - t0) * 1e3, gather_ms=0.0, gather_bytes=0, chunk_rays=[b - a for a, b in bounds],
                       chunk_ms=([round(e0.elapsed_time(e1), 3) for e0, e1 in spans] if spans else
                                 [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(len(marks) - 1)]), chunk_streams=n_streams)
  return chunks, n_rays, dist, world, rank, count


def _frame(chunks, groups, primary, n_rays, Hs, Ws, dist, world, rank, count, mode):
  all_ret = OrderedDict()
  for g in groups:
    if not chunks[g]:
      all_ret[g] = OrderedDict()
      continue
    eager = None if mode == 'all' else (_EAGER_KEYS if g == primary else ())
    all_ret[g] = _assemble(chunks[g], n_rays, Hs, Ws, dist, world, rank, count, eager)
  all_ret['outputs_fine'] = None
  return all_ret


def render_single_image_nvi(frame_idx, time_embedding, time_offset, ray_sampler, ray_batch, model, projector, chunk_size, N_samples, args,
                            inv_uniform=False, N_importance=0, det=False, white_bkgd=False, render_stride=1, coarse_featmaps=None,
                            fine_featmaps=None, is_train=True):
  """Reference render_image.py:9-217: full-frame coarse+fine rendering of one target view (Nvidia dynamic scenes)."""
  def render_chunk(chunk):
    return render_rays_mv(frame_idx=frame_idx, time_embedding=time_embedding, time_offset=time_offset, ray_batch=chunk, model=model,
                          coarse_featmaps=coarse_featmaps, fine_featmaps=fine_featmaps, projector=projector, N_samples=N_samples, args=args,
                          inv_uniform=inv_uniform, N_importance=N_importance, raw_noise_std=0.0, det=det, white_bkgd=white_bkgd,
                          is_train=is_train)

  groups = ('outputs_fine_anchor', 'outputs_fine_ref', 'outputs_coarse_ref')
  chunks, n_rays, dist, world, rank, count = _render_tiles(ray_batch, chunk_size, render_chunk, groups)
  Hs = len(range(0, ray_sampler.H, render_stride))
  Ws = len(range(0, ray_sampler.W, render_stride))
  return _frame(chunks, groups, 'outputs_fine_ref', n_rays, Hs, Ws, dist, world, rank, count, _frame_mode(args))


def render_single_image_mono(frame_idx, time_embedding, time_offset, ray_sampler, ray_batch, model, projector, chunk_size, N_samples, args,
                             inv_uniform=False, N_importance=0, det=False, white_bkgd=False, render_stride=1, featmaps=None, is_train=True,
                             num_vv=2):
  """Reference render_image.py:220-439: full-frame coarse rendering of one target view (monocular video)."""
  def render_chunk(chunk):
    return render_rays_mono(frame_idx=frame_idx, time_embedding=time_embedding, time_offset=time_offset, ray_batch=chunk, model=model,
                            featmaps=featmaps, projector=projector, N_samples=N_samples, args=args, inv_uniform=inv_uniform,
                            N_importance=N_importance, raw_noise_std=0.0, det=det, white_bkgd=white_bkgd, is_train=is_train, num_vv=num_vv)

  groups = ('outputs_coarse_ref', 'outputs_coarse_st', 'outputs_coarse_anchor')
  chunks, n_rays, dist, world, rank, count = _render_tiles(ray_batch, chunk_size, render_chunk, groups)
  Hs = len(range(0, ray_sampler.H, render_stride))
  Ws = len(range(0, ray_sampler.W, render_stride))
  return _frame(chunks, groups, 'outputs_coarse_ref', n_rays, Hs, Ws, dist, world, rank, count, _frame_mode(args))
