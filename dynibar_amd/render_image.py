"""Drop-in for the reference's ``ibrnet/render_image.py``: ``render_single_image_nvi`` / ``render_single_image_mono`` with the
reference signatures and return structure (reference render_image.py:9-217, :220-439).

Multi-GPU: rays are independent, so when ``torch.distributed`` is initialised (one process per GPU, RCCL) the H*W rays of the
target view are split into ``world_size`` contiguous, equally padded tiles; every rank renders its own tile chunk by chunk and
the tiles are concatenated with ONE all-gather per output tensor (reference: ``nn.DataParallel`` scatter/gather inside every
module call, model.py:134-159).  Every rank returns the full frame.  With world_size == 1 no collective is issued.
Chunk results stay on the device until the frame is assembled (the reference copies every tensor of every chunk to the host).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from .render_ray import render_rays_mono, render_rays_mv

_PER_VIEW_KEYS = ('camera', 'anchor_camera', 'render_camera', 'depth_range', 'src_rgbs', 'src_cameras', 'anchor_src_rgbs', 'anchor_src_cameras', 'static_src_rgbs',
                  'static_src_cameras')


def _dist():
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
    return dist, dist.get_world_size(), dist.get_rank()
  return None, 1, 0


def ray_tile(n_rays, world, rank):
  """Contiguous tile [lo, hi) of rank ``rank``; all tiles have the padded size ``tile`` except that the last ones may be short/empty."""
  tile = (n_rays + world - 1) // world
  lo = min(rank * tile, n_rays)
  return lo, min(lo + tile, n_rays), tile


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


def gather_ray_outputs(local, n_rays, dist, world, rank):
  """local: {key: tensor} covering this rank's tile (ray axis 0, or 1 for 3-D [V,N,c] tensors; 4-D tensors are dropped like the
  reference drops them).  Returns {key: full tensor over all n_rays} on every rank, via one all_gather per key of equal padded tiles."""
  lo, hi, tile = ray_tile(n_rays, world, rank)
  out = OrderedDict()
  for k, t in local.items():
    if t is None or t.dim() == 4:
      continue
    axis = 1 if t.dim() == 3 else 0
    if world == 1:
      out[k] = t
      continue
    tt = t.transpose(0, axis) if axis else t
    pad_shape = (tile,) + tuple(tt.shape[1:])
    send = torch.zeros(pad_shape, dtype=tt.dtype, device=tt.device)
    send[: hi - lo] = tt
    recv = torch.empty((world,) + pad_shape, dtype=tt.dtype, device=tt.device)
    dist.all_gather_into_tensor(recv.view((world * tile,) + pad_shape[1:]), send.contiguous())
    full = recv.view((world * tile,) + pad_shape[1:])[:n_rays]
    out[k] = full.transpose(0, axis).contiguous() if axis else full
  return out


def _assemble(per_chunk, n_rays, Hs, Ws, dist, world, rank):
  """list of per-chunk output dicts -> full-frame dict, reshaped like render_image.py:137-188 and moved to the host."""
  keys = list(per_chunk[0].keys()) if per_chunk else []
  local = OrderedDict()
  lists4 = OrderedDict()
  for k in keys:
    parts = [c[k] for c in per_chunk]
    if parts[0] is None or k == 'random_sigma':
      continue
    if parts[0].dim() == 4:
      lists4[k] = [t.cpu() for t in parts]  # the reference leaves 4-D entries as its per-chunk list of host tensors (render_image.py:368-369)
      continue
    local[k] = torch.cat(parts, dim=1 if parts[0].dim() == 3 else 0)
  full = gather_ray_outputs(local, n_rays, dist, world, rank)
  # to the host like the reference (.cpu() per tensor, render_image.py:113-118), but as one batch of asynchronous copies into pinned
  # memory and a single synchronisation: the per-sample arrays of a 288x512 frame are ~1 GB, 0.1 s of pageable copies one by one
  host = OrderedDict()
  try:
    for k, t in full.items():
      host[k] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True) if t.is_cuda else t
    if any(t.is_cuda for t in full.values()):
      torch.cuda.current_stream().synchronize()
  except RuntimeError:  # no pinned memory to be had: plain synchronous copies
    host = OrderedDict((k, t.cpu()) for k, t in full.items())
  ret = OrderedDict()
  for k in keys:  # the reference's key order
    if k in lists4:
      ret[k] = lists4[k]
  for k, t in host.items():
    if t.dim() == 3:
      ret[k] = t.reshape((t.shape[0], Hs, Ws, -1)).squeeze()
    else:
      ret[k] = t.reshape((Hs, Ws, -1)).squeeze()
  if 'rgb' in ret and 'mask' in ret:
    ret['rgb'][ret['mask'] == 0] = 0.0
  return ret


def _render_tiles(ray_batch, chunk_size, render_chunk, group_names):
  dist, world, rank = _dist()
  n_rays = ray_batch['ray_o'].shape[0]
  lo, hi, _ = ray_tile(n_rays, world, rank)
  chunks = {g: [] for g in group_names}
  for i in range(lo, hi, chunk_size):
    ret = render_chunk(slice_ray_batch(ray_batch, i, min(i + chunk_size, hi)))
    for g in group_names:
      if ret.get(g) is not None:
        chunks[g].append(ret[g])
  return chunks, n_rays, dist, world, rank


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
  chunks, n_rays, dist, world, rank = _render_tiles(ray_batch, chunk_size, render_chunk, groups)
  Hs = len(range(0, ray_sampler.H, render_stride))
  Ws = len(range(0, ray_sampler.W, render_stride))
  all_ret = OrderedDict()
  for g in groups:
    all_ret[g] = _assemble(chunks[g], n_rays, Hs, Ws, dist, world, rank) if chunks[g] else OrderedDict()
  all_ret['outputs_fine'] = None
  return all_ret


def render_single_image_mono(frame_idx, time_embedding, time_offset, ray_sampler, ray_batch, model, projector, chunk_size, N_samples, args,
                             inv_uniform=False, N_importance=0, det=False, white_bkgd=False, render_stride=1, featmaps=None, is_train=True,
                             num_vv=2):
  """Reference render_image.py:220-439: full-frame coarse rendering of one target view (monocular video)."""
  def render_chunk(chunk):
    return render_rays_mono(frame_idx=frame_idx, time_embedding=time_embedding, time_offset=time_offset, ray_batch=chunk, model=model,
                            featmaps=featmaps, projector=projector, N_samples=N_samples, args=args, inv_uniform=inv_uniform,
                            N_importance=N_importance, raw_noise_std=0.0, det=det, white_bkgd=white_bkgd, is_train=is_train, num_vv=num_vv)

  groups = ('outputs_coarse_ref', 'outputs_coarse_st', 'outputs_coarse_anchor')
  chunks, n_rays, dist, world, rank = _render_tiles(ray_batch, chunk_size, render_chunk, groups)
  Hs = len(range(0, ray_sampler.H, render_stride))
  Ws = len(range(0, ray_sampler.W, render_stride))
  all_ret = OrderedDict()
  for g in groups:
    all_ret[g] = _assemble(chunks[g], n_rays, Hs, Ws, dist, world, rank) if chunks[g] else OrderedDict()
  all_ret['outputs_fine'] = None
  return all_ret
