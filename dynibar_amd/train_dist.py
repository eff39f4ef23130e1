"""Data-parallel training across the GPUs of a node: one process per GPU, every rank renders its own N_rand rays, and the gradients
are averaged with ONE bucketed all-reduce per step over RCCL / xGMI ("a gradient all-reduce of the trainable
parameters", section 8(f)3).

The reference wraps its nets in ``nn.DataParallel`` (model.py:130-159), whose gradient reduction happens inside ``module.forward`` /
replicate; this package reads the modules' parameters instead of calling their ``forward``, so the reduction is an explicit call after
``loss.backward()``::

    loss.backward()
    dynibar_amd.train_dist.allreduce_gradients(dynibar_amd.train_dist.trainable_parameters(model))   # the one added line
    model.optimizer.step()

All trainable parameters of DynibarMono are 1.34 M (three MLPs) + 2 x 0.24 M used ResNet weights + the DCT basis: about 7 MB, i.e. a
single bucket and a single latency-bound collective per step.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def trainable_parameters(model, names=('net_coarse_st', 'net_coarse_dy', 'motion_mlp', 'feature_net', 'feature_net_st', 'trajectory_basis')):
  """The parameters the reference's optimizer steps (model.py:339-378): those of the named sub-modules (DataParallel-wrapped or not) and
  bare nn.Parameter attributes, each once."""
  out, seen = [], set()
  for n in names:
    obj = getattr(model, n, None)
    if obj is None:
      continue
    if isinstance(obj, torch.Tensor):
      ps = [obj]
    elif hasattr(obj, 'parameters'):
      ps = list(obj.parameters())
    elif isinstance(obj, dict):  # a state dict (checkpoint.load_model keeps the nets that way): its tensors are the parameters
      ps = [v for v in obj.values() if isinstance(v, torch.Tensor)]
      if ps and not any(p.requires_grad for p in ps):
        raise RuntimeError(f'trainable_parameters: model.{n} is an inference-only state dict (no tensor requires grad); build the model from '
                           'nn.Modules (or call requires_grad_() on its tensors) before training')
    else:
      raise RuntimeError(f'trainable_parameters: model.{n} ({type(obj).__name__}) has no parameters(); it is an inference-only object')
    for p in ps:
      if p.requires_grad and id(p) not in seen:
        seen.add(id(p))
        out.append(p)
  return out


def allreduce_gradients(params, average=True, bucket_bytes=64 << 20, group=None):
  """In-place mean (or sum) of ``p.grad`` over the ranks of the process group: gradients are packed into contiguous fp32 buckets (one for a
  DynibarMono-sized model), each all-reduced once.  A parameter without a gradient on this rank contributes zeros (every rank issues the
  same collectives in the same order).  Returns the number of collectives issued; a no-op when torch.distributed is not initialised."""
  if not (dist.is_available() and dist.is_initialized()):
    return 0
  world = dist.get_world_size(group)
  if world == 1:
    return 0
  params = [p for p in params if p.requires_grad]
  n_coll, i = 0, 0
  while i < len(params):
    j, size = i, 0
    while j < len(params) and (j == i or size + params[j].numel() * 4 <= bucket_bytes):
      size += params[j].numel() * 4
      j += 1
    chunk = params[i:j]
    dev = chunk[0].device
    flat = torch.zeros(sum(p.numel() for p in chunk), dtype=torch.float32, device=dev)
    off = 0
    for p in chunk:
      if p.grad is not None:
        flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
      off += p.numel()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
      flat.div_(world)
    off = 0
    for p in chunk:
      g = flat[off:off + p.numel()].view_as(p)
      if p.grad is None:
        p.grad = g.clone()
      else:
        p.grad.copy_(g)
      off += p.numel()
    n_coll += 1
    i = j
  return n_coll
