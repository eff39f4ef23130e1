"""DynIBaR checkpoint files -> a model object the drop-in renderer takes.

The reference saves plain ``torch.save`` dictionaries of de-parallelised state dicts (model.py:177-190 DynibarFF fine stage, :424-441
DynibarMono / the coarse stage DynibarFF loads with load_coarse_model :192-209):
  coarse / monocular file: net_coarse_st, net_coarse_dy, feature_net, [feature_net_st], motion_mlp, traj_basis, global_step, optimizer, scheduler
  fine file              : net_fine_st, net_fine_dy, feature_net_fine, motion_mlp_fine, traj_basis_fine, global_step, optimizer, scheduler
``load_model`` reads them (files or already-loaded dictionaries) into a namespace with the attribute names ``render_rays_mv`` /
``render_rays_mono`` read (net_coarse_st, net_coarse_dy, motion_mlp, trajectory_basis, net_fine_*, motion_mlp_fine,
trajectory_basis_fine) plus the HIP feature encoders (feature_net, feature_net_st, feature_net_fine).  The networks stay state dicts:
the adapter packs them into MFMA operand images on first use (shape-checked; optimizer / scheduler entries are ignored).
"""
from __future__ import annotations

import types

import torch

from . import feature_network

_NETS = {'net_coarse_st': 'net_coarse_st', 'net_coarse_dy': 'net_coarse_dy', 'motion_mlp': 'motion_mlp', 'net_fine_st': 'net_fine_st',
         'net_fine_dy': 'net_fine_dy', 'motion_mlp_fine': 'motion_mlp_fine'}
_BASES = {'traj_basis': 'trajectory_basis', 'traj_basis_fine': 'trajectory_basis_fine'}
_ENCODERS = ('feature_net', 'feature_net_st', 'feature_net_fine')


def _read(src):
  if isinstance(src, dict):
    return src
  try:
    return torch.load(src, map_location='cpu', weights_only=False)
  except TypeError:  # older torch without weights_only
    return torch.load(src, map_location='cpu')


def load_model(coarse, fine=None, device='cuda:0', dynamic_shift=None):
  """coarse: the coarse-stage / monocular checkpoint (path or dict); fine: the Nvidia-benchmark fine-stage checkpoint or None.
  dynamic_shift: the ``shift`` DynibarDynamic was constructed with (a constructor argument, not part of the state dict: 5.0 in
  DynibarMono, model.py:304-309; 0 in DynibarFF) -- default: 5.0 when the file is a monocular one (has feature_net_st), else 0."""
  model = types.SimpleNamespace()
  ck = _read(coarse)
  files = [ck] + ([_read(fine)] if fine is not None else [])
  mono = 'feature_net_st' in ck
  shift = float(dynamic_shift) if dynamic_shift is not None else (5.0 if mono else 0.0)
  for f in files:
    for key, attr in _NETS.items():
      if key in f:
        sd = {k: v.detach().cpu() for k, v in f[key].items()}
        if key.endswith('_dy'):
          sd = _WithAttrs(sd, shift=shift)
        setattr(model, attr, sd)
    for key, attr in _BASES.items():
      if key in f:
        setattr(model, attr, torch.as_tensor(f[key]).detach().float().to(device))
    for key in _ENCODERS:
      if key in f:
        setattr(model, key, feature_network.ResNet.from_module({k: v.detach().cpu() for k, v in f[key].items()}))
    if 'global_step' in f:
      model.global_step = int(f['global_step'])
  if not hasattr(model, 'net_coarse_st'):
    raise KeyError("checkpoint has no 'net_coarse_st' (keys: %s)" % sorted(ck.keys()))
  return model


class _WithAttrs(dict):
  """A state dict that also carries constructor arguments the adapter reads as attributes (``shift`` of DynibarDynamic)."""

  def __init__(self, d, **attrs):
    super().__init__(d)
    for k, v in attrs.items():
      setattr(self, k, v)
