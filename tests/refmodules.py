"""Stand-ins for the reference's nn.Modules (test infrastructure).

`/root/reference` does not exist on the GPU box, but the adapter must be exercised there with *real* ``nn.Module`` objects
(optionally wrapped in ``nn.DataParallel`` exactly as model.py:134-159 / :382-397 wrap them), not with dicts.  ``like_reference``
builds a module whose parameters, names, shapes and constructor-dependent attributes (``anti_alias_pooling``, ``mask_rgb``, ``shift``,
``sf_mag_div``, the conditional ``s`` parameter of mlp_network.py:330-331) match DynibarStatic / DynibarDynamic / MotionMLP; it has no
forward.  ``tests/test_model_adapter.py`` proves on the build container that its state dict is key-for-key and shape-for-shape the
real reference module's.
"""
import torch
from torch import nn

from dynibar_amd import synthetic as syn


class _Holder(nn.Module):
  def forward(self, *a, **k):
    raise RuntimeError('stand-in module: the product never calls forward() on the reference modules, it reads their parameters')


def _set_linear(root, dotted, nout, nin, bias):
  """root.<a>.<idx> = nn.Linear(nin, nout): 'ray_attention.w_qs' -> attribute chain, 'base_fc.0' -> index 0 of an nn.Sequential-like."""
  parts = dotted.split('.')
  cur = root
  for p in parts[:-1]:
    if not hasattr(cur, p):
      setattr(cur, p, _Holder())
    cur = getattr(cur, p)
  cur.add_module(parts[-1], nn.Linear(nin, nout, bias=bias))


def like_reference(kind, args=None, num_basis=6, shift=0.0, F=32):
  """kind: 'static' | 'dynamic' | 'motion'.  args: namespace with anti_alias_pooling / mask_rgb (static only)."""
  m = _Holder()
  table = {'static': syn.static_layer_table, 'dynamic': syn.dynamic_layer_table}[kind](F) if kind != 'motion' else syn.motion_layer_table(num_basis)
  for name, nout, nin, has_bias in table:
    _set_linear(m, name, nout, nin, has_bias)
  if kind in ('static', 'dynamic'):
    m.ray_attention.add_module('layer_norm', nn.LayerNorm(128, eps=1e-6))
  if kind == 'static':
    m.anti_alias_pooling = args.anti_alias_pooling
    m.mask_rgb = args.mask_rgb
    if m.anti_alias_pooling:
      m.s = nn.Parameter(torch.tensor(0.2), requires_grad=True)  # only then (mlp_network.py:330-331)
  if kind == 'dynamic':
    m.anti_alias_pooling = False
    m.shift = shift
  if kind == 'motion':
    m.sf_mag_div = 1.0
    m.num_basis = num_basis
  return m


def load_numpy_state(module, sd):
  """Loads a numpy state dict (dynibar_amd.synthetic.make_weights); an `s` entry is dropped when the module has no such parameter."""
  own = module.state_dict()
  module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in own}, strict=True)
  return module.eval()
