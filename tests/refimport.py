"""Import the real reference (read-only, build container only) for golden generation / live pinning.

/root/reference does not exist on the GPU box; everything that uses this module must skip when
``have_reference()`` is false.  The only missing dependency of the hot-path modules is
kornia.create_meshgrid (reference ibrnet/sample_ray.py:6,83), shimmed below.
"""
import os
import sys
import types

REF_ROOT = '/root/reference'


def have_reference():
  return os.path.isdir(os.path.join(REF_ROOT, 'ibrnet'))


def import_reference():
  import torch
  sys.dont_write_bytecode = True  # never write __pycache__ into the reference tree
  if 'kornia' not in sys.modules:
    k = types.ModuleType('kornia')

    def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
      assert not normalized_coordinates
      xs = torch.linspace(0, width - 1, width, dtype=dtype)
      ys = torch.linspace(0, height - 1, height, dtype=dtype)
      gy, gx = torch.meshgrid(ys, xs, indexing='ij')
      return torch.stack([gx, gy], dim=-1)[None]

    k.create_meshgrid = create_meshgrid
    sys.modules['kornia'] = k
  if REF_ROOT not in sys.path:
    sys.path.insert(0, REF_ROOT)
  import importlib
  mods = types.SimpleNamespace()
  for name in ('sample_ray', 'projection', 'render_ray', 'mlp_network', 'render_image', 'feature_network'):
    setattr(mods, name, importlib.import_module('ibrnet.' + name))
  mods.init_dct_basis = importlib.import_module('ibrnet.model').init_dct_basis
  return mods
