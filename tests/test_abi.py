"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/dynibar_hip.h declares; struct layouts parsed from the header match the compiled library's expectations; the product
path refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from dynibar_amd import _lib


def test_library_exports_every_declared_symbol():
  lib = _lib.lib()  # binds every prototype parsed from the header; a missing export raises AttributeError
  src = open(_lib.HEADER).read()
  declared = set(re.findall(r'\b(dyn_\w+)\s*\(', _lib._strip_comments(src)))
  assert len(declared) >= 25
  for name in declared:
    assert hasattr(lib, name), f'{name} is declared in include/dynibar_hip.h but not exported by libdynibar_hip.so'
  assert lib.dyn_abi_version() == 1
  assert lib.dyn_profile_count() > 10 and lib.dyn_profile_name(0).decode().startswith('k_')


def test_blob_and_workspace_sizes():
  lib = _lib.lib()
  assert lib.dyn_static_net_blob_floats() > 100 * 4096
  assert lib.dyn_dynamic_net_blob_floats() > 100 * 4096
  assert lib.dyn_motion_mlp_blob_floats() > 100 * 4096
  assert lib.dyn_static_net_workspace_bytes(4096, 64, 8) > lib.dyn_dynamic_net_workspace_bytes(4096, 64, 8) > 0
  assert lib.dyn_static_net_workspace_bytes(16, 64, 33) == 0  # more than 32 views is unsupported and says so


def test_invalid_arguments_are_reported():
  lib = _lib.lib()
  rc = lib.dyn_project_gather(None, None)
  assert rc == -1 and b'null params' in lib.dyn_last_error()
  with pytest.raises(RuntimeError, match='dyn_sample_along_ray failed'):
    _lib.call('dyn_sample_along_ray', None, None)


def test_gather_entry_validates_without_a_device():
  """dyn_gather_tiles / dyn_comm_* (the RCCL pixel gather): exported, argument-checked on the host, RCCL resolved at run time (no link dependency)."""
  lib = _lib.lib()
  assert lib.dyn_comm_available() in (0, 1)
  assert lib.dyn_gather_tiles(None, None, 0, 0, None, None) == -1 and b'dyn_gather_tiles' in lib.dyn_last_error()
  assert lib.dyn_comm_init_rank(None, 0, None, 0) == -1 and b'dyn_comm_init_rank' in lib.dyn_last_error()
  assert lib.dyn_comm_destroy(None) == 0  # destroying nothing is not an error
  import subprocess
  out = subprocess.run(['readelf', '-d', _lib.LIB_PATH], capture_output=True, text=True).stdout
  assert 'librccl' not in out, 'the kernels\' library must not link RCCL (a PyTorch host has loaded its own copy)'


def test_weight_packing_runs_on_the_host():
  """Packing is host code: it must work (and validate its input) without a GPU."""
  from dynibar_amd import ops, synthetic as syn
  blob = ops._pack('dyn_static_net_pack', 'dyn_static_net_blob_floats', ops.STATIC_TENSORS, syn.make_weights('static', 0), 32, 'static')
  assert blob.shape[0] == _lib.lib().dyn_static_net_blob_floats() and bool(torch.isfinite(blob).all())
  assert float(blob.abs().sum()) > 0
  with pytest.raises(RuntimeError, match='32 feature channels'):
    ops._pack('dyn_static_net_pack', 'dyn_static_net_blob_floats', ops.STATIC_TENSORS, syn.make_weights('static', 0, F=16), 16, 'static')


def test_no_cpu_fallback():
  """Kernels refuse host tensors: there is no eager / CPU path in the product package."""
  from dynibar_amd import ops
  o = torch.zeros(4, 3)
  with pytest.raises(RuntimeError, match='HIP device'):
    ops.sample_along_ray(o, o, torch.tensor([[1.0, 2.0]]), 8, True)


def test_product_package_does_not_import_the_oracle():
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for dirpath, _, files in os.walk(os.path.join(root, 'dynibar_amd')):
    for f in files:
      if f.endswith(('.py', '.hip', '.h')):
        text = open(os.path.join(dirpath, f)).read()
        assert 'import oracle' not in text and 'from oracle' not in text and '/root/reference' not in text, f'{f} reaches for test infrastructure'


def test_state_dict_encoder_source_with_trainable_tensors_takes_the_training_form():
  """feature_network.ResNet wrapping a {name: tensor} source: tensors that require grad make the call carry a graph (train_encoder), like a module's
  parameters do -- it must not silently run the forward-only kernels and leave the optimizer's tensors without gradients."""
  from dynibar_amd import feature_network, synthetic as syn
  sd = {k: torch.from_numpy(v) for k, v in syn.make_encoder_weights(0).items()}
  net = feature_network.ResNet.from_module(sd)
  assert net._trains() is False
  for v in sd.values():
    v.requires_grad_(True)
  assert net._trains() is True
  with torch.no_grad():
    assert net._trains() is False
  k0 = net._state()[1]
  with torch.no_grad():
    next(iter(sd.values())).add_(1.0)  # an optimizer step on the source changes the packing key
  assert net._state()[1] != k0
