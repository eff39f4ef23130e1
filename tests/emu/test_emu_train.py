"""The training kernels (csrc/dyn_train.hip, the gather backward) under the wave-level emulator: a tiny static bootstrap step, values and
every gradient against autograd through the oracle.  Debugging aid in a container without a GPU; -m gpu is authoritative."""
import pytest
import torch

import parity

pytestmark = pytest.mark.emu


def test_train_gemm_modes(emu):
  parity.check_train_gemm(emu)


def test_static_bootstrap_step(emu):
  parity.check_train_static(emu, 'few', S=8, R=2)


def test_static_bootstrap_step_kid_config(emu):
  parity.check_train_static(emu, 'few', S=8, R=2, aa=False, mask_rgb=True)


def test_dual_branch_step(emu):
  """second slice: DynibarDynamic + raw2outputs, gradients to both nets and both feature-map sets"""
  parity.check_train_dual(emu, 'few', S=8, R=2)
