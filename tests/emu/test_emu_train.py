"""The training kernels (csrc/dyn_train.hip, the gather backward) under the wave-level emulator: a tiny static bootstrap step, values and
every gradient against autograd through the oracle.  Debugging aid in a container without a GPU; -m gpu is authoritative."""
import os

import pytest
import torch

import parity

pytestmark = pytest.mark.emu


def test_train_gemm_modes(emu):
  parity.check_train_gemm(emu)


def test_train_gemm_random_shapes(emu):
  parity.check_train_gemm_fuzz(emu, n_cases=9, max_rows=400)


def test_train_composite(emu):
  parity.check_train_composite(emu, lengths=(5, 64, 100), R=3)


def test_train_attention(emu):
  parity.check_train_attention(emu, lengths=(5, 16, 37, 120), R=2)


def test_static_bootstrap_step(emu):
  parity.check_train_static(emu, 'few', S=8, R=2)


def test_static_bootstrap_step_three_views(emu):
  """3 static source views: the moments of the training embed kernel follow the reference's torch.cross over the view axis (csrc/dyn_device.h)."""
  parity.check_train_static(emu, 'cross_views', S=8, R=2)


def test_static_bootstrap_step_kid_config(emu):
  parity.check_train_static(emu, 'few', S=8, R=2, aa=False, mask_rgb=True)


def test_dual_branch_step(emu):
  """second slice: DynibarDynamic + raw2outputs, gradients to both nets and both feature-map sets"""
  parity.check_train_dual(emu, 'few', S=8, R=2)


@pytest.mark.skipif(not os.environ.get('DYN_EMU_FULL'), reason='17 minutes under the emulator: set DYN_EMU_FULL=1 (the -m gpu suite runs the same check on hardware)')
def test_full_training_iteration(emu, golden_dir):
  """third slice: render_rays_mono(is_train=True) under grad mode, the reference's main-loop loss, every gradient incl. MotionMLP and the
  trajectory basis against the real reference's autograd digests"""
  import os
  import numpy as np
  parity.check_train_mono(emu, dict(np.load(os.path.join(golden_dir, 'mono_train_grad.npz'))), losses=('full', 'flow', 'cycle'))
