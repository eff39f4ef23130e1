"""The feature-encoder convolution kernels under the wave-level emulator on a tiny batch, against the real reference's feature maps."""
import os

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.emu


def test_encoder(emu, golden_dir):
  parity.check_encoder(emu, dict(np.load(os.path.join(golden_dir, 'encoder.npz'))), 'small')


def test_encoder_training_form(emu):
  """forward with saved activations + backward (im2col + training GEMM + InstanceNorm kernels) vs autograd through the oracle"""
  parity.check_encoder_training(emu, 'tiny')
