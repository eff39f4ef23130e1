"""The feature-encoder convolution kernels under the wave-level emulator on a tiny batch, against the real reference's feature maps."""
import os

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.emu


def test_encoder(emu, golden_dir):
  parity.check_encoder(emu, dict(np.load(os.path.join(golden_dir, 'encoder.npz'))), 'small')
