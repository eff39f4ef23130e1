"""The MFMA network kernels (weight ring, register-resident layer chains, ray attention) executed under the wave-level emulator on
one ray, checked against the oracle.  Debugging aid for fragment maps / packing in a container without a GPU; -m gpu is authoritative."""
import os

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.emu


def test_mlp_engine(emu):
  parity.check_mlp_selftest(emu, rows=70)


def test_static_net(emu):
  parity.check_static_net(emu, 'small', S=32, R=1)


def test_static_net_generic_views(emu):
  parity.check_static_net(emu, 'harsh', S=32, R=1, aa=False, mask_rgb=True)  # 11 views: the non-power-of-two segment path


def test_dynamic_net(emu):
  parity.check_dynamic_net(emu, 'small', S=32, R=1, shift=5.0)


def test_motion_mlp(emu):
  parity.check_motion(emu, 'small', S=32, R=1)


def test_dense_rows_dynamic_and_static(emu):
  """View counts from 9 that are not powers of two run the dense-row flavour (cross-view reductions through LDS tables): 13 dynamic and 20 static views."""
  parity.check_dynamic_net(emu, 'many', S=32, R=1)
  parity.check_static_net(emu, 'many', S=32, R=1)


def test_point_kernel_walks_several_row_tiles(emu):
  """12 rays x 32 samples = 3 workgroups of k_net_points.  (In a -DDYN_POINTS_PERSIST=1 build an emulator with fewer "CUs" than workgroups makes some run a second
  pass: slot rotation of the weight ring, the next pass's first chunks requested in the tail of the first.)"""
  parity.check_static_net(emu, "small", S=32, R=12)
  parity.check_dynamic_net(emu, "small", S=32, R=12, shift=5.0)


def test_ragged_rows_with_dark_colours(emu):
  """11 static views (ragged dense rows) with mask_rgb removing valid rows and whole points: the blend's product mask and its all-masked path."""
  parity.check_static_net(emu, 'harsh', S=32, R=2, mask_rgb=True, dark=0.4)


@pytest.mark.parametrize('name', ['cross_views', 'cross_rays_samples'])
def test_cross_axis_shapes(emu, golden_dir, name):
  """3 static source views (the reference's torch.cross runs over the views) and a chunk of 3 rays x 3 samples (over the rays, for both moments)."""
  parity.check_cross_axis(emu, dict(np.load(os.path.join(golden_dir, 'cross_axis.npz'))), name)


def test_accuracy_against_double(emu):
  parity.check_accuracy_against_double(emu, 'small', S=16, R=2)


def test_dual_branch_accuracy_against_double(emu):
  parity.check_dual_accuracy_against_double(emu, 'small', S=16, R=2)
