"""The geometry / compositing / resampling kernels executed under the wave-level emulator, checked against the oracle.
This is a debugging aid for kernel indexing in a container without a GPU; the authoritative parity run is -m gpu."""
import os

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.emu


def test_sampling(emu):
  parity.check_sampling(emu, 'small')


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_project_gather(emu, name):
  parity.check_project_gather(emu, name)


def test_composite(emu):
  parity.check_composite(emu, R=9, S=64)
  parity.check_composite(emu, R=6, S=128, seed=1)


def test_fine_samples(emu, golden_dir):
  g = dict(np.load(os.path.join(golden_dir, 'stages_small.npz')))
  parity.check_fine_samples(emu, g)


def test_module_helper_exports(emu, golden_dir):
  g = dict(np.load(os.path.join(golden_dir, 'stages_small.npz')))
  parity.check_module_helpers(emu, g, 'small', with_fine=False)


@pytest.mark.parametrize('name', ['small', 'harsh', 'noise'])
def test_project_gather_same_matrix(emu, name):
  parity.check_project_gather_same_matrix(emu, name)


def test_projector_helper_methods(emu):
  print('  compute_angle max |err| vs oracle:', parity.check_projector_helpers(emu, 'small'))


def test_trajectory_points_fused_into_gather_and_flows(emu):
  parity.check_fused_trajectory(emu, 'small', S=16, R=3)
  parity.check_fused_trajectory(emu, 'kid', S=8, R=2, virtual_views=3)


def test_expected_scene_flow(emu):
  parity.check_expected_scene_flow(emu)
