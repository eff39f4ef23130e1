import ctypes
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


@pytest.fixture(scope='module')
def emu():
  """Build the csrc sources against the wave-level emulator and point the ctypes binding at it for the duration of one test module."""
  import emu_build
  from dynibar_amd import _lib
  path = emu_build.build()
  _lib._install_for_tests(ctypes.CDLL(path), require_device=False)
  yield 'cpu'
  _lib._install_for_tests(None, require_device=True)
