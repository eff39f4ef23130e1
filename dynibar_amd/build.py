"""Build libdynibar_hip.so for gfx950 in-tree (dynibar_amd/csrc/).   python -m dynibar_amd.build [--force]

Translation units: the geometry/compositing kernels are compiled with -ffp-contract=off (bit-exact sample depths,
points and indices versus the reference's un-fused fp32 ops), the MFMA network kernels and the feature encoder with default contraction.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(CSRC, 'libdynibar_hip.so')
OUT_X6 = os.path.join(CSRC, 'libdynibar_hip_x6.so')
UNITS = [
    ('dyn_geometry.hip', ['-ffp-contract=off', '-munsafe-fp-atomics']),
    ('dyn_nets.hip', []),
    ('dyn_encoder.hip', ['-munsafe-fp-atomics']),  # (the training form's col2im adds with hardware fp32 atomics, not CAS loops)
    ('dyn_train.hip', ['-munsafe-fp-atomics']),
    ('dyn_comm.hip', []),  # the RCCL pixel gather (RCCL itself is resolved at run time: no link dependency)
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _deps():
  return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))] + [
      os.path.join(os.path.dirname(HERE), 'include', 'dynibar_hip.h')]


STAMP = os.path.join(CSRC, '.build_stamp')


def _stamp():
  """sha256 over the sources, headers and flags the libraries are made from.  (Modification times are not enough: a build that was started before
  an edit finishes after it and leaves a stale library that is newer than its sources.)"""
  import hashlib
  h = hashlib.sha256(repr((COMMON, UNITS)).encode())
  for d in sorted(_deps()):
    h.update(os.path.basename(d).encode())
    with open(d, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def build(force=False, verbose=True):
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  units = [(s, f) for s, f in UNITS if os.path.exists(os.path.join(CSRC, s))]
  stamp = _stamp()
  if not force and os.path.exists(OUT) and os.path.exists(OUT_X6) and os.path.exists(STAMP) and open(STAMP).read().strip() == stamp:
    return OUT
  if os.path.exists(STAMP):
    os.remove(STAMP)
  # the translation units are independent: compile them (and the 6-term flavour of the network unit) side by side
  from concurrent.futures import ThreadPoolExecutor
  jobs = []
  for src, flags in units:
    obj = os.path.join(CSRC, src.replace('.hip', '.o'))
    jobs.append((obj, [hipcc] + COMMON + flags + ['-c', os.path.join(CSRC, src), '-o', obj]))
  # the same library with the fp32-class 6-term split engine: used by the precision A/B test and bench leg
  obj6 = os.path.join(CSRC, 'dyn_nets_x6.o')
  jobs.append((obj6, [hipcc] + COMMON + ['-DDYN_SPLIT_TERMS=6', '-DDYN_SPLIT_F16=0', '-c', os.path.join(CSRC, 'dyn_nets.hip'), '-o', obj6]))

  def run(job):
    if verbose:
      print(' '.join(job[1]), flush=True)
    subprocess.check_call(job[1])
    return job[0]

  with ThreadPoolExecutor(max_workers=int(os.environ.get('DYNIBAR_BUILD_JOBS', '6'))) as ex:
    built = list(ex.map(run, jobs))
  objs = built[:-1]
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-ldl', '-o', OUT]
  if verbose:
    print(' '.join(cmd), flush=True)
  subprocess.check_call(cmd)
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', objs[0], obj6] + objs[2:] + ['-ldl', '-o', OUT_X6]
  subprocess.check_call(cmd)
  if _stamp() == stamp:  # the sources did not change while the compilers ran
    with open(STAMP, 'w') as f:
      f.write(stamp + '\n')
  return OUT


if __name__ == '__main__':
  print(build(force='--force' in sys.argv))
