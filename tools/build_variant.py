"""Developer A/B builds: python tools/build_variant.py <tag> [--geom FLAGS...] [--nets FLAGS...]  -> dynibar_amd/csrc/libdynibar_hip_<tag>.so
(select with DYNIBAR_HIP_LIB=<path>).  Units without extra flags reuse the default build's objects."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dynibar_amd', 'csrc')
COMMON = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
tag = sys.argv[1]
rest = sys.argv[2:]
geom, nets, cur = [], [], None
for a in rest:
  if a == '--geom': cur = geom
  elif a == '--nets': cur = nets
  else: cur.append(a)
objs = []
for src, base, extra in (('dyn_geometry.hip', ['-ffp-contract=off', '-munsafe-fp-atomics'], geom), ('dyn_nets.hip', [], nets), ('dyn_encoder.hip', ['-munsafe-fp-atomics'], []),
                         ('dyn_train.hip', ['-munsafe-fp-atomics'], []), ('dyn_comm.hip', [], [])):
  obj = os.path.join(CSRC, src.replace('.hip', '.o'))
  if extra:
    obj = os.path.join(CSRC, src.replace('.hip', f'_{tag}.o'))
    subprocess.check_call(COMMON + base + extra + ['-c', os.path.join(CSRC, src), '-o', obj])
  objs.append(obj)
out = os.path.join(CSRC, f'libdynibar_hip_{tag}.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-ldl', '-o', out])
print(out)
