"""Build the csrc sources against the wave-level HIP emulator (tests/emu/hip/hip_runtime.h) -> _build/libdynibar_emu.so.
TEST INFRASTRUCTURE ONLY: the product package never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'dynibar_amd', 'csrc')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'libdynibar_emu.so')
CXX = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
UNITS = [('dyn_geometry.hip', ['-ffp-contract=off']), ('dyn_nets.hip', []), ('dyn_encoder.hip', []), ('dyn_train.hip', []), ('dyn_comm.hip', [])]


def build(opt='-O2', sanitize=False):
  os.makedirs(OUT_DIR, exist_ok=True)
  deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
  deps += [os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(HERE, 'emu_runtime.cpp'), os.path.join(ROOT, 'include', 'dynibar_hip.h')]
  out = OUT.replace('.so', '_asan.so') if sanitize else OUT
  if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
    return out
  common = [CXX, '-std=c++20', opt, '-g', '-pthread', '-fPIC', '-I', HERE, '-Wno-unused-function']
  if sanitize:
    common += ['-fsanitize=address', '-fno-omit-frame-pointer']
  objs, procs = [], []
  shared = [d for d in deps if not d.endswith('.hip')]  # headers and the emulator runtime: every unit depends on them

  def fresh(obj, *srcs):  # an object newer than its own source and every shared header is reused (only the changed unit recompiles)
    return os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in list(srcs) + shared)

  for src, flags in UNITS:  # the translation units compile side by side (dyn_nets.hip alone takes minutes at -O2)
    path = os.path.join(CSRC, src)
    if not os.path.exists(path):
      continue
    obj = os.path.join(OUT_DIR, src.replace('.hip', '_asan.o' if sanitize else '.o'))
    if not fresh(obj, path):
      procs.append((src, subprocess.Popen(common + flags + ['-x', 'c++', '-c', path, '-o', obj])))
    objs.append(obj)
  obj = os.path.join(OUT_DIR, 'emu_runtime_asan.o' if sanitize else 'emu_runtime.o')
  if not fresh(obj):
    procs.append(('emu_runtime.cpp', subprocess.Popen(common + ['-c', os.path.join(HERE, 'emu_runtime.cpp'), '-o', obj])))
  objs.append(obj)
  for src, pr in procs:
    if pr.wait() != 0:
      raise RuntimeError(f'emulator build: compiling {src} failed')
  subprocess.check_call(common + ['-shared'] + objs + ['-o', out])
  return out


if __name__ == '__main__':
  print(build())
