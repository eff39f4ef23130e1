"""ctypes binding of libdynibar_hip.so, generated from include/dynibar_hip.h.

The header is the single source of truth: struct layouts and prototypes are parsed from it, so the Python side cannot
drift from the C ABI.  There is no fallback: if the library has not been built (``python -m dynibar_amd.build``)
``lib()`` raises, and every tensor handed to a kernel must live on a HIP device.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'dynibar_hip.h')
LIB_PATH = os.environ.get('DYNIBAR_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libdynibar_hip.so')  # env: developer A/B builds

_CTYPES = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64,
    'size_t': ctypes.c_size_t, 'void': None, 'long': ctypes.c_long, 'unsigned': ctypes.c_uint,
}


def _strip_comments(src):
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return re.sub(r'//[^\n]*', '', src)


def _ctype_of(decl):
  """'const float*' -> c_void_p, 'int' -> c_int, 'const char*' -> c_char_p."""
  d = decl.replace('const', '').strip()
  if d.endswith('*'):
    return ctypes.c_char_p if d[:-1].strip() == 'char' else ctypes.c_void_p
  return _CTYPES[d]


def parse_header(path=HEADER):
  """-> (structs: {name: [(field, ctype)]}, funcs: {name: (restype, [(argname, ctype_or_structname)])})."""
  src = _strip_comments(open(path).read())
  structs = {}
  for body, name in re.findall(r'typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
    fields = []
    for stmt in body.split(';'):
      stmt = ' '.join(stmt.split())
      if not stmt:
        continue
      m = re.match(r'((?:const\s+)?\w+\s*\*?)\s*(.*)', stmt)
      base, names = m.group(1).strip(), m.group(2)
      for nm in names.split(','):
        nm = nm.strip()
        star = nm.startswith('*')
        nm = nm.lstrip('* ')
        fields.append((nm, _ctype_of(base + ('*' if star else ''))))
    structs[name] = fields
  funcs = {}
  nostruct = re.sub(r'typedef\s+struct\s*\{.*?\}\s*\w+\s*;', '', src, flags=re.S)
  for ret, name, args in re.findall(r'\n\s*((?:const\s+)?\w+\s*\*?)\s*(dyn_\w+)\s*\(([^)]*)\)\s*;', nostruct):
    alist = []
    args = ' '.join(args.split())
    if args and args != 'void':
      for a in args.split(','):
        a = a.strip()
        m = re.match(r'(.*?)(\w+)$', a)
        typ, an = m.group(1).strip(), m.group(2)
        base = typ.replace('const', '').replace('*', '').strip()
        if base in structs:
          alist.append((an, base))
        else:
          alist.append((an, _ctype_of(typ)))
    funcs[name] = (_ctype_of(ret), alist)
  return structs, funcs


_STRUCT_SPECS, _FUNC_SPECS = parse_header()
STRUCTS = {}
for _n, _f in _STRUCT_SPECS.items():
  STRUCTS[_n] = type(_n, (ctypes.Structure,), {'_fields_': _f})

_LIB = None
_REQUIRE_DEVICE = True


def _bind(cdll):
  for name, (restype, args) in _FUNC_SPECS.items():
    fn = getattr(cdll, name)  # AttributeError here = the library does not export a declared symbol
    fn.restype = restype
    fn.argtypes = [ctypes.POINTER(STRUCTS[t]) if isinstance(t, str) else t for _, t in args]
  return cdll


def lib():
  global _LIB
  if _LIB is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          f'{LIB_PATH} is missing: build the gfx950 kernels with `python -m dynibar_amd.build` '
          '(dynibar_amd has no CPU or eager-PyTorch fallback).')
    _LIB = _bind(ctypes.CDLL(LIB_PATH))
    ver = _LIB.dyn_abi_version()
    if ver != 1:
      raise RuntimeError(f'libdynibar_hip.so ABI {ver} != 1')
  return _LIB


def _install_for_tests(cdll, require_device):
  """tests/emu only: point the binding at the wave-level emulator build of the same sources."""
  global _LIB, _REQUIRE_DEVICE
  _LIB = _bind(cdll) if cdll is not None else None
  _REQUIRE_DEVICE = require_device


def ptr(t, dtype=torch.float32):
  """Device pointer of a contiguous tensor (None -> NULL)."""
  if t is None:
    return None
  if t.dtype != dtype:
    raise TypeError(f'expected {dtype}, got {t.dtype}')
  if not t.is_contiguous():
    raise ValueError('tensor must be contiguous')
  if _REQUIRE_DEVICE and not t.is_cuda:
    raise RuntimeError('dynibar_amd kernels need tensors on a HIP device (cuda:N); got ' + str(t.device))
  return ctypes.c_void_p(t.data_ptr())


class _Stream(ctypes.c_void_p):
  """A hipStream_t handle that remembers which device it belongs to (see call())."""
  device_index = None


def stream_of(t):
  """torch's current stream on the tensor's device.  Kernels launch on the CURRENT device and the C side refuses a stream of another one, so
  call() makes the stream's device current for the duration of the launch only -- the caller's current device is never changed."""
  if t is not None and t.is_cuda:
    st = _Stream(torch.cuda.current_stream(t.device).cuda_stream)
    st.device_index = t.device.index
    return st
  return ctypes.c_void_p(0)


def call(name, *args):
  fn = getattr(lib(), name)
  dev = next((a.device_index for a in args if isinstance(a, _Stream)), None)
  if dev is not None and dev != torch.cuda.current_device():
    with torch.cuda.device(dev):
      rc = fn(*args)
  else:
    rc = fn(*args)
  if rc != 0:
    raise RuntimeError(f'{name} failed ({rc}): {lib().dyn_last_error().decode()}')


def params(struct_name, **kw):
  st = STRUCTS[struct_name]()
  known = {f for f, _ in _STRUCT_SPECS[struct_name]}
  for k, v in kw.items():
    if k not in known:
      raise KeyError(f'{struct_name} has no field {k}')
    setattr(st, k, v)
  return st
