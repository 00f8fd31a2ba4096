"""ctypes binding of libsherf_hip.so (the C ABI declared in include/sherf_hip.h).

Mirrors the reference's lazy native-plugin idiom (sherf/torch_utils/ops/bias_act.py:37-50: module-level
`_plugin = None`, `_init()` on first use) -- except that there is NO fallback: if the library is missing or a
tensor is not on the GPU the call raises, because a silent CPU path would invalidate every parity claim.
Prototypes are parsed from the header, so the header is the single source of truth for the ABI.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, '..', 'include', 'sherf_hip.h')
LIB_PATH = os.path.join(_HERE, 'libsherf_hip.so')

_lib = None
_protos = None


class VoxLevel(ctypes.Structure):
    _fields_ = [('wp', ctypes.c_void_p), ('rows', ctypes.c_void_p),
                ('D', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32)]


_SCALARS = {'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
            'sherf_stream_t': ctypes.c_void_p}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, is_pointer, name)])} for every `int sherf_*(...)` / `const char* sherf_*` prototype."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'(int|const char\*)\s+(sherf_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                is_ptr = '*' in a
                base = a.replace('const ', '').replace('*', ' ').split()
                ctype = ctypes.c_void_p if is_ptr else _SCALARS[base[0]]
                parsed.append((ctype, is_ptr, base[-1]))
        protos[name] = (ctypes.c_char_p if ret != 'int' else ctypes.c_int, parsed)
    return protos


def _init():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: build it with `python -m sherf_amd.build` '
                           '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, args) in _protos.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype = ret
        fn.argtypes = [a[0] for a in args]
    _lib = lib
    return lib


def lib():
    return _init()


def ptr(t, dtype=None):
    """Device pointer of a dense CUDA tensor (argument checks in the spirit of bias_act.cpp:39-55)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f'expected a tensor, got {type(t)}')
    if not t.is_cuda:
        raise RuntimeError('sherf_amd: tensor is not on a GPU; the HIP path has no CPU fallback')
    if not t.is_contiguous():
        raise RuntimeError('sherf_amd: tensor must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f'sherf_amd: expected dtype {dtype}, got {t.dtype}')
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    l = _init()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {l.sherf_last_error().decode()}')
