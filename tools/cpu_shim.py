"""TEST / DIAGNOSTIC INFRASTRUCTURE: points sherf_amd at the HOST builds of its kernel libraries (tests/hipcpu) outside pytest, the way the
`cpu_product` fixture of tests/test_hipcpu_frame.py does -- so that a GPU diagnostic (tools/enc_sp_diag.py, ...) can be run on the CPU build of
the same sources first, and its numbers compared with the hardware's.   import tools.cpu_shim; tools.cpu_shim.enable()"""
import ctypes
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def enable():
    from sherf_amd import _lib
    from sherf_amd.build import SOURCES
    from tests import gpu_common as G
    from tests.hipcpu import build_cpu
    tmp = tempfile.mkdtemp(prefix='hipcpu_')
    fwd = build_cpu.build('sherf_hipcpu_full', SOURCES, os.path.join(tmp, 'full'), compiler=build_cpu.CLANG)
    _lib.LIB_PATH, _lib._lib = fwd, None
    _lib.ptr = lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
    _lib.stream = lambda: ctypes.c_void_p(0)
    torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
    torch.cuda.synchronize = lambda dev=None: None
    from sherf_amd.renderer import ImportanceRenderer
    ImportanceRenderer.SMPL_NEUTRAL = property(lambda self: self._smpl(torch.device('cpu')))
    G.CPU_SHIM = True
    G.hip_modules.cache_clear()
    return G
