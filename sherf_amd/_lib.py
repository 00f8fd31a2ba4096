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
# SHERF_HIP_LIB: alternative build of the same ABI (kernel-variant experiments, tools/gpu_variants.sh); default = in-tree build
LIB_PATH = os.environ.get('SHERF_HIP_LIB') or os.path.join(_HERE, 'libsherf_hip.so')

_lib = None
_protos = None


_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64


class VoxLevel(ctypes.Structure):                     # sherf_vox_level
    _fields_ = [('wp', _vp), ('rows', _vp), ('D', _i32), ('H', _i32), ('W', _i32)]


class SvoxLevelWs(ctypes.Structure):                  # sherf_svox_level_ws
    _fields_ = [(n, _vp) for n in ('bitmap', 'prefix', 'n_rows', 'chunk_ws', 'wp', 'keys')] + \
               [(n, _i32) for n in ('n_words', 'cap', 'D', 'H', 'W', 'pad_')]


class SvoxLayer(ctypes.Structure):                    # sherf_svox_layer
    _fields_ = [(n, _i32) for n in ('cin', 'cout', 'down', 'tap')] + \
               [(n, _vp) for n in ('wt', 'gamma', 'beta', 'stats', 'bnparam', 'out', 'acc')]


SVOX_MAX_LAYERS = 16


class SvoxPlan(ctypes.Structure):                     # sherf_svox_plan
    _fields_ = [('lev', SvoxLevelWs * 4), ('layers', SvoxLayer * SVOX_MAX_LAYERS), ('n_layers', _i32), ('pad_', _i32),
                ('acc_fix', _vp), ('g0', _vp), ('mult', _vp), ('n_total', _vp), ('zero_ptr', _vp), ('zero_bytes', _i64),
                ('fold_mat', _vp * 3), ('fold_rows', _vp * 3)]


def _frame_fields():
    f = []
    P = lambda *names: f.extend((n, _vp) for n in names)
    I = lambda *names: f.extend((n, _i32) for n in names)
    P('poses', 'shapes', 'J_template', 'J_shapedirs', 'parents', 'posedirs', 'shapedirs', 'weights',
      'A', 'posefeat', 'PO', 'SO', 'T2C', 'C2S', 'obs_R', 'obs_Th', 'cam_R', 'cam_T', 'cam_K',
      'verts', 'Rg', 'Th', 'tverts', 'grid_hdr', 'cell_start', 'cell_pts', 'cell_scratch', 'near_mask',
      'ray_o', 'ray_d', 'near', 'far')
    I('R', 'S')
    f.append(('capacity', _i64))
    P('counters', 'ray_base', 'ray_cnt', 'cs_idx', 'cs_vid', 'cs_xs', 'dense_vid', 'ray_mask', 'scan_ws')
    P('planes', 'Wa_t', 'planes_f'); I('P', 'flags')
    P('obs_feat', 'Wb_t', 'feat_f'); I('Hf', 'Wf')
    P('obs_img', 'img4'); I('H', 'W')
    P('geom', 'cs_tvid', 'tok_bias', 'bounds', 'vox_min')
    f.append(('vox_sh', _i32 * 3)); I('gather_split')
    P('tokens', 'extras', 'vox_plan', 'vox_coord', 'vox_feat'); I('vox_n', 'vox_training')
    P('wstream', 'wbias'); I('mlp_prec', 'mlp_parts')
    P('sample_out'); I('white_back', 'main_after_layer')
    P('rgb', 'depth', 'acc', 'zfrag', 'near_hdr', 'near_list')
    f.append(('near_list_cap', _i64))
    f.append(('tok_capacity', _i64))
    P('pefrag')
    P('sticky')
    return f


class Frame(ctypes.Structure):                        # sherf_frame
    _fields_ = _frame_fields()


_STRUCTS = (VoxLevel, SvoxLevelWs, SvoxLayer, SvoxPlan, Frame)


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
    sizes = (ctypes.c_int32 * len(_STRUCTS))()
    if lib.sherf_struct_sizes(sizes, len(_STRUCTS)) != 0 or [int(v) for v in sizes] != [ctypes.sizeof(c) for c in _STRUCTS]:
        raise RuntimeError(f'struct layout mismatch between include/sherf_hip.h and sherf_amd/_lib.py: '
                           f'{[int(v) for v in sizes]} vs {[ctypes.sizeof(c) for c in _STRUCTS]}')
    _lib = lib
    return lib


def lib():
    return _init()


def addr(t, dtype=None):
    """Like `ptr` but returns the integer address (what a c_void_p struct field takes)."""
    p = ptr(t, dtype)
    return None if p is None else p.value


def ptr(t, dtype=None, channels_last_ok=False):
    """Device pointer of a dense CUDA tensor (argument checks in the spirit of bias_act.cpp:39-55).  `channels_last_ok`: a
    4-D tensor dense in the channels_last format is accepted too (the operators that index by stride, bias_act.cpp:57)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f'expected a tensor, got {type(t)}')
    if not t.is_cuda:
        raise RuntimeError('sherf_amd: tensor is not on a GPU; the HIP path has no CPU fallback')
    if not t.is_contiguous() and not (channels_last_ok and t.ndim == 4 and t.is_contiguous(memory_format=torch.channels_last)):
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


# ---- the (experimental) backward library: include/sherf_hip_bwd.h -> libsherf_hip_bwd.so ----------------------------
HEADER_BWD = os.path.join(_HERE, '..', 'include', 'sherf_hip_bwd.h')
LIB_BWD_PATH = os.path.join(_HERE, 'libsherf_hip_bwd.so')
_lib_bwd = None


def lib_bwd():
    global _lib_bwd
    if _lib_bwd is None:
        if not os.path.exists(LIB_BWD_PATH):
            raise RuntimeError(f'{LIB_BWD_PATH} not found: build it with `python -m sherf_amd.build`')
        l = ctypes.CDLL(LIB_BWD_PATH)
        for name, (ret, args) in parse_header(HEADER_BWD).items():
            fn = getattr(l, name)
            fn.restype = ret
            fn.argtypes = [a[0] for a in args]
        _lib_bwd = l
    return _lib_bwd


def call_bwd(name, *args):
    l = lib_bwd()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {l.sherf_bwd_last_error().decode()}')


# ---- the producers' element-wise / FIR operators: include/sherf_hip_ops.h -> libsherf_hip_ops.so -----------------------------
HEADER_OPS = os.path.join(_HERE, '..', 'include', 'sherf_hip_ops.h')
LIB_OPS_PATH = os.path.join(_HERE, 'libsherf_hip_ops.so')
_lib_ops = None


def lib_ops():
    global _lib_ops
    if _lib_ops is None:
        if not os.path.exists(LIB_OPS_PATH):
            raise RuntimeError(f'{LIB_OPS_PATH} not found: build it with `python -m sherf_amd.build`. There is no CPU fallback.')
        l = ctypes.CDLL(LIB_OPS_PATH)
        for name, (ret, args) in parse_header(HEADER_OPS).items():
            fn = getattr(l, name)
            fn.restype = ret
            fn.argtypes = [a[0] for a in args]
        _lib_ops = l
    return _lib_ops


def call_ops(name, *args):
    l = lib_ops()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {l.sherf_ops_last_error().decode()}')
