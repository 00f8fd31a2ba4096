"""Functional check of a kernel VARIANT on the CPU: builds the forward library for the host (tests/hipcpu) with the given -D defines and
renders tiny / tiny_nv through the whole product path against the oracle.   python tools/cpu_variant_check.py SHERF_MLP_FAST_ERF=1"""
import sys, ctypes, time, torch, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'hipcpu'))
import build_cpu
from tests import gpu_common as G
from tests import test_hipcpu_frame as T
defs=sys.argv[1].split(',') if len(sys.argv)>1 and sys.argv[1] else []
path=build_cpu.build('sherf_hipcpu_full', T.FWD_SOURCES, '/tmp/hipcpu_var_'+'_'.join(d.replace('=','') for d in defs), compiler=build_cpu.CLANG, defines=defs)
from sherf_amd import _lib
_lib.LIB_PATH=path; _lib._lib=None
_lib.ptr = lambda t, dtype=None: None if t is None else ctypes.c_void_p(t.data_ptr())
_lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
torch.cuda.synchronize = lambda dev=None: None
G.CPU_SHIM=True
for cfg in ('tiny','tiny_nv'):
    o=G.oracle_render(cfg); h=G.hip_render(cfg)
    nv=o['valid'].numel(); out=h['last']['ws']['sample_out'][:nv]
    sr=torch.relu(o['sample_sigma'])
    print(cfg, defs, 'sigma+ %.2e rgb %.2e img %.2e'%(float((torch.relu(out[:,3])-sr).abs().max()/sr.max()), float((out[:,:3]-o['sample_rgb']).abs().max()), G.rel(h['rgb'],o['rgb'])))
