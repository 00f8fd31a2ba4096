"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares,
and the product modules expose exactly the reference's parameter / buffer names and shapes (the checkpoint
contract of training_loop.py:207-208, dumped from the unmodified reference by oracle/make_golden.py)."""
import ctypes
import json
import os
import pickle

import pytest
import torch

from sherf_amd import _lib


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 32
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), name
    l = _lib.lib()
    assert l.sherf_version() >= 100


def test_bad_arguments_return_error_codes_not_crashes():
    l = _lib.lib()
    assert l.sherf_smpl_bones(None, None, 3, None, None, None, None, None, None) == -1
    assert b'bad argument' in l.sherf_last_error()
    with pytest.raises(RuntimeError):
        _lib.call('sherf_svox_scan', None, 0, None, None, None, None, None)
    # the native drivers validate before touching the device
    assert l.sherf_render_frame(None, 3, None, None, None, None) == -1
    assert l.sherf_svox_encode(None, None, None, 0, 1, None, None) == -1
    fr = _lib.Frame()
    lv = (_lib.VoxLevel * 3)()
    assert l.sherf_render_frame(ctypes.byref(fr), 0, lv, None, ctypes.c_void_p(8), None) == -1      # phase 0
    assert l.sherf_render_frame(ctypes.byref(fr), 3, lv, None, None, None) == -1                    # side == main stream
    n = ctypes.c_int32(0)
    assert l.sherf_profile_frames_read(None, 4, ctypes.byref(n)) == -1


def test_cpu_tensors_are_rejected_loudly():
    with pytest.raises(RuntimeError, match='not on a GPU'):
        _lib.ptr(torch.zeros(4))
    from sherf_amd.ray_sampler import RaySampler
    with pytest.raises(RuntimeError, match='GPU only'):
        RaySampler()(torch.eye(4)[None], torch.eye(3)[None], 4)


def test_parameter_names_match_reference(golden_dir):
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder
    ref = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl={})
    dec = NeRFDecoder(32)
    ours = {'renderer.' + k: list(v.shape) for k, v in rend.state_dict().items()}
    ours.update({'decoder.' + k: list(v.shape) for k, v in dec.state_dict().items()})
    assert set(ours) == set(ref), (sorted(set(ref) - set(ours))[:5], sorted(set(ours) - set(ref))[:5])
    for k in ref:
        assert ours[k] == ref[k], (k, ours[k], ref[k])


def test_renderer_is_picklable_like_reference_snapshots():
    from sherf_amd.renderer import ImportanceRenderer
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl={})
    r2 = pickle.loads(pickle.dumps(rend))               # training_loop.py:566-579 pickles the whole generator
    assert set(r2.state_dict()) == set(rend.state_dict())


def test_mlp_stream_layout_matches_packer():
    """The step table of the weight stream: the library's own export (csrc/mlp.hip, constexpr -- what the kernel walks) against its
    restatement in the packer, for both precisions."""
    from sherf_amd import mlp_pack
    for prec in (0, 1, 2):
        n = ctypes.c_int32(0)
        pieces = (ctypes.c_int32 * 64)()
        units = (ctypes.c_int32 * 640)()
        assert _lib.lib().sherf_mlp_stream_layout(prec, ctypes.byref(n), pieces, units, 64) == 0
        assert n.value == mlp_pack.N_STEPS
        for s in range(n.value):
            assert pieces[s] == mlp_pack.step_pieces(s, prec), s
            for u in range(10):
                want = mlp_pack.step_unit(s, u) if u < mlp_pack.step_units(s) else None
                assert units[s * 10 + u] == (-1 if want is None else want[0] * 16 + want[1]), (s, u)
    assert _lib.lib().sherf_mlp_stream_layout(3, ctypes.byref(n), pieces, units, 64) != 0


def test_struct_layouts_match_header(tmp_path):
    """Every field of the ctypes mirrors in sherf_amd/_lib.py sits at the offset the C header gives it."""
    import ctypes
    import subprocess
    from sherf_amd import _lib
    names = {'VoxLevel': 'sherf_vox_level', 'SvoxLevelWs': 'sherf_svox_level_ws', 'SvoxLayer': 'sherf_svox_layer',
             'SvoxPlan': 'sherf_svox_plan', 'Frame': 'sherf_frame'}
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){\n' % os.path.abspath(_lib.HEADER)
    want = []
    for c in _lib._STRUCTS:
        for f in c._fields_:
            src += 'printf("%%zu\\n", offsetof(%s, %s));\n' % (names[c.__name__], f[0])
            want.append((c.__name__, f[0], getattr(c, f[0]).offset))
        src += 'printf("%%zu\\n", sizeof(%s));\n' % names[c.__name__]
        want.append((c.__name__, 'sizeof', ctypes.sizeof(c)))
    src += 'return 0;}\n'
    (tmp_path / 't.c').write_text(src)
    subprocess.check_call(['gcc', str(tmp_path / 't.c'), '-o', str(tmp_path / 't')])
    got = [int(v) for v in subprocess.check_output([str(tmp_path / 't')]).decode().split()]
    assert [w[2] for w in want] == got, [(w, g) for w, g in zip(want, got) if w[2] != g]


def test_backward_library_exports_every_declared_symbol():
    """libsherf_hip_bwd.so (include/sherf_hip_bwd.h, experimental backward building blocks) loads next to torch's rocBLAS and
    validates its arguments before touching a device."""
    protos = _lib.parse_header(_lib.HEADER_BWD)
    assert len(protos) >= 23
    l = _lib.lib_bwd()
    for name in protos:
        assert hasattr(l, name), name
    assert l.sherf_bwd_gemm(0, 0, 0, 0, 0, None, 0, None, 0, None, 0, ctypes.c_float(0.0), None) == -1
    assert b'bad argument' in l.sherf_bwd_last_error()
    assert l.sherf_bwd_conv_dgrad(None, None, 1, 1, 1, None, 1, 1, 1, None, 32, None, 32, 0, 1, None, None) == -1


def test_ops_library_exports_every_declared_symbol():
    """libsherf_hip_ops.so (include/sherf_hip_ops.h: bias_act, upfirdn2d) loads and validates its arguments before touching a device."""
    protos = _lib.parse_header(_lib.HEADER_OPS)
    assert set(protos) == {'sherf_ops_last_error', 'sherf_bias_act', 'sherf_upfirdn2d'}
    l = _lib.lib_ops()
    for name in protos:
        assert hasattr(l, name), name
    assert l.sherf_bias_act(None, None, None, None, None, None, 4, 1, 1, 0, 3, ctypes.c_float(0.2), ctypes.c_float(1.0), ctypes.c_float(-1.0), 0, None) == -1
    assert b'bad argument' in l.sherf_ops_last_error()
    assert l.sherf_upfirdn2d(None, None, None, 1, 1, 4, 4, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, ctypes.c_float(1.0), 0, None) == -1


def test_sparse_encoder_plan_host_logic(monkeypatch):
    """SparseConvNet.plan (the descriptor of the native encoder driver) built on CPU tensors: layer table, level capacities,
    accumulator placement -- the host logic of sherf_svox_encode, no device involved."""
    from sherf_amd import voxel
    from sherf_amd.renderer import _Workspace
    monkeypatch.setattr(_lib, 'addr', lambda t, dtype=None: None if t is None else t.data_ptr())
    net = voxel.SparseConvNet()
    pl = net.plan([96, 320, 384], 6890, [torch.zeros(4) for _ in range(3)], _Workspace(), torch.device('cpu'))
    p = pl['plan']
    assert p.n_layers == 13
    lay = [(p.layers[i].cin, p.layers[i].cout, p.layers[i].down, p.layers[i].tap) for i in range(13)]
    assert lay == [(32, 32, 0, 0), (32, 32, 0, 0), (32, 32, 1, 0), (32, 32, 0, 0), (32, 32, 0, 1), (32, 64, 1, 0), (64, 64, 0, 0), (64, 64, 0, 0),
                   (64, 64, 0, 1), (64, 96, 1, 0), (96, 96, 0, 0), (96, 96, 0, 0), (96, 96, 0, 1)]
    assert [(p.lev[i].D, p.lev[i].H, p.lev[i].W) for i in range(4)] == [(96, 320, 384), (48, 160, 192), (24, 80, 96), (12, 40, 48)]
    assert [p.lev[i].cap for i in range(4)] == [6890, 55120, 55120, 23040]
    z0, z1 = p.zero_ptr, p.zero_ptr + p.zero_bytes                       # everything that must be zero at frame start is inside
    for i in range(4):
        assert z0 <= p.lev[i].bitmap < z1
    for i in range(13):
        assert z0 <= p.layers[i].acc and p.layers[i].acc + 8 * 2 * p.layers[i].cout * 8 <= z1 and p.layers[i].acc % 8 == 0
    assert z0 <= p.acc_fix < z1 and z0 <= p.mult < z1
    assert [m['wname'] for m in pl['meta']][:3] == ['conv0.0', 'conv0.3', 'down0.0'] and [t[0] for t in pl['taps']] == [1, 2, 3]
