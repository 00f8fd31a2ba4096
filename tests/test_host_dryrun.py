"""Host-side dry run of ImportanceRenderer.forward on CPU tensors with the native calls stubbed: exercises every line of the
Python that fills the frame descriptor (sherf_frame / sherf_svox_plan), in training, eval and density-noise mode, without a
GPU.  The kernels are not involved; this guards the host logic between GPU sessions."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, sherf_oracle as O
from sherf_amd import _lib
from tests import gpu_common as G


class _FakeCuda(torch.Tensor):
    is_cuda = True


@pytest.fixture()
def stubbed(monkeypatch):
    calls, frames = [], []
    def call(name, *a):
        if name == 'sherf_mlp_pack_stream':            # (the weight stream packed on the device: its flag word must read "all finite")
            ctypes.memset(a[8], 0, 4)
            return
        calls.append((name, a))
    monkeypatch.setattr(_lib, 'call', call)
    monkeypatch.setattr(_lib, 'addr', lambda t, dtype=None: None if t is None else t.data_ptr())
    monkeypatch.setattr(_lib, 'ptr', lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr()))
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda dev=None: type('S', (), {'cuda_stream': 0})())
    orig = _lib.Frame

    def make():
        frames.append(orig())
        return frames[-1]
    monkeypatch.setattr(_lib, 'Frame', make)
    return calls, frames


def _run(training, options=None):
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder
    from sherf_amd.voxel import SparseConvTensor
    fx = G.fixture('tiny')
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=G.smpl(), mlp_precision='f16x3')   # (one frame per call: no calibration)
    dec = NeRFDecoder(32)
    fixtures.load_seeded_state(rend, 'renderer.'); fixtures.load_seeded_state(dec, 'decoder.')
    rend.train(training); dec.train(training)
    rend._side = lambda dev, idx=0: type('X', (), {'cuda_stream': 8 + 8 * idx})()
    d = fixtures.to_torch(fx['input_data'])
    spi = O.render_from_fixture(fx, G.seeded_state(), keep=False)['sp_input']
    sp = SparseConvTensor(torch.from_numpy(fx['vertex_feat']), spi['coord'], spi['out_sh'], 1)
    spd = dict(coord=spi['coord'], out_sh=spi['out_sh'], batch_size=1, bounds=spi['bounds'][None])
    opts = dict(fx['options'])
    opts.update(options or {})
    with torch.no_grad():
        out = rend(torch.from_numpy(fx['planes']), d['obs_img_all'][:, 0], torch.from_numpy(fx['obs_feat']), sp, None, spd, dec,
                   d['ray_o_all'][:, 0].as_subclass(_FakeCuda), d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, opts)
    return rend, out


def test_training_frame_descriptor_is_complete(stubbed):
    calls, frames = stubbed
    rend, (rgb, depth, acc) = _run(True)
    # one native call enqueues the frame; a train-mode forward then advances the BatchNorm running statistics (one more launch)
    # the first frame on a workspace runs the sampler alone (phase 4) to size the token-side buffers from the frame's own count, then
    # one native call enqueues the frame
    assert [c[0] for c in calls] == ['sherf_render_frame', 'sherf_render_frame', 'sherf_svox_bn_running_update']
    assert [c[1][1] for c in calls[:2]] == [4, 3]
    assert rgb.shape == (1, 1024, 3) and depth.shape == (1, 1024, 1) and acc.shape == (1, 1024, 1)
    fr = frames[-1]
    for name, ctype in fr._fields_:
        if ctype is ctypes.c_void_p and name not in ('zfrag', 'pefrag'):   # (scratch of the opt-in two-launch network / encodings-in-gather forms only)
            assert getattr(fr, name), f'frame.{name} is NULL'
    assert (fr.R, fr.S, fr.capacity) == (1024, 16, 1024 * 16) and fr.vox_n == 6890 and fr.vox_training == 1
    assert (fr.P, fr.Hf, fr.Wf, fr.H, fr.W) == (32, 16, 16, 32, 32) and list(fr.vox_sh) == [int(v) for v in rend.last['bwd']['vox_sh']]
    assert fr.mlp_prec == 1 and fr.main_after_layer == -1
    assert {'plan', 'levels_struct', 'bwd', 'ws'} <= set(rend.last)


def test_eval_mode_prepares_batchnorm_constants_once(stubbed):
    calls, frames = stubbed
    rend, _ = _run(False)
    names = [c[0] for c in calls]
    assert names.count('sherf_svox_bn_finalize') == 13 and names[-1] == 'sherf_render_frame'     # running statistics -> bnparam
    assert frames[-1].vox_training == 0


def test_density_noise_splits_the_frame_in_two_phases(stubbed, monkeypatch):
    calls, frames = stubbed
    monkeypatch.setattr(torch, 'randn', lambda n, device=None: torch.zeros(n))
    _run(True, dict(density_noise=0.5))
    phases = [c[1][1] for c in calls if c[0] == 'sherf_render_frame']
    assert phases == [4, 1, 2]                 # (4: the first frame's sampler-only probe)


def test_backward_glue_dry_run(stubbed, monkeypatch):
    """render_backward's glue (contexts built from the forward's workspace and plan, ~120 staged calls) with every native call
    stubbed: buffer shapes / strides are checked by Mat's bounds assertions, names by the returned gradient keys."""
    from sherf_amd import backward, backward_dense
    calls, frames = stubbed
    bwd_calls = []
    monkeypatch.setattr(_lib, 'call_bwd', lambda name, *a: bwd_calls.append(name))
    monkeypatch.setattr(backward_dense.HipOps, '_p', staticmethod(lambda m: ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off)))
    rend, (rgb, depth, acc) = _run(True)
    rend.last['ws']['counters'][0] = 77                                  # pretend the sampler found 77 valid samples
    from sherf_amd.triplane import NeRFDecoder
    dec = NeRFDecoder(32)
    out = backward.render_backward(rend, dec, torch.zeros_like(rgb), torch.zeros_like(acc))
    names = {'renderer.' + k for k, _ in rend.named_parameters() if not k.startswith('encoder_3d.conv4') and not k.startswith('encoder_3d.down3')}
    names |= {'decoder.' + k for k, _ in dec.named_parameters()}
    assert set(out['params']) == names, set(out['params']) ^ names
    for k, p in list(rend.named_parameters()) + list(dec.named_parameters()):
        full = ('renderer.' if p is not None and k in dict(rend.named_parameters()) else 'decoder.') + k
        if full in out['params']:
            assert out['params'][full].shape == p.shape, full
    assert out['planes'].shape == (1, 3, 32, 32, 32) and out['obs_feat'].shape == (1, 64, 16, 16) and out['vertex_feat'].shape == (6890, 32)
    assert bwd_calls.count('sherf_bwd_gemm') + bwd_calls.count('sherf_bwd_gemm_bias_act') + bwd_calls.count('sherf_bwd_gemm_dgrad_fused') >= 50 and bwd_calls.count('sherf_bwd_relu_mask_colsum') == 1 and bwd_calls.count('sherf_bwd_gemm_dgrad_fused') == 8 and 'sherf_bwd_conv_wgrad' in bwd_calls and 'sherf_bwd_unfold32' in bwd_calls
    assert [c[0] for c in calls].count('sherf_gather_tokens_bwd_binned') == 1 and [c[0] for c in calls].count('sherf_composite_compact_bwd') == 1


def test_autograd_node_wiring(stubbed, monkeypatch):
    """enable_autograd: forward recorded as one node, loss.backward() routes through render_backward and lands a gradient of
    the right shape on every input and parameter the reference trains (native calls stubbed: values are zeros)."""
    from sherf_amd import backward_dense
    from sherf_amd.renderer import ImportanceRenderer
    from sherf_amd.triplane import NeRFDecoder
    from sherf_amd.voxel import SparseConvTensor
    monkeypatch.setattr(_lib, 'call_bwd', lambda name, *a: None)
    monkeypatch.setattr(backward_dense.HipOps, '_p', staticmethod(lambda m: ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off)))
    fx = G.fixture('tiny')
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=G.smpl())
    dec = NeRFDecoder(32)
    rend.train(); dec.train()
    rend.enable_autograd = True
    rend._side = lambda dev, idx=0: type('X', (), {'cuda_stream': 8 + 8 * idx})()
    d = fixtures.to_torch(fx['input_data'])
    spi = O.render_from_fixture(fx, G.seeded_state(), keep=False)['sp_input']
    planes = torch.from_numpy(fx['planes']).requires_grad_(True)
    obs_feat = torch.from_numpy(fx['obs_feat']).requires_grad_(True)
    vfeat = torch.from_numpy(fx['vertex_feat']).requires_grad_(True)
    sp = SparseConvTensor(vfeat, spi['coord'], spi['out_sh'], 1)
    spd = dict(coord=spi['coord'], out_sh=spi['out_sh'], batch_size=1, bounds=spi['bounds'][None])
    calls, _ = stubbed
    del calls[:]
    rgb, depth, acc = rend(planes, d['obs_img_all'][:, 0], torch.from_numpy(fx['obs_feat']) * 0 + obs_feat, sp, None, spd, dec,
                           d['ray_o_all'][:, 0].as_subclass(_FakeCuda), d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d,
                           dict(fx['options']))
    assert rgb.requires_grad and acc.requires_grad and not depth.requires_grad
    assert 'sherf_svox_bn_running_update' in [c[0] for c in calls]                      # the running statistics are advanced (natively)
    rend.last['ws']['counters'][0] = 50
    (rgb.sum() + acc.sum()).backward()
    assert planes.grad.shape == planes.shape and obs_feat.grad.shape == obs_feat.shape and vfeat.grad.shape == vfeat.shape
    trained = [n for n, p in list(rend.named_parameters()) + list(dec.named_parameters()) if p.grad is not None]
    assert len(trained) == 78                     # + the 3 inputs above = the 81 gradient tensors of the reference


def test_full_size_property_test_plumbing(monkeypatch):
    """The full-size GPU test (tests/test_gpu_parity.py: _full_size_properties) executed on the CPU with the HIP render replaced
    by the oracle, on a small configuration: checks the test's own plumbing (subset selection, shapes, comparisons)."""
    from tests import test_gpu_parity as T

    def fake_render(cfg, precision='f16x3', training=True, fx=None, sp_input=None, options=None):
        f = dict(fx or G.fixture(cfg))
        opts = dict(f['options']); opts.update(options or {})
        f['options'] = dict(opts, margins=True)
        r = O.render_from_fixture(f, G.state_for(cfg), training=training, keep=False)
        nv = r['valid'].numel()
        ws = dict(counters=torch.tensor([nv, 0, 0, 0]), cs_idx=r['valid'].int(), cs_vid=r['vert_id'].int(), cs_tvid=r['t_vert_id'].int(),
                  sample_out=torch.cat([r['sample_rgb'], r['sample_sigma'].view(-1, 1)], 1))
        return dict(rgb=r['rgb'], depth=r['depth'], acc=r['acc'], last=dict(ws=ws, mlp_precision=precision, table_precision='f32' if precision == 'f16x3' else 'f16', encoder_precision=(options or {}).get('encoder_precision', 'f16x3')), rend=None)
    monkeypatch.setattr(T.G, 'hip_render', fake_render)
    T._full_size_properties('tiny', 7)                                   # subset mode (host build of the kernels)
    T._full_size_properties('tiny_ri', 7, check_stride=5, device='cpu')  # whole-frame mode: the oracle "on the device" + its cross-check
    T._full_size_properties('tiny', 7, check_stride=5, device='cpu')


# ---- bench.py: roofline.traffic from rocprofv3 --pmc child passes ------------------------------------------------------------------
def test_bench_pmc_traffic_parses_counter_csv_and_reports_failures(monkeypatch, tmp_path):
    """No GPU / no profiler here: the two `rocprofv3 --pmc` child passes are faked -- their counter_collection.csv is parsed into HBM
    bytes per nerf_mlp_kernel launch (FETCH_SIZE doubled, the gfx950 rule), and a failing pass yields {'error': ...}, never an exception."""
    import argparse
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    a = argparse.Namespace(config='cfg2', precision='f16x3', bn_mode='train')
    seen = []

    def fake_run(cmd, env=None, cwd=None, **kw):
        seen.append((cmd, env))
        counter, outdir = cmd[cmd.index('--pmc') + 1], cmd[cmd.index('-d') + 1]
        os.makedirs(os.path.join(outdir, 'host', '123'), exist_ok=True)
        with open(os.path.join(outdir, 'host', '123', '123_counter_collection.csv'), 'w') as f:
            f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
            for i, v in enumerate((1000.0, 3000.0)):
                f.write(f'{i},"void (anonymous namespace)::nerf_mlp_kernel<1>(int const*)","{counter}",{v if counter == "FETCH_SIZE" else v / 10}\n')
            f.write(f'9,"gather_tokens_kernel","{counter}",777\n')
        return subprocess.CompletedProcess(cmd, 0, stdout='', stderr='')
    monkeypatch.setattr(bench.shutil if hasattr(bench, 'shutil') else __import__('shutil'), 'which', lambda n: '/bin/true')
    monkeypatch.setenv('RANK', '3'); monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    r = bench.pmc_traffic(a, 3)
    assert r['hbm_bytes_per_launch'] == int(2 * 2000.0 * 1024 + 200.0 * 1024) and r['dispatches'] == 2
    assert all('--pmc-child' in c and e['LOCAL_RANK'] == '3' and 'RANK' not in e and e['TMPDIR'] == '/tmp' for c, e in seen)
    assert [c[c.index('--pmc') + 1] for c, _ in seen] == ['FETCH_SIZE', 'WRITE_SIZE']                       # separate passes
    assert not any(x in c for c, _ in seen for x in ('--sys-trace', '--kernel-trace', '--hip-trace', '--stats'))   # counters alone

    def boom(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, 1)
    monkeypatch.setattr(subprocess, 'run', boom)
    assert 'error' in bench.pmc_traffic(a, 0)


def test_workspaces_are_bounded_per_renderer():
    """a caller that creates fresh streams must not leak one multi-gigabyte workspace per stream: the least recently used
    goes once MAX_WORKSPACES exist; a stream that keeps rendering keeps its own."""
    from sherf_amd.renderer import ImportanceRenderer
    rend = ImportanceRenderer(True, True, True, use_trans=True, use_NeRF_decoder=True, smpl=G.smpl())
    dev = torch.device('cpu')
    S = lambda sid: type('X', (), {'cuda_stream': sid})()
    first = rend._workspace(dev, main=S(100))
    for sid in range(101, 101 + rend.MAX_WORKSPACES - 1):
        rend._workspace(dev, main=S(sid))
    assert rend._workspace(dev, main=S(100)) is first and len(rend._ws) == rend.MAX_WORKSPACES      # touched: most recent again
    rend._workspace(dev, main=S(999))                                                                # evicts stream 101's, not 100's
    assert len(rend._ws) == rend.MAX_WORKSPACES and rend._workspace(dev, main=S(100)) is first
    assert (str(dev), 101) not in rend._ws



def test_bench_child_command_line(monkeypatch):
    """bench.py's child runs (one frame in flight for the roofline / timeline / secondary lines; the wide framing on N streams): flags given
    to the child override the ones inherited from the parent's run, everything that must not run twice is switched off."""
    import argparse
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, **kw):
        seen['cmd'] = cmd
        return type('R', (), dict(stdout='noise\n{"value": 1.5, "ms_per_step": 2.0}\n', stderr='', returncode=0))()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    a = argparse.Namespace(config='cfg2_dense_ri', precision='auto', bn_mode='train', table_precision=None, encoder_precision='f16')
    out = bench.bench_child(a, 0, ['--streams', '1', '--steps', '12', '--config', 'cfg2_ri', '--no-secondary'])
    assert out == dict(value=1.5, ms_per_step=2.0)
    cmd = seen['cmd'][2:]
    assert cmd.count('--config') == 1 and cmd[cmd.index('--config') + 1] == 'cfg2_ri'            # the child's own value wins
    assert cmd[cmd.index('--streams') + 1] == '1' and cmd[cmd.index('--encoder-precision') + 1] == 'f16' and '--table-precision' not in cmd
    for flag in ('--no-cpu-baseline', '--no-torch-gpu-baseline', '--no-pmc', '--no-train', '--no-secondary'):
        assert flag in cmd
    monkeypatch.setattr(subprocess, 'run', lambda cmd, **kw: type('R', (), dict(stdout='', stderr='boom', returncode=3))())
    assert 'error' in bench.bench_child(a, 0, ['--streams', '1'])
