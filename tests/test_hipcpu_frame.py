"""The WHOLE product path on the CPU: sherf_amd's Python host code + the native frame driver (csrc/frame.hip) + every kernel
source of libsherf_hip.so / libsherf_hip_bwd.so, compiled unchanged for the host against the HIP-on-CPU shim (tests/hipcpu) and
driven through the same ctypes binding as on the MI355X.  The tests are the `-m gpu` parity tests themselves
(tests/test_gpu_parity.py, switched to CPU tensors by tests.gpu_common.CPU_SHIM) plus the BASELINE config-5 training step through
the autograd node.  They check the kernels' arithmetic and indexing between GPU sessions -- not their timing, and nothing
gfx950-specific (LDS-DMA, wave scheduling, code generation)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import sherf_oracle as O
from synthdata import fixtures
from sherf_amd import _lib
from tests import gpu_common as G
from tests import test_gpu_parity as P
from tests.hipcpu import build_cpu

from sherf_amd.build import SOURCES as FWD_SOURCES   # every source of libsherf_hip.so


@pytest.fixture(scope='module')
def cpu_product(tmp_path_factory):
    """sherf_amd._lib pointed at host builds of both libraries; tests.gpu_common switched to CPU tensors (G.CPU_SHIM)."""
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the bf16 kernels')
    fwd = build_cpu.build('sherf_hipcpu_full', FWD_SOURCES, str(tmp_path_factory.mktemp('hipcpu_full')), compiler=build_cpu.CLANG)
    bwd = build_cpu.build('sherf_hipcpu_bwd', ['bwd_dense.hip', 'bwd_gemm.hip', 'bwd_encoder.hip'], str(tmp_path_factory.mktemp('hipcpu_bwd')), compiler=build_cpu.CLANG)
    ops = build_cpu.build('sherf_hipcpu_ops', ['ops_lib.hip', 'ops_bias_act.hip', 'ops_upfirdn2d.hip'], str(tmp_path_factory.mktemp('hipcpu_ops')),
                          compiler=build_cpu.CLANG)
    from sherf_amd import backward_dense
    mp = pytest.MonkeyPatch()
    mp.setattr(_lib, 'LIB_PATH', fwd); mp.setattr(_lib, '_lib', None)
    mp.setattr(_lib, 'LIB_BWD_PATH', bwd); mp.setattr(_lib, '_lib_bwd', None)
    mp.setattr(_lib, 'LIB_OPS_PATH', ops); mp.setattr(_lib, '_lib_ops', None)
    mp.setattr(_lib, 'ptr', lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr()))
    mp.setattr(_lib, 'addr', lambda t, dtype=None: None if t is None else t.data_ptr())
    mp.setattr(_lib, 'stream', lambda: ctypes.c_void_p(0))
    mp.setattr(torch.cuda, 'current_stream', lambda dev=None: type('S', (), {'cuda_stream': 0})())
    mp.setattr(torch.cuda, 'synchronize', lambda dev=None: None)
    mp.setattr(backward_dense.HipOps, '_p', staticmethod(lambda m: ctypes.c_void_p(m.buf.data_ptr() + 4 * m.off)))
    from sherf_amd.renderer import ImportanceRenderer
    mp.setattr(ImportanceRenderer, 'SMPL_NEUTRAL', property(lambda self: self._smpl(torch.device('cpu'))))
    mp.setattr(G, 'CPU_SHIM', True)
    G.hip_modules.cache_clear()
    yield
    G.hip_modules.cache_clear()
    mp.undo()


# ---- the `-m gpu` parity tests themselves (tests/test_gpu_parity.py), run against the host build ----------------------------
@pytest.fixture(scope='module', params=P.CFGS)
def run(request, cpu_product):
    cfg = request.param
    return cfg, G.oracle_render(cfg), G.hip_render(cfg)


def test_stage_by_stage_parity(run):
    """shell mask / vertex ids / compact order bit exact, warp, voxel levels, gathered tokens -- and the frame against the oracle
    and the unmodified reference's golden image."""
    P.test_mask_and_nearest_vertex_bit_exact(run)
    P.test_warp_matches_literal_lbs_chain(run)
    P.test_sparse_voxel_encoder_levels(run)
    P.test_gathered_tokens(run)
    cfg, o, h = run
    nv = o['valid'].numel()
    out = h['last']['ws']['sample_out'][:nv]
    sig_ref = torch.relu(o['sample_sigma'])
    assert float((torch.relu(out[:, 3]) - sig_ref).abs().max() / sig_ref.max()) < 1e-3
    assert float((out[:, :3] - o['sample_rgb']).abs().max()) < 1e-3
    g = np.load(os.path.join(G.GOLDEN, f'renderer_{cfg}.npz'))                     # outputs of the UNMODIFIED reference
    assert G.rel(h['rgb'], o['rgb']) < 1e-3 and G.rel(h['acc'], o['acc']) < 1e-3
    assert torch.allclose(h['depth'], o['depth'], rtol=1e-3, atol=1e-4)
    assert G.rel(h['rgb'], torch.from_numpy(g['rgb'])) < 1e-3 and G.rel(h['acc'], torch.from_numpy(g['acc'][:, 0])) < 1e-3
    assert O.psnr(h['rgb'], torch.from_numpy(g['rgb'])) > 60.0
    P.test_margin_protocol_whole_frame(cfg)                                          # flips / per-sample relative error / explained rays


def test_frame_launch_switches_render_the_same_bits(cpu_product):
    """Grids sized by the frame's own sample count (SHERF_FRAME_EXACT_GRIDS) and the gather's schedule variants: same arithmetic."""
    h = G.hip_render('tiny_nv')
    for opts in (dict(exact_grids=True), dict(gather_branchless=True), dict(gather_branchless='128'), dict(near_lists=False)):
        b = G.hip_render('tiny_nv', options=opts)
        assert torch.equal(b['rgb'], h['rgb']) and torch.equal(b['acc'], h['acc']) and torch.equal(b['depth'], h['depth']), opts
    # the single-product bf16 mode (north_star's nominal precision) runs through the same frame, at its own (looser) accuracy
    b = G.hip_render('tiny_nv', precision='bf16')
    assert 1e-4 < G.rel(b['rgb'], h['rgb']) < 0.2
    # the per-sample network as two launches (sherf_nerf_mlp_split: tokens kernel + decoder kernel) == the one-launch kernel, in
    # every precision (the default is the one-launch kernel: measured faster on the MI355X)
    for prec in ('f16x3', 'f16', 'bf16'):
        one = G.hip_render('tiny_nv', precision=prec, options=dict(mlp_split=False))
        two = G.hip_render('tiny_nv', precision=prec, options=dict(mlp_split=True))
        dflt = G.hip_render('tiny_nv', precision=prec)
        for k in ('rgb', 'acc', 'depth'):
            assert torch.equal(one[k], two[k]) and torch.equal(one[k], dflt[k]), (prec, k)
    # round 5's launch forms of the single-product network (epilogues inside the MFMA stream: the default; two tiles per wave) == the one-tile kernel
    for prec in ('f16', 'bf16'):
        one = G.hip_render('tiny_nv', precision=prec, options=dict(mlp_form='one'))
        for form in ('pipelined', 'two_tiles'):
            b = G.hip_render('tiny_nv', precision=prec, options=dict(mlp_form=form))
            assert b['last']['mlp_form'] == form
            for k in ('rgb', 'acc', 'depth'):
                assert torch.equal(one[k], b[k]), (prec, form, k)
    # round 6: the positional encodings evaluated by the gather and handed to the pipelined fp16 network as operand fragments (SHERF_FRAME_PE_FRAGS:
    # opt-in -- measured slower on the MI355X) == the network evaluating them itself; other configurations ignore the option
    on = G.hip_render('tiny_nv', precision='f16', options=dict(pe_in_gather=True))
    off = G.hip_render('tiny_nv', precision='f16', options=dict(pe_in_gather=False))
    assert on['last']['pe_in_gather'] and not off['last']['pe_in_gather'] and not G.hip_render('tiny_nv', precision='f16')['last']['pe_in_gather']
    pf = on['last']['ws']['pefrag']
    assert pf is not None and int((pf != 0).sum()) > 1000
    for k in ('rgb', 'acc', 'depth'):
        assert torch.equal(on[k], off[k]), k
    assert torch.equal(on['last']['ws']['sample_out'], off['last']['ws']['sample_out'])
    for opts in (dict(mlp_form='one'), dict(mlp_parts=2), dict(table_precision='f32'), dict(mlp_split=True), dict(gather_split=True)):
        assert not G.hip_render('tiny_nv', precision='f16', options=dict(pe_in_gather=True, **opts))['last']['pe_in_gather'], opts
    assert not G.hip_render('tiny_nv', precision='bf16', options=dict(pe_in_gather=True))['last']['pe_in_gather']
    # use_trans = False (round 5): every launch form of the network against the golden of the unmodified reference built without its transformer
    check_without_transformer()
    # the feature-branch switches (round 6): goldens of the unmodified reference built with each combination
    check_feature_branch_switches()
    # gather + network cut into parts on two streams (sherf_nerf_mlp_part): a schedule, not an arithmetic, variant
    for parts in (2, 3, 8):
        b = G.hip_render('tiny_nv', options=dict(mlp_parts=parts))
        assert b['last']['mlp_parts'] == parts
        assert torch.equal(b['rgb'], h['rgb']) and torch.equal(b['acc'], h['acc']) and torch.equal(b['depth'], h['depth']), parts
    # round 5's launch-order / stream-placement experiments of the first phase (SHERF_EXPERIMENT, csrc/common.h: level-0 rows scattered before the level
    # builds are queued, level builds on the encoder's own stream, encoder queued before the ray side; bit 4 = host-clock stamps on stderr): order only
    import os
    try:
        for word in (1, 2, 3, 8, 9, 16, 24576, 49152, 1 << 20):     # (24576 / 49152, round 6: sixteen-lane compaction + the deeper list search)
            os.environ['SHERF_EXPERIMENT'] = str(word)
            b = G.hip_render('tiny_nv')
            assert torch.equal(b['rgb'], h['rgb']) and torch.equal(b['acc'], h['acc']) and torch.equal(b['depth'], h['depth']), word
        # round 6, bit 12: the two 96-column single-product sparse convolutions as three 32-column workgroups per row tile (svox.hip, sconv3_kernel: CS)
        single = dict(encoder_precision='f16')
        whole = G.hip_render('tiny_nv', precision='f16', options=single)
        os.environ['SHERF_EXPERIMENT'] = '4096'
        cs = G.hip_render('tiny_nv', precision='f16', options=single)
        assert cs['last']['encoder_precision'] == 'f16'
        for k in ('rgb', 'acc', 'depth'):
            assert torch.equal(cs[k], whole[k]), k
        assert torch.equal(cs['last']['ws']['sample_out'], whole['last']['ws']['sample_out'])
    finally:
        os.environ.pop('SHERF_EXPERIMENT', None)


def test_token_workspace_is_sized_from_the_frame(cpu_product):
    check_token_workspace()


def check_token_workspace():
    """geom / tokens / extras / sample_out (480 of the workspace's ~510 bytes per sample) are sized from the frame's
    own number of valid samples, not R * S: the first frame runs the sampler alone to learn it, later frames grow it from the counts of
    finished frames; a frame that does not fit renders its rays as NaN and raises the flag instead of compositing memory nobody wrote.
    Shared by the host-build test above and tests/test_gpu_parity.py (pinned count read-back, events: the real HIP runtime)."""
    G.hip_modules.cache_clear()
    rend, dec = G.hip_modules()
    rend.TOKEN_GRANULE = 16
    try:
        worst = G.hip_render('tiny_nv', options=dict(token_capacity='worst'))
        nv = int(worst['last']['ws']['counters'][0])
        R, S = worst['last']['R'], worst['last']['S']
        assert worst['last']['cap'] == R * S and worst['last']['sampler_cap'] == R * S and nv > 100
        wsp = rend._workspace(torch.device('cpu') if G.CPU_SHIM else torch.device('cuda', torch.cuda.current_device()))
        tok_bytes = lambda: sum(wsp.t[k].numel() * 4 for k in ('geom', 'tokens', 'extras', 'sample_out'))
        big, big_tok = wsp.nbytes(), tok_bytes()
        assert big_tok > 480 * R * S
        wsp.nv_sized_for = None                                  # as on a fresh workspace
        auto = G.hip_render('tiny_nv')                          # default: 'auto'
        assert auto['last']['cap'] == (int(1.5 * nv) + 15) // 16 * 16 < R * S // 4 and auto['last']['sampler_cap'] == R * S
        assert tok_bytes() < 0.06 * big_tok and big - wsp.nbytes() > 0.9 * big_tok          # (the rest of the workspace does not scale with R * S * 480 B)
        for k in ('rgb', 'depth', 'acc'):
            assert torch.equal(auto[k], worst[k]), k
        assert rend.check_finite() and not rend.poll_flags(wait=True).get('overflowed')
        # too small on purpose: the rays whose samples lie beyond the capacity are NaN, the others are the same bits, the flag is up
        small = G.hip_render('tiny_nv', options=dict(token_capacity=nv // 2))
        cap = small['last']['cap']
        assert cap < nv
        base, cnt = small['last']['ws']['ray_base'], small['last']['ws']['ray_cnt']
        lost = G.plain((base.long() + cnt.long()) > cap)
        assert 0 < int(lost.sum()) < R
        assert torch.isnan(small['rgb'][lost]).all() and torch.isnan(small['acc'][lost]).all() and torch.isnan(small['depth'][lost]).all()
        assert torch.equal(small['rgb'][~lost], worst['rgb'][~lost]) and torch.equal(small['acc'][~lost], worst['acc'][~lost])
        assert not rend.check_finite() and rend.poll_flags(wait=True)['overflowed'] >= 1
        # back on 'auto': the count the watch has seen (nv) exceeds 80 % of the capacity -> the next frame re-sizes before rendering
        again = G.hip_render('tiny_nv')
        assert again['last']['cap'] == (int(1.5 * nv) + 15) // 16 * 16 and rend._flags.get('token_regrowths', 0) >= 1
        for k in ('rgb', 'depth', 'acc'):
            assert torch.equal(again[k], worst[k]), k
        assert rend.check_finite()
        # other inputs under the same workspace (new ray / vertex tensors): the frame's count is read right behind its sampler and the frame
        # rendered again if it did not fit -- first a frame with few valid samples (rays cut short), then the full one, 3 x its size
        import copy
        fx_short = copy.deepcopy(dict(G.fixture('tiny_nv')))
        d = fx_short['input_data']
        d['far_all'] = (d['near_all'] + 0.3 * (d['far_all'] - d['near_all'])).astype(d['far_all'].dtype)
        wsp.nv_sized_for = None
        short = G.hip_render('tiny_nv', fx=fx_short)
        nv_short = int(short['last']['ws']['counters'][0])
        assert 0 < nv_short < nv / 1.6 and short['last']['cap'] < nv
        before = rend._flags.get('token_rerenders', 0)
        full = G.hip_render('tiny_nv')
        assert rend._flags.get('token_rerenders', 0) == before + 1 and full['last']['cap'] >= nv
        for k in ('rgb', 'depth', 'acc'):
            assert torch.equal(full[k], worst[k]), k
        assert rend.check_finite()
    finally:
        del rend.TOKEN_GRANULE
        G.hip_modules.cache_clear()


def test_bench_main_dry_run(cpu_product, monkeypatch, capsys):
    """bench.py's own plumbing (workload construction, the timed loop, the JSON line with the roofline object, the secondary lines, the
    stock-ops baseline + margin-protocol parity) executed on the host build -- the numbers mean nothing here, the point is that the
    script the driver runs on the MI355X has been run."""
    import json
    import sys
    import bench
    from sherf_amd.renderer import ImportanceRenderer
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    monkeypatch.setattr(bench, '_device', lambda lrank: torch.device('cpu'))
    monkeypatch.setattr(ImportanceRenderer, '_side', lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})())

    class HostEvent:                                     # torch.cuda.Event stand-in for mlp_kernel_alone
        def __init__(self, enable_timing=False): self.t = 0.0
        def record(self, stream=None):
            import time
            self.t = time.perf_counter()
        def elapsed_time(self, other): return 1e3 * (other.t - self.t)
    monkeypatch.setattr(torch.cuda, 'Event', HostEvent)
    monkeypatch.setattr(torch.cuda, 'empty_cache', lambda: None)
    # the stock-ops baseline's child is run in-process here (no GPU for a real child): its timing entry and the oracle frame it saves
    # feed the `parity` entry -- BASELINE's "PSNR vs ref" for the very frame that was timed
    monkeypatch.setattr(bench, 'torch_gpu_baseline_child',
                        lambda a, lrank, timeout=300, save=None: bench.torch_gpu_baseline(a.config, torch.device('cpu'), a.bn_mode == 'train', iters=1, save=save))
    monkeypatch.setattr(bench, 'pmc_traffic', lambda a, lrank, timeout=150, child=None, keys=None: dict(hbm_bytes_per_launch=12345, note='faked'))
    monkeypatch.setattr(bench, 'SECONDARY_ITERS', dict(mlp=1, mlp_warmup=0, frames=1, frames_warmup=0))
    keep = {k: fixtures.CONFIGS[k] for k in ('cfg3', 'cfg2_dense', 'cfg3_ri', 'cfg2_dense_ri', 'cfg2', 'cfg2_ri')}
    for sfx, var in (('', {}), ('_ri', dict(variant='ri'))):                       # the secondary workloads, tiny-sized here
        fixtures.CONFIGS['cfg3' + sfx] = dict(fixtures.CONFIGS['tiny'], **var)
        fixtures.CONFIGS['cfg2_dense' + sfx] = dict(fixtures.CONFIGS['tiny'], fill=1.55, **var)
    fixtures.CONFIGS['cfg2'] = dict(fixtures.CONFIGS['tiny_nv'])
    fixtures.CONFIGS['cfg2_ri'] = dict(fixtures.CONFIGS['tiny_nv'], variant='ri')
    try:
        # the default workload: the reference-init network, precision 'auto' (-> one fp16 product), parity = truth protocol + plain 1e-3
        monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', 'tiny_ri', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--streams', '1', '--exact-grids'])
        bench.main()
        res = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert res['n_gpus'] == 1 and res['steps'] == 1 and res['unit'] == 'rays/s' and res['value'] > 0
        assert res['config']['mlp_precision'] == 'f16' and res['config']['mlp_precision_requested'] == 'auto' and res['dtype'].startswith('f16 MFMA')
        assert res['config']['mlp_precision_auto']['choice'] == 'f16' and res['config']['exact_grids'] is True and res['config']['valid_samples'] > 0
        assert res['roofline']['kernel'] == 'nerf_mlp3_kernel' and res['roofline']['frac'] > 0 and res['roofline']['traffic'] == 12345     # (round 5's default form)
        assert list(res)[:7] == ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step'] and res['config']['caller_streams'] == 1
        assert res['latency_ms_per_frame'] > 0 and 'same input tensors' in res['config']['inputs']
        assert res['roofline']['executed_mfma_flop'] < res['roofline']['algorithmic_flop_per_launch'] * 1.2 and res['rccl_ranks'] == 1
        assert 'frame_timeline_ms' in res and res['torch_gpu_baseline']['value'] > 0 and res['torch_gpu_baseline']['speedup_vs_it'] > 0
        sec = res['secondary']
        assert sec['mlp_kernel_alone']['f16x3']['kernel_ms'] > 0 and 1e-6 < sec['mlp_kernel_alone']['f16']['rgb_rel_err_max_vs_f16x3'] < 1e-3
        assert sec['mlp_kernel_alone']['bf16']['rgb_rel_err_max_vs_f16x3'] > sec['mlp_kernel_alone']['f16']['rgb_rel_err_max_vs_f16x3']
        assert all(sec['mlp_kernel_alone'][k + '_two_launches']['bit_identical_to_one_launch'] for k in ('f16', 'bf16', 'f16x3'))
        assert all(sec['mlp_kernel_alone'][f'{k}_{f}']['bit_identical_to_one_tile_kernel'] for k in ('f16', 'bf16') for f in ('pipelined', 'two_tiles'))
        assert sec['fresh_inputs']['rays_per_s'] > 0 and sec['fresh_inputs']['token_rerenders'] >= 0
        assert sec['cfg3_ri']['rays_per_s'] > 0 and sec['cfg2_dense_ri']['valid_fraction'] > sec['cfg3_ri']['valid_fraction']
        assert sec['cfg2']['mlp_precision'] == 'f16x3'                               # the adversarial weights stay fp32-grade under `auto`
        assert res['value_dense'] == sec['cfg2_dense_ri']['rays_per_s'] and res['valid_fraction_dense'] == sec['cfg2_dense_ri']['valid_fraction']
        par = res['parity']
        assert res['parity_ok'] is True and par['ok'] and par['truth_ok'] and par['plain_ok'] and par['flips_ok'] and par['image_ok']
        assert par['samples']['sigma_rel_max'] < 1e-3 and par['samples']['rgb_rel_max'] < 1e-3
        assert par['truth']['table']['sigma']['max']['ours_vs_truth'] < 1e-3 and par['image']['psnr_vs_oracle_db'] > 60.0
        # the adversarial workload under f16x3: the truth protocol alone
        monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', 'tiny', '--precision', 'f16x3', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--streams', '1',
                                          '--no-secondary'])
        bench.main()
        res = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert res['config']['mlp_precision'] == 'f16x3' and res['parity_ok'] is True and 'plain_ok' not in res['parity']
        assert res['roofline']['kernel'] == 'nerf_mlp_kernel'
        # a precision that misses the tolerance: the JSON line says so and the process exits non-zero
        monkeypatch.setattr(sys, 'argv', ['bench.py', '--config', 'tiny', '--precision', 'bf16', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--streams', '1',
                                          '--no-secondary', '--no-pmc'])
        with pytest.raises(SystemExit) as ex:
            bench.main()
        res = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert ex.value.code == 3 and res['parity_ok'] is False and res['parity']['truth_ok'] is False
    finally:
        fixtures.CONFIGS.update(keep)


def test_bench_generator_dry_run(cpu_product, monkeypatch, capsys):
    """bench_generator.py (TriPlaneGenerator.forward with its own producers, stage table, cached-backbone variant: what bench.py reports as
    `secondary.generator_forward`) executed on the host build at a tiny size with a narrowed backbone -- its plumbing, not its numbers."""
    import json
    import sys
    import bench
    import bench_generator as BG
    from sherf_amd.renderer import ImportanceRenderer
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    monkeypatch.setattr(bench, '_device', lambda lrank: torch.device('cpu'))
    monkeypatch.setattr(ImportanceRenderer, '_side', lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})())
    monkeypatch.setattr(BG._Mark, 'on_gpu', False)
    monkeypatch.setattr(sys, 'argv', ['bench_generator.py', '--config', 'tiny_ri', '--steps', '1', '--warmup', '0', '--small-backbone'])
    BG.main()
    res = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert res['unit'] == 'rays/s' and res['value'] > 0 and res['output']['finite'] and res['output']['image_raw'] == [1, 3, 32, 32]
    full, cached = res['recomputed_every_frame'], res['use_cached_backbone']
    assert set(BG.STAGES) <= set(full['stages_ms']) and full['floor'] in BG.STAGES
    assert 'backbone_synthesis' not in cached['stages_ms'] and {'encoder_2d', 'mapping', 'encoder_2d_feature', 'glue', 'renderer'} <= set(cached['stages_ms'])
    assert res['config']['mlp_form'] == 'pipelined' and res['config']['mlp_precision'] == 'f16'


@pytest.mark.parametrize('partition', ['views', 'rays', 'views-3-streams'])
def test_bench_two_ranks_dry_run(cpu_product, partition):
    """bench.py as the driver launches it for N > 1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), two ranks over gloo
    on the host build: every rank renders its own view (weak scaling) or its interleaved ray tiles of ONE frame (--partition rays,
    strong scaling), the step ends with the all_gather of the tiles, rank 0 prints the one line."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   SHERF_DIST_BACKEND='gloo', SHERF_HIPCPU_LIB=_lib.LIB_PATH, OMP_NUM_THREADS='2', SHERF_BENCH_PARTITION=partition.split('-')[0])
        if partition == 'views-3-streams':        # round 5: several caller streams per rank, one gather queue per stream (4 steps on 3 streams: every queue used, one wraps)
            env.update(SHERF_BENCH_STREAMS='3', SHERF_BENCH_STEPS='4')
        procs.append(subprocess.Popen([sys.executable, os.path.join(G.ROOT, 'tests', 'bench_dist_child.py')], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-600:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith('{')]          # rank 0 prints ONE line
    res = json.loads(lines[0])
    per_step = res['ms_per_step'] * 1e-3
    ex = res['config']['exchange']
    assert ex['bytes_per_rank'] > 0 and ex['predicted_us_at_8_gpus'] > 20 and 'none by us' in ex['measured_scaling']
    if partition.startswith('views'):
        if partition == 'views-3-streams':
            # 3 frames to create the workspaces + 1 warm-up + 4 timed + 3 for the host's own cost = 11 gathers, every one awaited (drain)
            assert res['config']['caller_streams'] == 3 and res['steps'] == 4 and ex['gathers_issued'] == 11
        assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['config']['parallelism'] == 'views x2'
        assert res['value'] > 0 and abs(res['value'] - 2 * res['config']['rays'] * res['steps'] / (per_step * res['steps'])) < 1e-6 * res['value']
    else:
        assert res['n_gpus'] == 2 and res['scaling'] == 'strong' and res['config']['parallelism'] == 'ray tiles x2 (one frame)'
        assert res['config']['rays'] == 1024 and abs(res['value'] - 1024 / per_step) < 1e-6 * res['value']      # the FRAME's rays per second


@pytest.mark.slow
def test_bench_train_two_ranks_dry_run(cpu_product):
    """Round 6: bench_train.py ITSELF as the driver would launch BASELINE config 5 on two GPUs -- two ranks over gloo on the host
    build: the reference's start-up broadcast of rank 0's weights (training_loop.py:231-236; rank 1 is perturbed first, so the broadcast is what makes the ranks
    equal), one training step (forward, backward through the kernels' sources, the flat-gradient all-reduce of training_loop.py:374-383, Adam), ONE JSON line."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SHERF_DIST_BACKEND='gloo',
                   SHERF_HIPCPU_LIB=_lib.LIB_PATH, SHERF_HIPCPU_LIB_BWD=_lib.LIB_BWD_PATH, OMP_NUM_THREADS='2', HIPCPU_THREADS='4')
        procs.append(subprocess.Popen([sys.executable, os.path.join(G.ROOT, 'bench_train.py'), '--gpus', '2', '--config', 'tiny', '--steps', '1', '--warmup', '0'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith('{')]          # rank 0 prints ONE line
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['rccl_ranks'] == 2 and res['scaling'] == 'weak' and res['steps'] == 1
    assert res['dist']['params_equal_after_broadcast'] is True and res['dist']['broadcast_tensors'] > 100
    assert np.isfinite(res['final_loss']) and res['value'] > 0 and set(res['phases_ms']) == {'forward', 'backward', 'allreduce_adam'}


def test_bench_launches_its_own_ranks(cpu_product):
    """`python bench.py --gpus 2 --config tiny` with NO launcher in front (the form bench.py's docstring advertises and the driver uses for
    N = 1): the script re-executes itself under torch.distributed.run, two ranks over gloo on the host build, and rank 0's line says
    n_gpus == 2 with both ranks' devices; started by a launcher with a different world size it refuses."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SHERF_HIPCPU_LIB=_lib.LIB_PATH, SHERF_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    base = [sys.executable, os.path.join(G.ROOT, 'bench.py'), '--gpus', '2', '--config', 'tiny', '--steps', '1', '--warmup', '1', '--no-cpu-baseline',
            '--no-torch-gpu-baseline']
    for extra, par in (([], 'views x2'), (['--partition', 'rays'], 'ray tiles x2 (one frame)')):
        r = subprocess.run(base + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-800:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1
        res = json.loads(lines[0])
        assert res['n_gpus'] == 2 and res['rccl_ranks'] == 2 and len(res['rank_devices']) == 2 and res['config']['parallelism'] == par
    # a launcher that started ONE rank for --gpus 2: refused, no line
    r = subprocess.run(base, env=dict(env, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 2 and 'refusing' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_ray_tile_sharding_two_ranks(cpu_product):
    """sherf_amd.dist.render_ray_tiles: one frame split over two ranks by interleaved ray tiles (gloo), the REAL renderer on the host
    build; rgb / depth / acc of the gathered frame bit-equal to the single-process frame -- depth included: ray_marcher.py:57 clamps
    with the frame-wide depth range, which a rank must not replace by its subset's."""
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   SHERF_HIPCPU_LIB=_lib.LIB_PATH, OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(G.ROOT, 'tests', 'dist_raytile_child.py')], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-900:] for o in outs]
    assert all('RAYTILE_OK' in o[0] for o in outs)


def test_eval_mode_batchnorm_and_edge_cases(cpu_product):
    P.test_eval_mode_batchnorm_uses_running_stats()
    P.test_no_valid_samples_returns_background()
    P.test_white_back_identity()
    P.test_ragged_shapes(17, 23, 33)
    P.test_units_ray_sampler_and_dense_marcher()
    P.test_dataset_rays_on_device()


def test_generator_glue(cpu_product):
    P.test_generator_glue_vertex_features_and_voxelisation()
    P.test_generator_synthesis_end_to_end()


def check_fused_glue():
    """csrc/glue.hip (sherf_vertex_features, sherf_voxelize: SURVEY 8f rank 1) against the unmodified reference's glue outputs (golden),
    the oracle's voxelisation, and the tensor-op glue it replaces at inference time: same culled vertices, features to 1e-5, identical
    voxel coordinates / shape / bounds, identical image."""
    fx = dict(G.fixture('tiny'))
    gen = P._generator(fx)
    d = G.to_cuda(fx['input_data'])
    feat = G.to_cuda(fx['obs_feat'])
    with torch.no_grad():
        f_ref, m_ref = gen.vertex_features(d, d['obs_img_all'][:, 0], feat)
        f_hip, m_hip = gen.fused_vertex_features(d, d['obs_img_all'][:, 0], feat)
        can = gen.canonical_obs_vertices(d)
        s_ref, _ = gen.prepare_sp_input(d['t_vertices'].float(), can)
        s_hip, _ = gen.fused_prepare_sp_input(d['t_vertices'].float(), can)
    m_ref, m_hip = G.plain(m_ref), G.plain(m_hip)
    # directly against the UNMODIFIED reference's own glue (tests/golden/glue_tiny.npz, oracle/make_golden.py: triplane.py:105-126) and the
    # oracle's voxelisation (pinned to the reference by renderer_tiny.npz) -- not only against our tensor-op glue
    gold = np.load(os.path.join(G.GOLDEN, 'glue_tiny.npz'))
    g_same = torch.from_numpy(gold['front_mask']) == m_hip[0]
    assert float((~g_same).float().mean()) < 2e-3                               # grazing vertices: the sign of a ~0 dot product
    assert G.rel(G.plain(f_hip)[0][g_same], torch.from_numpy(gold['vertex_feat'])[g_same]) < 1e-4
    o_sp = G.oracle_render('tiny')['sp_input']
    assert s_hip['out_sh'] == o_sp['out_sh'] and float((G.plain(s_hip['coord']) != o_sp['coord']).any(1).float().mean()) < 2e-3
    same = m_ref == m_hip
    assert tuple(m_hip.shape) == (1, 6890) and m_hip.dtype == torch.bool and float((~same).float().mean()) < 5e-4      # (dot products ~ 0)
    assert float((G.plain(f_hip) - G.plain(f_ref))[same].abs().max()) < 1e-5 * max(1.0, float(G.plain(f_ref).abs().max()))
    assert s_hip['out_sh'] == s_ref['out_sh'] and torch.equal(G.plain(s_hip['bounds']), G.plain(s_ref['bounds']))
    assert s_hip['coord'].dtype == torch.int32 and torch.equal(G.plain(s_hip['coord']), G.plain(s_ref['coord']))
    planes = G.to_cuda(fx['planes']).view(1, 96, 32, 32)
    with torch.no_grad():
        a = gen.synthesis(None, d, None, use_sr_module=False, test_flag=True, planes=planes)
        gen.fused_glue = True
        try:
            b = gen.synthesis(None, d, None, use_sr_module=False, test_flag=True, planes=planes)
        finally:
            gen.fused_glue = False
    if bool(same.all()):
        # (1e-5 on the vertex features passes through 13 sparse convolutions with batch statistics and the decoder)
        assert G.rel(G.plain(b['image_raw']), G.plain(a['image_raw'])) < 5e-4 and G.rel(G.plain(b['weights_image']), G.plain(a['weights_image'])) < 5e-4
    return float((~same).float().mean())


def test_fused_glue_kernels(cpu_product):
    check_fused_glue()


def test_renderer_helpers_the_reference_generator_calls(cpu_product):
    """ImportanceRenderer.projection / coarse_deform_target2c (the two renderer methods the reference's TriPlaneGenerator.synthesis
    calls, triplane.py:113,132) against the oracle's literal chains -- for arbitrary query points with view directions, not only the
    vertices themselves."""
    fx = G.fixture('tiny_nv')
    rend, _ = G.hip_modules()
    d = G.to_cuda(fx['input_data'])
    dd = fixtures.to_torch(fx['input_data'])
    st = O.smpl_tensors(fx['smpl'])
    op, tp = dd['obs_params'], dd['t_params']
    verts = dd['obs_vertices'][0]
    xs = torch.matmul(verts - op['Th'].view(1, 3), op['R'].view(3, 3))
    g = torch.Generator().manual_seed(3)
    pick = torch.randint(0, xs.shape[0], (777,), generator=g)
    q = xs[pick] + 0.02 * torch.randn(777, 3, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(777, 3, generator=g), dim=-1)
    _, vid = O.nearest_vertex(q, xs)
    ref_c, ref_v = O.target_to_canonical(st, op, tp, None, q, dirs, vid)
    can, vdir = rend.coarse_deform_target2c(d['obs_params'], d['obs_vertices'], d['t_params'], G.dev_tensor(q[None].clone()), G.dev_tensor(dirs[None].clone()))
    assert float((G.plain(can)[0] - ref_c).abs().max()) < 5e-6 and float((G.plain(vdir)[0] - ref_v).abs().max()) < 5e-6
    own = rend.coarse_deform_target2c(d['obs_params'], d['obs_vertices'], d['t_params'], G.dev_tensor(xs[None].clone()))   # triplane.py:132
    ref_own, _ = O.target_to_canonical(st, op, tp, None, xs, None, torch.arange(xs.shape[0]))
    assert float((G.plain(own)[0] - ref_own).abs().max()) < 5e-6
    xy, mask = rend.projection(d['obs_vertices'].reshape(1, -1, 3), d['obs_R_all'], d['obs_T_all'], d['obs_K_all'], rend.SMPL_NEUTRAL['f'])
    ref_uv = O.project_uv(verts, dd['obs_R_all'][0, 0], dd['obs_T_all'][0, 0], dd['obs_K_all'][0, 0])
    assert tuple(xy.shape) == (1, 1, xs.shape[0], 2) and tuple(mask.shape) == (1, xs.shape[0]) and mask.dtype == torch.bool
    assert float((G.plain(xy)[0, 0] - ref_uv).abs().max()) < 1e-3                       # pixels
    only_xy = rend.projection(d['obs_vertices'].reshape(1, -1, 3), d['obs_R_all'], d['obs_T_all'], d['obs_K_all'])
    assert torch.equal(G.plain(only_xy), G.plain(xy)) and 0.2 < float(G.plain(mask).float().mean()) < 0.8


def check_without_transformer():
    """ImportanceRenderer(use_trans=False) (renderer.py:261, 427): the HIP frame against tests/golden/renderer_tiny_ri_notrans.npz (the unmodified
    reference built the same way) -- fp32-grade f16x3 to 1e-4 per sample, the single-product launch forms bit-identical to each other and within
    their class of it; the backward is refused.  Shared by the host-build test and tests/test_gpu_glue.py."""
    g = np.load(os.path.join(G.GOLDEN, 'renderer_tiny_ri_notrans.npz'))
    ref = G.hip_render('tiny_ri', precision='f16x3', use_trans=False)
    nv = int(ref['last']['ws']['counters'][0])
    assert nv == int(g['n_valid']) and ref['rend'].transformer is None
    so = G.plain(ref['last']['ws']['sample_out'][:nv])
    assert G.rel(so[:, :3], g['sample_rgb']) < 1e-4 and G.rel(torch.relu(so[:, 3]), np.maximum(g['sample_sigma'], 0)) < 1e-4
    assert G.rel(ref['rgb'], g['rgb']) < 1e-4 and G.rel(ref['acc'], g['acc'][:, 0]) < 1e-4
    one = G.hip_render('tiny_ri', precision='f16', use_trans=False, options=dict(mlp_form='one'))
    assert G.rel(one['rgb'], g['rgb']) < 2e-3
    for opts in (dict(mlp_form='pipelined'), dict(mlp_form='two_tiles'), dict(mlp_split=True)):
        b = G.hip_render('tiny_ri', precision='f16', use_trans=False, options=opts)
        assert torch.equal(b['rgb'], one['rgb']) and torch.equal(b['acc'], one['acc']), opts
    with_t = G.hip_render('tiny_ri', precision='f16x3')
    assert G.rel(with_t['rgb'], ref['rgb']) > 1e-3                       # the transformer does change the image


def check_feature_branch_switches():
    """ImportanceRenderer(use_1d_feature, use_2d_feature, use_3d_feature) in every combination run_model distinguishes (renderer.py:261-269, 405-425;
    round 6): the HIP frame against goldens of the unmodified reference built with the same switches (tests/golden/renderer_tiny_ri_f<abc>.npz,
    oracle/make_golden.py branches) -- fp32-grade f16x3 to 1e-4 per sample and on the image, the fp16 configuration within its class; a switch
    combination with a single 2-D or 3-D source renders the tri-plane features alone, as the reference's if / elif chain does; the backward is refused."""
    full = G.hip_render('tiny_ri', precision='f16x3')
    for tag in ('110', '101', '011', '100'):
        br = tuple(ch == '1' for ch in tag)
        g = np.load(os.path.join(G.GOLDEN, f'renderer_tiny_ri_f{tag}.npz'))
        h = G.hip_render('tiny_ri', precision='f16x3', branches=br)
        nv = int(h['last']['ws']['counters'][0])
        assert nv == int(g['n_valid']), tag
        rend = h['rend']
        assert rend.feature_branches() == br
        assert (not hasattr(rend, 'conv1d_reprojection')) if tag == '100' else rend.conv1d_reprojection.weight.shape[1] == 64
        so = G.plain(h['last']['ws']['sample_out'][:nv])
        e_rgb, e_sig = G.rel(so[:, :3], g['sample_rgb']), G.rel(torch.relu(so[:, 3]), np.maximum(g['sample_sigma'], 0))
        e_img = max(G.rel(h['rgb'], g['rgb']), G.rel(h['acc'], g['acc'][:, 0]))
        print(f'branches {tag}: per-sample rgb {e_rgb:.2e} sigma+ {e_sig:.2e}, image {e_img:.2e}')
        assert e_rgb < 1e-4 and e_sig < 1e-4 and e_img < 1e-4, (tag, e_rgb, e_sig, e_img)
        assert G.rel(h['rgb'], full['rgb']) > 1e-3, tag                   # every switch changes the image
        h16 = G.hip_render('tiny_ri', precision='f16', branches=br)
        assert G.rel(h16['rgb'], g['rgb']) < 2e-3, tag
    # a single 2-D / 3-D source: no branch of run_model's chain -> the tri-plane samples alone (== '100')
    a = G.hip_render('tiny_ri', precision='f16x3', branches=(True, False, False))
    for br in ((False, True, False), (False, False, True), (False, False, False)):
        b = G.hip_render('tiny_ri', precision='f16x3', branches=br)
        assert torch.equal(a['rgb'], b['rgb']), br


def check_whole_generator():
    """TriPlaneGenerator.forward(input_data, z, c) as the reference calls it (test_loop.py:189-190): ResNet-18 code -> mapping ->
    StyleGAN2 tri-planes (bias_act / upfirdn2d kernels), ResNet-18 feature map, glue, renderer -- against the oracle renderer fed with
    the planes / feature map the same producers output.  Shared by the host-build test below and tests/test_gpu_producers.py."""
    from sherf_amd.triplane import TriPlaneGenerator
    fx = dict(G.fixture('tiny'))
    rend, dec = G.hip_modules.__wrapped__()
    gen = TriPlaneGenerator(512, 0, 48, True, True, True, True, True, img_resolution=32, img_channels=3, mapping_kwargs=dict(num_layers=2),
                            rendering_kwargs=dict(fx['options']), smpl=G.smpl(), channel_base=512, channel_max=16, num_fp16_res=0,
                            conv_clamp=None, fused_modconv_default='inference_only')
    gen.renderer, gen.decoder = rend, dec
    fixtures.load_seeded_state(gen.conv1d_projection, 'generator.conv1d_projection.')
    for name, mod in (('backbone', gen.backbone), ('encoder_2d', gen.encoder_2d), ('encoder_2d_feature', gen.encoder_2d_feature)):
        with torch.no_grad():
            for n, t in list(mod.named_parameters()) + list(mod.named_buffers()):
                v = None if n.endswith('resample_filter') else fixtures.seeded_param(f'{name}.{n}', t.shape)
                if v is not None:
                    t.copy_(torch.from_numpy(np.asarray(v, np.float32).reshape(tuple(t.shape))).to(t.dtype))
    gen = G.dev_module(gen)
    gen.eval(); rend.train(); dec.train()
    d = G.to_cuda(fx['input_data'])
    c0 = G.dev_tensor(torch.zeros(1, 25))
    with torch.no_grad():
        out = gen(d, None, c0, use_sr_module=False, test_flag=True, noise_mode='const')
        ws = gen.mapping(None, c0, input_img=d['obs_img_all'][:, 0])
        planes = gen.backbone.synthesis(ws, noise_mode='const')
        feat = gen.encoder_2d_feature(d['obs_img_all'][:, 0], extract_feature=True)
    assert out['image_raw'].shape == (1, 3, 32, 32) and planes.shape == (1, 96, 256, 256) and feat.shape == (1, 64, 16, 16)
    st = O.smpl_tensors(fx['smpl'])
    state = {k: torch.from_numpy(fixtures.seeded_param(k, s)) for k, s in (('generator.conv1d_projection.weight', (32, 96, 1)),
                                                                          ('generator.conv1d_projection.bias', (32,)))}
    dd = fixtures.to_torch(fx['input_data'])
    fo, _ = O.vertex_features(state, st, dd['obs_vertices'][0], dd['obs_R_all'], dd['obs_T_all'], dd['obs_K_all'], G.plain(feat)[0], dd['obs_img_all'][0, 0])
    fx['vertex_feat'] = fo.numpy()
    fx['planes'] = G.plain(planes).view(1, 3, 32, 256, 256).numpy()
    fx['obs_feat'] = G.plain(feat).numpy()
    o = O.render_from_fixture(fx, G.seeded_state(), training=True, keep=False)
    img = G.plain(out['image_raw'])[0].permute(1, 2, 0).reshape(-1, 3)
    assert O.psnr(img, o['rgb']) > 55.0
    assert G.rel(G.plain(out['weights_image']).reshape(-1), o['acc']) < 2e-3


def check_snapshot_and_training_step_against_reference_golden(monkeypatch):
    """SURVEY 8(f) ranks 3 + 4 against the UNMODIFIED reference's own outputs (tests/golden/trainstep_tiny_nv.npz, written by
    oracle/make_golden_trainstep.py from `training.triplane.TriPlaneGenerator`, its persistence and `StyleGAN2Loss.accumulate_gradients`):

      f4  sherf_amd's generator exposes the reference snapshot's names / shapes (all but the unused super-resolution head) (`copy_params_and_buffers(require_all=True)`),
          takes the same values by name, renders the frame the reference rendered from its snapshot, and a pickled snapshot of it resumes
          into a differently initialised generator to the same bits;
      f3  one generator step (sherf_amd.loss: loss.py:103-176 + training_loop.py:365-383) returns the reference's loss terms and, for every
          one of its 240 parameter gradients, the reference's fingerprint.

    Shared by the host-build test below and tests/test_gpu_producers.py (the MI355X)."""
    import copy
    import io
    import pickle
    import sys
    from oracle import make_golden_trainstep as MG
    from sherf_amd import loss as L
    from sherf_amd import triplane as TP
    ref = np.load(os.path.join(G.GOLDEN, f'trainstep_{MG.CFG}.npz'))
    fx = dict(G.fixture(MG.CFG))
    normals = G.dev_tensor(torch.from_numpy(ref['normals']))
    monkeypatch.setattr(TP, 'compute_normal', lambda vertices, faces: normals)     # the reference's ill-defined normals, as it computed them

    def build(seed):
        torch.manual_seed(seed)
        kw = MG.gen_kwargs(fx['options'])
        gen = TP.TriPlaneGenerator(kw.pop('z_dim'), kw.pop('c_dim'), kw.pop('w_dim'), True, True, True, True, True, smpl=G.smpl(),
                                   **{k: v for k, v in kw.items() if not k.startswith('use_')})
        gen.fused_glue = False                                                      # (the fused glue derives its own normals: tests/test_gpu_glue.py)
        return gen

    gen = build(0)
    MG.load_whole_generator_state(gen)
    named = list(gen.named_parameters()) + list(gen.named_buffers())
    ours = {n: 'x'.join(str(int(x)) for x in t.shape) for n, t in named}
    # (the 2x super-resolution head is the one module this package does not carry: every SHERF script runs --use_sr_module False, its
    #  entries of a reference snapshot are simply not consumed; everything else must be there under the same name and shape)
    keep = [i for i, n in enumerate(ref['names'].tolist()) if not n.startswith('superresolution.')]
    theirs = {ref['names'][i].item(): ref['shapes'][i].item() for i in keep}
    assert ours == theirs, sorted(set(ours.items()) ^ set(theirs.items()))[:6]
    assert sorted(n for n, _ in gen.named_parameters()) == sorted(ref['names'][i].item() for i in keep if ref['is_param'][i])   # (parameter vs buffer)
    l2 = {ref['names'][i].item(): float(ref['state_l2'][i]) for i in keep}
    for n, t in named:
        assert abs(float(t.detach().double().norm()) - l2[n]) <= 1e-6 * max(1.0, l2[n]), n
    gen = G.dev_module(gen)
    d = G.to_cuda(MG.batch(fx))
    d['mask_at_box_all'] = G.plain(d['mask_at_box_all']).bool() if G.CPU_SHIM else d['mask_at_box_all'].bool()
    z, c0 = G.dev_tensor(torch.zeros(1, 512)), G.dev_tensor(torch.zeros(1, 0))

    def render(g):
        g.eval(); g.renderer.train(); g.decoder.train()
        g.renderer.enable_autograd = False
        with torch.no_grad():
            out = g(d, z, c0, use_sr_module=False, noise_mode='const')
        return {k: G.plain(v) for k, v in out.items() if torch.is_tensor(v)}

    snap = copy.deepcopy(gen)                                                       # before the render moves the running statistics
    a = render(gen)
    img_ref = torch.from_numpy(ref['image_raw'])
    assert O.psnr(a['image_raw'], img_ref) > 50.0, O.psnr(a['image_raw'], img_ref)
    assert G.rel(a['image_raw'], img_ref) < 2e-3 and G.rel(a['weights_image'], torch.from_numpy(ref['weights_image'])) < 2e-3
    # our snapshot, written as training_loop.py:563-579 writes it, resumes to the same bits
    buf = io.BytesIO()
    pickle.dump(dict(G=copy.deepcopy(snap).eval().requires_grad_(False).cpu(), G_ema=None), buf)
    buf.seek(0)
    data = pickle.load(buf)
    gen2 = build(1)
    with torch.no_grad():                                                           # misc.copy_params_and_buffers(require_all=True)
        src = dict(list(data['G'].named_parameters()) + list(data['G'].named_buffers()))
        for n, t in list(gen2.named_parameters()) + list(gen2.named_buffers()):
            t.copy_(src[n].detach())
    gen2 = G.dev_module(gen2)
    b = render(gen2)
    assert torch.equal(a['image_raw'], b['image_raw']) and torch.equal(a['weights_image'], b['weights_image'])

    # ---- f3 ----
    sys.path.insert(0, os.path.join(G.ROOT, 'oracle', 'ref_shims'))
    try:
        import lpips as lpips_stand_in                                              # the SAME stand-in the golden run drove the reference's loss with
    finally:
        sys.path.pop(0)
    gen3 = build(2)
    with torch.no_grad():
        for n, t in list(gen3.named_parameters()) + list(gen3.named_buffers()):
            t.copy_(src[n].detach())
    gen3 = G.dev_module(gen3)
    gen3.eval(); gen3.renderer.train(); gen3.decoder.train()
    gen3.renderer.enable_autograd = True
    H, W = d['obs_img_all'].shape[-2:]
    lp = lpips_stand_in.LPIPS()
    loss = L.ReconstructionLoss(torch.device('cpu') if G.CPU_SHIM else torch.device('cuda'), gen3, lpips_fn=lambda x, y: lp(x, y).reshape(-1),
                                neural_rendering_resolution_initial=max(H, W))
    opt = torch.optim.Adam([p for p in gen3.parameters()], lr=2e-4, betas=(0.0, 0.99), eps=1e-8)
    grads = {}
    orig_update = L.update_weights

    def update(module, opt_, num_gpus=None, scheduler=None):                        # the gradients as the optimiser sees them (after sanitising)
        sdist_params = [p for p in module.parameters() if p.numel() > 0]
        L.sdist.allreduce_flat_grads(sdist_params, world_size=num_gpus)
        grads.update({n: G.plain(p.grad.detach()).clone() for n, p in module.named_parameters() if p.grad is not None})
        opt_.step()
    monkeypatch.setattr(L, 'update_weights', update)
    out = L.training_step(gen3, opt, loss, d, z, G.dev_tensor(torch.zeros(1, 25)), gain=1, num_gpus=1, use_sr_module=False)
    monkeypatch.setattr(L, 'update_weights', orig_update)
    terms = [float(G.plain(t.detach()).reshape(-1)[0]) for t in out]
    for x, y in zip(terms, ref['loss_terms']):
        assert abs(x - y) <= 2e-4 * max(1.0, abs(y)), (terms, ref['loss_terms'].tolist())
    assert sorted(grads) == ref['grad_names'].tolist(), sorted(set(grads) ^ set(ref['grad_names'].tolist()))[:6]
    worst = {}
    for n in grads:
        f, r = O.grad_fingerprint(grads[n].float()), ref['grad.' + n]
        worst[n] = (abs(f[2] - r[2]) / (r[2] + 1e-30), float(np.linalg.norm(f[3:] - r[3:]) / (np.linalg.norm(r[3:]) + 1e-30)), r[2])
    top = sorted(((v[0], v[1], n) for n, v in worst.items() if v[2] > 1e-9), reverse=True)[:5]
    print(f'training step vs the reference: loss terms {terms} vs {ref["loss_terms"].tolist()}; {len(grads)} gradients, worst (norm, samples) {top}')
    enc = lambda k: 'encoder_3d' in k
    for n, (en, es, nr) in worst.items():
        if nr <= 1e-9:
            continue
        # (sparse-encoder entries: rounds 2-5 allowed 0.15 -- our three-product convolutions lost the low bits of their lo halves; round 6: one bound)
        assert en < 1e-2 and es < 2e-2, (n, en, es)
    return terms, worst


def test_snapshot_and_training_step_against_reference_golden(cpu_product, monkeypatch):
    from sherf_amd.renderer import ImportanceRenderer
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    monkeypatch.setattr(ImportanceRenderer, '_side', lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})())
    check_snapshot_and_training_step_against_reference_golden(monkeypatch)


def test_whole_generator_with_its_own_producers(cpu_product, monkeypatch):
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))       # module parameters are "device" tensors too
    check_whole_generator()


def test_full_training_step_with_the_reconstruction_loss(cpu_product, monkeypatch):
    """BASELINE config 5, the whole step as training_loop.py:354-386 runs it, on the host build: TriPlaneGenerator with its own producers
    -> renderer recorded as one autograd node -> reconstruction loss of loss.py:103-176 (sherf_amd/loss.py) -> backward through the HIP
    backward pipeline and on into the StyleGAN2 / ResNet producers -> flat-gradient sanitising -> optimiser step."""
    from sherf_amd import loss as L
    from sherf_amd.triplane import TriPlaneGenerator
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    fx = dict(G.fixture('tiny_nv'))
    rend, dec = G.hip_modules.__wrapped__()
    rend.enable_autograd = True
    opts = dict(fx['options']); opts['density_noise'] = 0
    gen = TriPlaneGenerator(512, 0, 48, True, True, True, True, True, img_resolution=32, img_channels=3, mapping_kwargs=dict(num_layers=2),
                            rendering_kwargs=opts, smpl=G.smpl(), channel_base=512, channel_max=16, num_fp16_res=0,
                            conv_clamp=None, fused_modconv_default='inference_only')
    gen.renderer, gen.decoder = rend, dec
    fixtures.load_seeded_state(gen.conv1d_projection, 'generator.conv1d_projection.')
    for name, mod in (('backbone', gen.backbone), ('encoder_2d', gen.encoder_2d), ('encoder_2d_feature', gen.encoder_2d_feature)):
        with torch.no_grad():
            for n, t in list(mod.named_parameters()) + list(mod.named_buffers()):
                v = None if n.endswith('resample_filter') else fixtures.seeded_param(f'{name}.{n}', t.shape)
                if v is not None:
                    t.copy_(torch.from_numpy(np.asarray(v, np.float32).reshape(tuple(t.shape))).to(t.dtype))
    gen.eval(); rend.train(); dec.train()
    d = G.to_cuda(fx['input_data'])
    H, W = d['obs_img_all'].shape[-2:]
    g = torch.Generator().manual_seed(4)
    d['img_all'] = torch.rand(1, 1, 3, H, W, generator=g)
    d['bkgd_msk_all'] = (torch.rand(1, 1, H * W, generator=g) > 0.5).to(torch.uint8)
    d['mask_at_box_all'] = G.plain(d['mask_at_box_all']).bool()
    assert int(d['mask_at_box_all'].sum()) > 100
    watch = {'decoder.pts_linears.0.weight': dec.pts_linears[0].weight, 'renderer.conv1d_reprojection.weight': rend.conv1d_reprojection.weight,
             'renderer.encoder_3d.conv0.0.weight': rend.encoder_3d.conv0[0].weight, 'conv1d_projection.weight': gen.conv1d_projection.weight,
             'backbone.synthesis.b256.torgb.weight': gen.backbone.synthesis.b256.torgb.weight,
             'encoder_2d_feature.conv1.weight': gen.encoder_2d_feature.feature_extractor.conv1.weight if hasattr(gen.encoder_2d_feature, 'feature_extractor')
             else next(gen.encoder_2d_feature.parameters())}
    before = {k: G.plain(v).clone() for k, v in watch.items()}
    loss = L.ReconstructionLoss(torch.device('cpu'), gen, neural_rendering_resolution_initial=32)
    opt = torch.optim.SGD([p for p in gen.parameters()], lr=1e-3)
    out = L.training_step(gen, opt, loss, d, torch.zeros(1, 512), torch.zeros(1, 25), gain=1, num_gpus=1, use_sr_module=False)
    total, img_l, acc_l, ssim_s, lp, _ = [t.detach() for t in out]
    assert all(bool(torch.isfinite(G.plain(t)).all()) for t in (total, img_l, acc_l, ssim_s)) and float(lp) == 0.0
    assert abs(float(total) - (100 * float(img_l) + 10 * float(acc_l) + 1 - float(ssim_s))) < 1e-3 * abs(float(total))
    for k, v in watch.items():
        assert v.grad is not None and bool(torch.isfinite(G.plain(v.grad)).all()) and float(G.plain(v.grad).abs().max()) > 0, k
        assert not torch.equal(G.plain(v), before[k]), k                       # the step moved it
    # a second forward with the updated weights lowers the loss it was stepped on (small step along -grad)
    with torch.no_grad():
        rend.enable_autograd = False
        gen_img, _ = loss.run_G(d, torch.zeros(1, 512), torch.zeros(1, 25), 32, use_sr_module=False)
        after = loss.terms(gen_img, d)[0]
    assert float(after) < float(total)


def test_reference_init_variant_plain_tolerance(cpu_product):
    """The well-conditioned workload (SURVEY 8(d)'s network, band-limited tables): every sample within 1e-3 of the oracle."""
    P.test_margin_protocol_whole_frame('tiny_ri')
    P.test_auto_precision_is_calibrated_per_weights()


def test_size_independent_properties_and_rotation(cpu_product):
    P.test_deterministic_and_ray_independent()
    P.test_global_rotation_flip_rate()
    if os.environ.get('SHERF_SLOW'):            # (one ragged shape runs with the edge cases above; S = 128 / S = 2 here only on request,
        P.test_ragged_shapes(9, 31, 128)        #  and always on the device: tests/test_gpu_parity.py)
        P.test_ragged_shapes(5, 7, 2)


@pytest.mark.skipif(not os.environ.get('SHERF_SLOW'), reason='BASELINE config 1 (128x128x32) and the per-sample precision sweep on the CPU (~1 min): SHERF_SLOW=1')
def test_config1_and_per_sample_precision(cpu_product):
    P.test_per_sample_sigma_rgb('f16x3', 1e-3, 1e-3)
    P.test_per_sample_sigma_rgb('bf16', 5e-2, 5e-2)
    P.test_end_to_end_vs_oracle_and_reference_golden('cfg1')
    P.test_margin_protocol_whole_frame('cfg1')              # adversarial weights: against the float64 truth
    P.test_margin_protocol_whole_frame('cfg1_ri')           # reference-init network: within 1e-3 of the oracle outright


def test_full_size_backward_check_plumbing(cpu_product, monkeypatch):
    """tests/test_gpu_backward.py::test_full_size_backward_against_oracle_autograd's own procedure (oracle autograd on a device, two
    HIP backward runs on fresh modules, per-gradient relative error + run-to-run spread) on the host build at the tiny size."""
    from tests import test_gpu_backward as GB
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    ours_t, ref_t = GB._full_size_backward('tiny_ri', 'cpu')          # (asserts the float64-truth rule and the run-to-run spread itself)
    assert len(ours_t) >= 81 and set(ours_t) == set(ref_t)
    nenc = [k for k in ours_t if 'encoder_3d' not in k and k != 'input.vertex_feat']
    # (at this size the fp32 reference's decoder gradients are themselves 1.2e-3 from the float64 truth -- small sums of cancelling per-sample terms --
    #  and ours sit at the same distance to three digits: outside the encoder the two fp32 backward passes agree far better than either holds the truth)
    assert all(abs(ours_t[k] - ref_t[k]) <= 0.2 * ref_t[k] + 1e-4 for k in nenc)


@pytest.mark.parametrize('use_trans', [True, False])
def test_training_step_through_autograd_matches_reference_gradients(cpu_product, monkeypatch, use_trans):
    """BASELINE config 5 on the CPU: forward recorded as ONE autograd node (renderer.enable_autograd), stub loss,
    loss.backward() through the native backward pipeline; gradients against the fingerprints of the UNMODIFIED reference's
    gradients (tests/golden/grad_tiny_nv.npz) and against the explicit backward evaluated at our forward point.
    use_trans = False (round 6): the same through a renderer built without its transformer, against the reference built the same way
    (tests/golden/grad_tiny_nv_notrans.npz, oracle/make_golden.py grad_notrans)."""
    from sherf_amd import backward as B
    from sherf_amd.voxel import SparseConvTensor
    from tests.bwd_emulator import EmuOps
    cfg = 'tiny_nv'
    fx = G.fixture(cfg)
    ref = np.load(os.path.join(G.GOLDEN, f'grad_{cfg}.npz' if use_trans else f'grad_{cfg}_notrans.npz'))
    rend, dec = G.hip_modules.__wrapped__() if use_trans else G._hip_modules.__wrapped__('f16x3', 'seeded', False)    # fresh modules: this test updates running statistics and .grad
    assert (rend.transformer is not None) == use_trans
    rend.enable_autograd = True
    d = fixtures.to_torch(fx['input_data'])
    spi = G.oracle_render(cfg)['sp_input']
    planes = torch.from_numpy(fx['planes']).requires_grad_(True)
    obs_feat = torch.from_numpy(fx['obs_feat']).requires_grad_(True)
    vfeat = torch.from_numpy(fx['vertex_feat']).requires_grad_(True)
    sp = SparseConvTensor(vfeat, spi['coord'], spi['out_sh'], 1)
    spd = dict(coord=spi['coord'], out_sh=spi['out_sh'], batch_size=1, bounds=spi['bounds'][None])
    mean0 = rend.encoder_3d.conv0[1].running_mean.clone()
    rgb, depth, acc = rend(planes, d['obs_img_all'][:, 0], obs_feat, sp, None, spd, dec, G.dev_tensor(d['ray_o_all'][:, 0]),
                           d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, dict(fx['options']))
    assert rgb.requires_grad and acc.requires_grad and not depth.requires_grad
    assert not torch.equal(rend.encoder_3d.conv0[1].running_mean, mean0)           # a training forward updates the running statistics
    seen = {}
    orig = B.encoder_backward
    monkeypatch.setattr(B, 'encoder_backward', lambda ops, state, ctx, d_levels: seen.update(state=state, ctx=ctx, d_levels=d_levels) or orig(ops, state, ctx, d_levels))
    O.stub_loss(rgb[0], acc[0, :, 0]).backward()
    grads = {'input.planes': planes.grad, 'input.obs_feat': obs_feat.grad, 'input.vertex_feat': vfeat.grad}
    for mod, pre in ((rend, 'renderer.'), (dec, 'decoder.')):
        for n, p in mod.named_parameters():
            if p.grad is not None:
                assert p.grad.shape == p.shape, pre + n
                grads[pre + n] = p.grad
    names = [k for k in ref.files if k not in ('loss', 'ref_cpu_seconds')]
    assert set(names) == set(grads), set(names) ^ set(grads)
    # The encoder's gradients are ill-conditioned on this fixture: one level-3 activation sits within 3e-5 of the ReLU kink and
    # carries a large gradient, so a 3e-5 relative perturbation of the forward (ours differs from the fp32 reference by about
    # that: fixed-point BatchNorm statistics, f16x3 MLP) moves the reference's OWN encoder gradients by 5-6 % (measured by
    # perturbing the oracle's input).  They are therefore held to a loose bound against the reference and to a tight one
    # against the explicit backward (tests/bwd_emulator.py, itself verified against autograd and the reference in
    # tests/test_backward_math.py / test_backward_dense.py) evaluated at OUR forward point.
    enc = lambda k: 'encoder_3d' in k or k == 'input.vertex_feat'
    for k in names:
        ours, r = O.grad_fingerprint(grads[k].float()), ref[k]
        tn, tv = (1e-2, 5e-2)                  # round 6: the encoder's entries too (rounds 2-5: 0.15 -- the "3e-5 perturbation" was our convolutions' lo halves)
        assert abs(ours[2] - r[2]) < tn * r[2] + 1e-30, (k, ours[2], r[2])
        assert np.linalg.norm(ours[3:] - r[3:]) < tv * np.linalg.norm(r[3:]) + 1e-30, k
    d_feat_e, g_e = orig(EmuOps(), seen['state'], seen['ctx'], seen['d_levels'])
    assert G.rel(vfeat.grad, d_feat_e.tensor()) < 1e-4
    for k, v in g_e.items():
        assert G.rel(grads[k].reshape(-1), v.reshape(-1)) < 1e-4, k
    # a backward whose frame has been overwritten by a later forward of the same renderer must refuse, not return the other frame's
    # gradients: two forwards, then the FIRST one's backward
    for p in list(rend.parameters()) + list(dec.parameters()):
        p.grad = None
    call = lambda: rend(planes, d['obs_img_all'][:, 0], obs_feat, SparseConvTensor(vfeat, spi['coord'], spi['out_sh'], 1), None, spd, dec,
                        G.dev_tensor(d['ray_o_all'][:, 0]), d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], d, dict(fx['options']))
    rgb1, _, acc1 = call()
    rgb2, _, acc2 = call()
    with pytest.raises(RuntimeError, match='workspace has been overwritten'):
        O.stub_loss(rgb1[0], acc1[0, :, 0]).backward()
    O.stub_loss(rgb2[0], acc2[0, :, 0]).backward()                              # the latest frame's backward is fine
