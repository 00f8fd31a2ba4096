"""Staged for hardware (`-m gpu_experimental`): the per-device tuner on an MI355X -- every launch shape of the MLP kernel and the
branchless gather must reproduce the default bit for bit on the device (they do on the CPU shim, tests/test_hipcpu_frame.py)."""
import pytest
import torch

from tests import gpu_common as G

pytestmark = [pytest.mark.gpu_experimental, pytest.mark.skipif(not torch.cuda.is_available(), reason='needs an MI355X')]


@pytest.mark.parametrize('cfg', ['tiny_nv', 'cfg1'])
def test_tuner_shapes_bit_identical_on_device(cfg):
    from sherf_amd import tune
    h = G.hip_render(cfg)
    rep = tune.tune_mlp(h['rend'], h['dec'])
    for name, e in rep['shapes'].items():
        print(f"{cfg} mlp {name:12s} ms={e.get('ms', float('nan')):.4f} max|d|={e.get('max_abs_diff')} ok={e.get('ok')} {e.get('error', '')}")
    g = tune.tune_gather(h['rend'], h['dec'])
    for name, e in g['variants'].items():
        print(f"{cfg} gather {name:10s} ms={e['ms']:.4f} max|d|={e['max_abs_diff']} ok={e['ok']}")
    print(f"{cfg}: best mlp shape {rep['best']}, best gather {g['best']}")
    bad = [n for n, e in rep['shapes'].items() if not e.get('ok')] + [n for n, e in g['variants'].items() if not e['ok']]
    assert not bad, bad
    again = G.hip_render(cfg)                                   # the tuner leaves the renderer and its workspace as found
    assert torch.equal(again['rgb'], h['rgb']) and torch.equal(again['acc'], h['acc'])


def test_fused_glue_kernels_on_device():
    """csrc/glue.hip on hardware: the same comparison the CPU suite runs on the host build (tests/test_hipcpu_frame.py)."""
    from tests.test_hipcpu_frame import check_fused_glue
    print('fraction of vertices whose back-face bit differs from the tensor-op glue:', check_fused_glue())


@pytest.mark.parametrize('cfg', ['tiny', 'cfg1'])
def test_exact_grids_and_gather_variants_on_device(cfg):
    """SHERF_FRAME_EXACT_GRIDS (launches sized by the frame's own sample count, one host wait) and the gather's schedule variants render
    the same bits as the default frame on the device."""
    a = G.hip_render(cfg)
    for opts in (dict(exact_grids=True), dict(gather_branchless=True), dict(gather_branchless='128'), dict(exact_grids=True, mlp_shape='4x1')):
        b = G.hip_render(cfg, options=opts)
        assert torch.equal(b['rgb'], a['rgb']) and torch.equal(b['acc'], a['acc']) and torch.equal(b['depth'], a['depth']), opts
