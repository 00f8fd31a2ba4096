"""CPU check of the backward ORCHESTRATION of the dense stage (sherf_amd/backward_dense.py) against autograd through the
oracle, with the C entry points replaced by their torch emulation (tests/bwd_emulator.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fixtures, sherf_oracle as O
from sherf_amd.backward_dense import Mat, dense_backward
from tests.bwd_emulator import EmuOps


@pytest.fixture(scope='module')
def state(golden_dir):
    shapes = json.load(open(os.path.join(golden_dir, 'param_shapes.json')))
    return {n: torch.from_numpy(fixtures.seeded_param(n, s)) for n, s in shapes.items() if fixtures.seeded_param(n, s) is not None}


def _close(a, b, tol):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-14


@pytest.mark.parametrize('cfg', ['tiny_nv', 'tiny'])
def test_dense_backward_orchestration(cfg, state):
    fx = fixtures.renderer_inputs(cfg)
    loss, g = O.gradients_from_fixture(fx, state, stages=True)
    with torch.no_grad():
        r = O.render_from_fixture(fx, state, training=True)
        n = r['x_c'].shape[0]
        # what the forward kernels hand over: gather tokens WITHOUT the slot-2 rgb encoding, and the extras rows
        Wb = state['renderer.conv1d_reprojection.weight'][:, 32:64, 0]
        tok = r['tokens_in'].clone()
        tok[:, 2] -= O.positional_encoding(r['tap_rgb'], 5)[:, :32] @ Wb.t()
        ext = torch.zeros(n, 12)
        ext[:, 0:3], ext[:, 3:6], ext[:, 6:9] = r['x_c'], r['v_c'], r['tap_rgb']
        d_sample = torch.cat([g['stage.sample_rgb'], g['stage.sample_sigma'][:, None]], 1).contiguous()
        d_tin, grads, dWb_pe = dense_backward(EmuOps(), state, Mat(tok.reshape(-1).clone(), n, 96), Mat(ext.reshape(-1).clone(), n, 12),
                                              Mat(d_sample.reshape(-1).clone(), n, 4))
    assert _close(d_tin.tensor().view(n, 3, 32), g['stage.tokens_in'], 5e-4)
    want = [k for k in g if k.startswith('decoder.') or k.startswith('renderer.transformer.')]
    assert set(want) == set(grads), set(want) ^ set(grads)
    for k in want:
        assert grads[k].shape == g[k].shape, k
        assert _close(grads[k], g[k], 1e-3), (k, float((grads[k] - g[k]).abs().max()), float(g[k].abs().max()))
    assert _close(dWb_pe, g['stage.tokens_in'][:, 2].t() @ O.positional_encoding(r['tap_rgb'], 5)[:, :32], 1e-4)
