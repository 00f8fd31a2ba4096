"""SURVEY section 8(f) rank 4, executed: a network snapshot written by the UNMODIFIED reference exactly as its training loop writes it
(training_loop.py:563-579, `@persistence.persistent_class` pickle) is resumed through the reference's own `legacy.load_network_pkl` +
`misc.copy_params_and_buffers(require_all=True)` into the generator hosting this package's renderer -- in a process that has neither
pytorch3d nor spconv -- and renders the image the writer rendered.  Needs /root/reference (build container only)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(mode, d):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ckpt_roundtrip_child.py'), mode, d], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1500, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith('CKPT_JSON ')]
    assert r.returncode == 0 and line, (mode, r.stderr[-1500:])
    return json.loads(line[-1][len('CKPT_JSON '):])


@pytest.mark.skipif(not os.path.isdir('/root/reference/sherf'), reason='needs the reference checkout (build container only)')
def test_reference_snapshot_resumes_into_the_hosted_generator(tmp_path):
    from tests.hipcpu import build_cpu
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the kernels')
    w = _child('write', str(tmp_path))
    assert w['wrote'] > 1e6 and w['n_tensors'] > 500
    r = _child('read', str(tmp_path))
    print(r)
    assert r['hosted_renderer'] == 'sherf_amd.renderer.ImportanceRenderer' and not r['generator_class'].startswith('sherf_amd')
    assert r['pickled_generator_class'] == 'TriPlaneGenerator' and r['pickled_sparse_layer'] == 'spconv.pytorch.SubMConv3d'   # the unpickle-only container
    assert r['names_equal'] and r['n_tensors'] == w['n_tensors'] and r['sparse_weight_shape'] == [32, 3, 3, 3, 32]
    assert r['valid_samples'] > 0 and r['image_range'][1] > -0.5
    assert r['image_rel'] < 1e-3 and r['weights_rel'] < 1e-3
    assert r['second_generation_bit_equal']
