"""Drop-in claim of SURVEY section 8(b), executed: the UNMODIFIED reference `training.triplane.TriPlaneGenerator` hosting this package's
renderer after `sherf_amd.install.install()`, against its own unpatched run (tests/ref_dropin_child.py).  Needs /root/reference, i.e.
runs in the build container only; the kernels execute on the host through tests/hipcpu."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_install_stubs_the_cuda_only_packages_when_absent():
    """Without pytorch3d / spconv on the path `install.stub_missing_packages()` registers import stubs whose only live attribute is the
    sparse tensor container (everything else raises if called)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import sherf_amd.install as I, sherf_amd.voxel as V\n"
            "made = I.stub_missing_packages()\n"
            "import spconv.pytorch as spconv\n"
            "from pytorch3d.ops.knn import knn_points\n"
            "assert spconv.core.SparseConvTensor is V.SparseConvTensor\n"
            "try:\n    knn_points(None, None)\n    raise SystemExit('stub did not raise')\nexcept RuntimeError:\n    pass\n"
            "print(sorted(made))\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.strip().splitlines()[-1] == "['pytorch3d', 'spconv']"


@pytest.mark.skipif(not os.path.isdir('/root/reference/sherf'), reason='needs the reference checkout (build container only)')
def test_unmodified_reference_generator_hosts_our_renderer():
    from tests.hipcpu import build_cpu
    if not os.path.exists(build_cpu.CLANG):
        pytest.skip('needs the ROCm clang for the host build of the bf16 kernels')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ref_dropin_child.py')], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1200, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith('DROPIN_JSON ')]
    assert r.returncode == 0 and line, r.stderr[-800:]
    res = json.loads(line[-1][len('DROPIN_JSON '):])
    print(res)
    assert res['renderer_class'] == 'sherf_amd.renderer.ImportanceRenderer'          # the reference's generator class ...
    assert 'TriPlaneGenerator' in res['generator_class'] and not res['generator_class'].startswith('sherf_amd')     # ... hosting our renderer
    assert res['n_state'] > 500 and res['valid_samples'] > 0                           # strict state-dict load of the reference's checkpoint
    assert res['image_rel'] < 1e-3 and res['weights_rel'] < 1e-3 and res['depth_rel'] < 1e-3 and res['psnr'] > 60.0
    assert res['image_range'][1] > -0.5                                                # (a non-trivial image: not all background)
