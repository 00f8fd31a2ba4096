"""The static MFMA hazard audit (tools/mfma_hazard_check.py) as a regression test on the compiler's FINAL gfx950 assembly of the MFMA
kernels (hipcc cross-compiles here, no GPU):
  * csrc/mlp.hip -- the hand-placed wait states around its asm-volatile MFMA blocks: no instruction touches an in-flight MFMA result
    (R1-R3), no VALU-written operand reaches an MFMA early (R4);
  * every MFMA kernel -- no MFMA inside an EXEC-predicated region without a skip branch (R5): MFMAs ignore EXEC on gfx950, and hipcc drops
    the s_cbranch_execz of short predicated blocks.  Round 3's "single-product sparse convolutions: right on the host build, wrong on
    the MI355X" was exactly this (30 FOLD instances of sconv3_kernel); the fix is a provably uniform condition (csrc/svox.hip)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _asm(tmp_path, src, extra=()):
    from sherf_amd import build as B
    out = subprocess.run([HIPCC] + B.FLAGS + list(extra) + ['-I' + os.path.join(ROOT, 'include'), '-c', os.path.join(B.CSRC, src), '-save-temps=obj',
                          '-o', str(tmp_path / (src + '.o'))], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    path = tmp_path / (src.replace('.hip', '') + '-hip-amdgcn-amd-amdhsa-gfx950.s')
    assert out.returncode == 0 and path.exists(), out.stdout[-600:]
    return path.read_text().splitlines()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
@pytest.mark.parametrize('src,extra,rules', [('mlp.hip', ('-fno-slp-vectorize',), 'R1 R2 R3 R4 R5'), ('svox.hip', (), 'R5'), ('bwd_gemm.hip', (), 'R5')])
def test_no_mfma_hazard_in_the_final_isa(tmp_path, src, extra, rules):
    import mfma_hazard_check as H
    lines = _asm(tmp_path, src, extra)
    seen = 0
    for name, a, b in H.kernels(lines):
        n, bad = H.check(lines, a, b)
        if n == 0:
            continue
        seen += 1
        bad = [x for x in bad if x[1] in rules.split()]
        assert not bad, (name[:80], bad[:5])
    assert seen > 0
