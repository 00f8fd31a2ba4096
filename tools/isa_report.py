"""Static report of every gfx950 kernel in sherf_amd/csrc: registers, LDS, scratch, occupancy bound and instruction mix, read from the
compiler's own assembly (`hipcc -save-temps`).  No GPU needed; the numbers the design notes quote for kernels that have not been timed
yet (launch shapes of the MLP, the branchless gather) come from here.

    python tools/isa_report.py [> profiles/r01_static_isa_report.txt]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sherf_amd import build as B  # noqa: E402

SOURCES = B.SOURCES + B.SOURCES_BWD + ['ops_lib.hip', 'ops_bias_act.hip', 'ops_upfirdn2d.hip']


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), stdout=subprocess.PIPE, text=True)
    return r.stdout.splitlines()


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0]


def main():
    tmp = tempfile.mkdtemp(prefix='isa_')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    print('# static ISA report, gfx950, flags:', ' '.join(B.FLAGS))
    print('# waves/SIMD = min(8, 512 // VGPRs incl. AGPRs); LDS in bytes per workgroup; mix = static instruction counts of the kernel body\n')
    for src in SOURCES:
        path = os.path.join(B.CSRC, src)
        if not os.path.exists(path):
            continue
        r = subprocess.run([hipcc] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ['-I' + os.path.join(ROOT, 'include'), '-c', path, '-save-temps=obj', '-o', os.path.join(tmp, src + '.o')],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=tmp)
        asm = os.path.join(tmp, src.replace('.hip', '') + '-hip-amdgcn-amd-amdhsa-gfx950.s')
        if r.returncode != 0 or not os.path.exists(asm):
            print(f'## {src}: compile failed\n{r.stdout[-400:]}')
            continue
        s = open(asm).read()
        meta = {}
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
            g = lambda k: int(re.search(k + r'\s+(\d+)', m.group(2)).group(1))
            meta[m.group(1)] = dict(lds=g(r'\.amdhsa_group_segment_fixed_size'), scratch=g(r'\.amdhsa_private_segment_fixed_size'),
                                    vgpr=g(r'\.amdhsa_next_free_vgpr'), sgpr=g(r'\.amdhsa_next_free_sgpr'))
        bodies = {m.group(1): m.group(2) for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S)}
        names = sorted(meta)
        print(f'## {src}')
        for mangled, nice in zip(names, demangle(names)):
            k = meta[mangled]
            c = collections.Counter(l.split()[0] for l in bodies.get(mangled, '').splitlines()
                                    if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':'))
            valu = sum(v for n, v in c.items() if n.startswith('v_') and not n.startswith('v_mfma'))
            mfma = sum(v for n, v in c.items() if n.startswith('v_mfma'))
            salu = sum(v for n, v in c.items() if n.startswith('s_') and n not in ('s_waitcnt', 's_nop', 's_barrier'))
            lds = sum(v for n, v in c.items() if n.startswith('ds_'))
            vmem = sum(v for n, v in c.items() if n.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
            waves = min(8, 512 // max(k['vgpr'], 1))
            print(f"  {short(nice):78s} vgpr {k['vgpr']:3d} sgpr {k['sgpr']:3d} lds {k['lds']:6d} scratch {k['scratch']:3d} waves/SIMD {waves} | "
                  f"mfma {mfma:4d} valu {valu:5d} salu {salu:5d} ds {lds:4d} vmem {vmem:4d} barrier {c.get('s_barrier', 0):3d}")
        print()


if __name__ == '__main__':
    main()
