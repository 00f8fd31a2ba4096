"""TEST INFRASTRUCTURE: compiles kernel sources of sherf_amd/csrc for the HOST against tests/hipcpu/hip/hip_runtime.h.
The only textual change is the declaration of dynamic shared memory (`extern __shared__ T name[];` -> a pointer to the
shim's buffer); everything else is the source as shipped."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'sherf_amd', 'csrc')


def build(name, sources, out_dir, extra_src=None):
    os.makedirs(out_dir, exist_ok=True)
    cpps = [os.path.join(HERE, 'runtime.cpp')]
    if extra_src:
        p = os.path.join(out_dir, 'extra.cpp')
        open(p, 'w').write(extra_src)
        cpps.append(p)
    for s in sources:
        src = open(os.path.join(CSRC, s)).read()
        src = re.sub(r'extern __shared__ (?:__attribute__\(\(aligned\(16\)\)\) )?(\w+) (\w+)\[\];', r'\1* \2 = reinterpret_cast<\1*>(hipcpu_dyn);', src)
        src = src.replace('#include "common.h"', f'#include "{os.path.join(CSRC, "common.h")}"')
        src = src.replace('#include "../../include/', f'#include "{os.path.join(ROOT, "include")}/')
        p = os.path.join(out_dir, s.replace('.hip', '.cpp'))
        open(p, 'w').write(src)
        cpps.append(p)
    lib = os.path.join(out_dir, f'lib{name}.so')
    cmd = ['g++', '-std=c++20', '-O1', '-pthread', '-shared', '-fPIC', '-w', f'-I{HERE}', f'-I{os.path.join(ROOT, "include")}', *cpps, '-o', lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('hipcpu build failed:\n' + r.stdout.decode()[-4000:])
    return lib
