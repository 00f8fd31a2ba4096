"""TEST INFRASTRUCTURE: compiles kernel sources of sherf_amd/csrc for the HOST against tests/hipcpu/hip/hip_runtime.h.
The only textual change is the declaration of dynamic shared memory (`extern __shared__ T name[];` -> a pointer to the
shim's buffer); everything else is the source as shipped."""
import hashlib
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'sherf_amd', 'csrc')


CLANG = '/opt/rocm/lib/llvm/bin/clang++'     # host compile of the sources that use __bf16 / ext_vector_type (MFMA kernels)


def _cpu_mlp(src):
    """mlp.hip for the host: the inline-asm sites (LDS-DMA, vmcnt wait, barrier, the ordered MFMA blocks and their settle nops)
    become their functional equivalents."""
    def cut(text, start, end, repl):
        a = text.index(start)
        b = text.index(end, a)
        return text[:a] + repl + text[b:]
    # LDS-DMA of one piece: what global_load_lds_dwordx4 does for this lane (the ring address is kept as an offset on the host)
    src = cut(src, '        uint32_t keep;\n        asm volatile("s_mov_b32 %0, m0', '    }\n}', '        memcpy(const_cast<char*>(cx.lds) + l, g, 16);\n')
    src = src.replace('cx.lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lptr_t)lds) + cx.wave * 1024;', 'cx.lds_addr = cx.wave * 1024;')
    src = cut(src, '__device__ __forceinline__ void wait_vm(int n) {', '__device__ __forceinline__ void wg_barrier()',
              '__device__ __forceinline__ void wait_vm(int) {}\n')
    src = re.sub(r'__device__ __forceinline__ void wg_barrier\(\) \{[^\n]*\}', '__device__ __forceinline__ void wg_barrier() { __syncthreads(); }', src)
    # the six / two MFMAs of a block, in the asm's order
    src = cut(src, '    if constexpr (PREC == 1)\n        asm volatile("s_nop 1', '}\n// an MFMA\'s result -> any reader',
              '    if constexpr (PREC == 1) {\n'
              '        acc0 = mfma<PREC>(al0, bh0, acc0); acc1 = mfma<PREC>(al1, bh1, acc1);\n'
              '        acc0 = mfma<PREC>(ah0, bl0, acc0); acc1 = mfma<PREC>(ah1, bl1, acc1);\n'
              '        acc0 = mfma<PREC>(ah0, bh0, acc0); acc1 = mfma<PREC>(ah1, bh1, acc1);\n'
              '    } else { acc0 = mfma<PREC>(ah0, bh0, acc0); acc1 = mfma<PREC>(ah1, bh1, acc1); }\n')
    src = src.replace('    asm volatile("s_nop 7\\n\\ts_nop 3" : "+v"(acc0), "+v"(acc1));', '    (void)acc0; (void)acc1;')
    # the two-tile kernel's block of four MFMAs (chain (t, u) <- a_u x b_tu, in the asm's order) and its settle
    src = cut(src, '    static_assert(PREC != 1, "two tiles per wave: single-product precisions only");', '}\n__device__ __forceinline__ void mfma_settle4',
              '    t0c0 = mfma<PREC>(a0, b00, t0c0); t1c0 = mfma<PREC>(a0, b10, t1c0);\n'
              '    t0c1 = mfma<PREC>(a1, b01, t0c1); t1c1 = mfma<PREC>(a1, b11, t1c1);\n')
    src = src.replace('    asm volatile("s_nop 7\\n\\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));', '    (void)a; (void)b; (void)c; (void)d;')
    src = src.replace('#define SHERF_MLP_FMA_MIX 1', '#define SHERF_MLP_FMA_MIX 0')       # v_fma_mix_f32 inline asm -> its plain-C equivalent
    src = re.sub(r'asm volatile\("" : "\+[sv]"\([^;]*;', ';', src)            # register-class launders (optimisation barriers only)
    assert 'asm volatile' not in src.replace('#if SHERF_MLP_FMA_MIX', '#if 0'), 'an inline-asm site of mlp.hip has no host equivalent'
    return src.replace('typedef __attribute__((address_space(3))) void* lptr_t;', 'typedef void* lptr_t;')


CACHE = os.environ.get('HIPCPU_CACHE', os.path.join(HERE, '_cache'))      # content-addressed builds, shared by every test module,
                                                                            # xdist worker and child process (git-ignored)


def _cache_key(name, sources, extra_src, compiler, defines):
    h = hashlib.sha1(repr((name, tuple(sources), compiler, tuple(defines), os.environ.get('HIPCPU_OPT', '-O2'))).encode())
    deps = [os.path.join(HERE, 'runtime.cpp'), os.path.join(HERE, 'build_cpu.py')]
    for d in (os.path.join(HERE, 'hip'), os.path.join(HERE, 'rocblas'), os.path.join(ROOT, 'include')):
        deps += [os.path.join(d, f) for f in sorted(os.listdir(d))]
    deps += [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')] + [os.path.join(CSRC, s) for s in sources]
    for p in deps:
        h.update(open(p, 'rb').read())
    h.update((extra_src or '').encode())
    return h.hexdigest()[:20]


def build(name, sources, out_dir, extra_src=None, compiler='g++', defines=()):
    """-> path of lib<name>.so.  Identical inputs (sources, headers, shim, flags) are built once: the result is kept under CACHE."""
    key = _cache_key(name, sources, extra_src, compiler, defines)
    cached = os.path.join(CACHE, key, f'lib{name}.so')
    if os.path.exists(cached):
        return cached
    lib = _build(name, sources, out_dir, extra_src, compiler, defines)
    os.makedirs(os.path.join(CACHE, key), exist_ok=True)
    tmp = cached + f'.{os.getpid()}.tmp'
    shutil.copyfile(lib, tmp)
    os.replace(tmp, cached)              # atomic: concurrent builders of the same key write the same bytes
    return cached


def _build(name, sources, out_dir, extra_src, compiler, defines):
    os.makedirs(out_dir, exist_ok=True)
    cpps = [os.path.join(HERE, 'runtime.cpp')]
    if extra_src:
        p = os.path.join(out_dir, 'extra.cpp')
        open(p, 'w').write(extra_src)
        cpps.append(p)
    for s in sources:
        src = open(os.path.join(CSRC, s)).read()
        src = re.sub(r'extern __shared__ (?:__attribute__\(\(aligned\(16\)\)\) )?(\w+) (\w+)\[\];', r'\1* \2 = reinterpret_cast<\1*>(hipcpu_dyn);', src)
        if s == 'mlp.hip':
            src = _cpu_mlp(src)
        src = src.replace('#include "common.h"', f'#include "{os.path.join(CSRC, "common.h")}"')
        src = src.replace('#include "ops_common.h"', f'#include "{os.path.join(CSRC, "ops_common.h")}"')
        src = src.replace('#include "../../include/', f'#include "{os.path.join(ROOT, "include")}/')
        p = os.path.join(out_dir, s.replace('.hip', '.cpp'))
        open(p, 'w').write(src)
        cpps.append(p)
    lib = os.path.join(out_dir, f'lib{name}.so')
    flags = [*[f'-D{d}' for d in defines], '-std=c++20', os.environ.get('HIPCPU_OPT', '-O2'), '-pthread', '-fPIC', '-w', f'-I{HERE}',
             f'-I{os.path.join(ROOT, "include")}']

    def compile_one(cpp):
        obj = os.path.join(out_dir, os.path.basename(cpp) + '.o')
        r = subprocess.run([compiler, *flags, '-c', cpp, '-o', obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError('hipcpu build failed:\n' + r.stdout.decode()[-4000:])
        return obj
    with ThreadPoolExecutor(max_workers=min(len(cpps), os.cpu_count() or 1)) as ex:      # one compiler process per translation unit
        objs = list(ex.map(compile_one, cpps))
    r = subprocess.run([compiler, '-shared', '-pthread', *objs, '-o', lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('hipcpu link failed:\n' + r.stdout.decode()[-4000:])
    return lib
