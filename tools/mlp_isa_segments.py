"""Static instruction mix of a kernel's ISA, cut at its workgroup barriers (one segment per weight-stream step of nerf_mlp).

    python tools/mlp_isa_segments.py <file.s> <kernel-name-substring>
"""
import collections
import re
import sys


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_'):
        if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', op):
            return 'trans'
        if op.startswith('v_pk_'):
            return 'vpk'
        if op.startswith('v_cvt'):
            return 'vcvt'
        if op.startswith('v_accvgpr'):
            return 'acc'
        return 'valu'
    if op.startswith('ds_'):
        return 'ds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem'
    if op == 's_nop':
        return 'nop'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op == 's_barrier':
        return 'bar'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l.split(':')[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    seg, segs = collections.Counter(), []
    for l in lines[start + 1:end]:
        l = l.strip()
        if not l or l.startswith((';', '.')) or l.endswith(':'):
            continue
        c = cls(l.split()[0])
        seg[c] += 1
        if c == 'bar':
            segs.append(seg)
            seg = collections.Counter()
    segs.append(seg)
    keys = ['mfma', 'valu', 'vcvt', 'vpk', 'trans', 'acc', 'ds', 'vmem', 'salu', 'nop', 'wait']
    print(len(segs), 'segments')
    print('seg ' + ' '.join(f'{k:>5}' for k in keys))
    tot = collections.Counter()
    for i, g in enumerate(segs):
        print(f'{i:3d} ' + ' '.join(f'{g[k]:5d}' for k in keys))
        tot += g
    print('tot ' + ' '.join(f'{tot[k]:5d}' for k in keys))


if __name__ == '__main__':
    main()
