"""Static audit of the MFMA data hazards gfx950 leaves to software, on the compiler's final assembly of a kernel.

hipcc's hazard recognizer inserts the `s_nop`s itself for MFMAs it can see; an MFMA inside an `asm volatile` block (csrc/mlp.hip:
mfma_block) is opaque to it, so every wait state around those is hand-placed -- and a register copy or spill the allocator drops
between such a block and the hand-placed settle is NOT covered by anything.  This tool replays the rules on the final ISA:

  R1  MFMA writes v[..]  ->  any non-MFMA instruction touching one of those registers      needs >= 12 wait states
  R2  MFMA writes v[..]  ->  MFMA reading one of them as SrcA / SrcB                       needs >= 12
  R3  MFMA writes v[..]  ->  MFMA whose SrcC overlaps them but is not the SAME tuple          needs >= 12   (the same tuple as SrcC:
      the back-to-back accumulate the hardware forwards, 0)
  R4  VALU writes vN     ->  MFMA reading vN (any source)                                  needs >= 2
  R5  an MFMA inside an EXEC-predicated region that has NO skip branch: `s_and_saveexec` ... `v_mfma` with no `s_cbranch_execz/execnz` in
      between.  MFMAs ignore EXEC on gfx950, so with EXEC = 0 the instruction still runs on whatever its operand registers hold; hipcc drops
      the skip branch of SHORT predicated blocks (one MFMA) -- round 3's single-product sparse convolutions failed on the hardware this way
      (csrc/svox.hip: `wave` had to be made provably uniform).  Reported per kernel; not a wait-state rule.

The constants are the compiler's own for v_mfma_f32_32x32x16_{f16,bf16} on gfx950 (8 passes): `s_nop 11` between such an MFMA and a
VALU / VMEM reader of its result, `s_nop 1` between a VALU write and the MFMA reading it (probe: tools/mfma_hazard_probe.hip, compiled
with the same hipcc).  A wait state = one issued instruction; `s_nop N` = N + 1.  The scan is linear over the kernel's text (branch
targets do not reset the window: conservative for the straight-line MLP kernel; for kernels with control flow R1-R4 over-report across
branches -- the sparse convolutions' MFMAs are builtins anyway, hipcc places their wait states -- while R5 is exact for them).

    python tools/mfma_hazard_check.py <file.s> [kernel-name-substring ...]      exit code 1 if any kernel has a violation
"""
import re
import sys

MFMA_TO_OTHER = 12
VALU_TO_MFMA = 2
REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])')


def regs(text):
    out = []
    for m in REG.finditer(text):
        if m.group(2) is not None:
            out.append({(m.group(1), int(m.group(2)))})
        else:
            out.append({(m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1)})
    return out


def kernels(lines):
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith('_Z') and ':' in l:
            name = l.split(':')[0]
            j = i + 1
            while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
                j += 1
            yield name, i + 1, j
            i = j
        i += 1


def check(lines, start, end, verbose=True):
    masked = None       # line of the last s_and_saveexec with no skip branch / restore behind it yet
    pending = []        # [regset, age in wait states, line no, text] per MFMA still inside the window
    valu = {}           # reg -> age of the last VALU write
    bad = []
    n_mfma = 0
    for ln in range(start, end):
        raw = lines[ln]
        t = raw.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        args = t[len(op):]
        ops = regs(args)
        is_mfma = op.startswith('v_mfma') or op.startswith('v_smfma')
        if op.startswith('s_and_saveexec') or op.startswith('s_andn2_saveexec') or op.startswith('s_or_saveexec'):
            masked = ln + 1
        elif op.startswith('s_cbranch_exec') or (op.startswith(('s_or_b64', 's_mov_b64', 's_xor_b64', 's_andn2_b64')) and args.strip().startswith('exec')):
            masked = None
        if is_mfma and masked is not None:
            bad.append((ln + 1, 'R5', 0, masked, t))
        if is_mfma:
            n_mfma += 1
            parts = [p.strip() for p in args.split(',')]
            dst = regs(parts[0])[0]
            srca = set().union(*regs(parts[1])) if regs(parts[1]) else set()
            srcb = set().union(*regs(parts[2])) if regs(parts[2]) else set()
            srcc = set().union(*regs(parts[3])) if len(parts) > 3 and regs(parts[3]) else set()
            for p in pending:
                if p[1] >= MFMA_TO_OTHER:
                    continue
                if (srca | srcb) & p[0]:
                    bad.append((ln + 1, 'R2', p[1], p[2], t))
                if srcc & p[0] and srcc != p[0]:             # (an overlapping DESTINATION is fine: the matrix pipe retires in order)
                    bad.append((ln + 1, 'R3', p[1], p[2], t))
            for r in srca | srcb | srcc:
                if r in valu and valu[r] < VALU_TO_MFMA:
                    bad.append((ln + 1, 'R4', valu[r], None, t))
        else:
            touched = set().union(*ops) if ops else set()
            for p in pending:
                if p[1] < MFMA_TO_OTHER and touched & p[0]:
                    bad.append((ln + 1, 'R1', p[1], p[2], t))
        # age everything by this instruction's wait states
        ws = 1
        if op == 's_nop':
            ws = int(args.strip()) + 1
        for p in pending:
            p[1] += ws
        pending = [p for p in pending if p[1] < MFMA_TO_OTHER + 4]
        for r in list(valu):
            valu[r] += ws
            if valu[r] > 8:
                del valu[r]
        if is_mfma:
            # the matrix pipe retires in order: a later MFMA's destination supersedes the overlapping part of an older one's (a reader of
            # those registers depends on the LATER writer -- hipcc's own recognizer looks at the last writer too; round 5: the two-tile
            # kernel's compiler-scheduled transformer re-uses half of a finished accumulator tuple as part of the next one)
            for p in pending:
                p[0] = p[0] - dst
            pending = [p for p in pending if p[0]]
            pending.append([set(dst), 0, ln + 1, t])
        elif op.startswith('v_') and ops and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
            for r in ops[0]:
                valu[r] = 0
            if op.startswith(('v_permlane32_swap', 'v_permlane16_swap', 'v_swap')) and len(ops) > 1:
                for r in ops[1]:
                    valu[r] = 0
    return n_mfma, bad


def main():
    path = sys.argv[1]
    want = sys.argv[2:]
    lines = open(path).read().splitlines()
    rc = 0
    for name, a, b in kernels(lines):
        if want and not any(w in name for w in want):
            continue
        n, bad = check(lines, a, b)
        if n == 0:
            continue
        print(f'{name}: {n} MFMAs, {len(bad)} hazard violations')
        for ln, rule, age, src_ln, text in bad[:40]:
            if rule == 'R5':
                print(f'   line {ln}: R5 MFMA predicated by the s_*_saveexec at line {src_ln} without a skip branch: {text}')
                continue
            print(f'   line {ln}: {rule} after {age} wait states' + (f' (MFMA at line {src_ln})' if src_ln else '') + f': {text}')
        if bad:
            rc = 1
    return rc


if __name__ == '__main__':
    sys.exit(main())
