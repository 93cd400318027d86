"""Build-time ISA check of triplane_walk2_kernel's hand-issued tap loads (csrc/field_walk2.inc, TDGP_WALK2_ASMLOAD).

Between a `buffer_load_dwordx4` into v[a:b] and the hand-written `s_waitcnt vmcnt(3 FQ | 3 FQ + 3)` that covers it, NO instruction may
read or write any of v[a:b]: the compiler believes those registers hold their values from the moment the asm statement ends.  A hipcc
upgrade or a flag change could break that silently; `build.build_native()` therefore compiles field.hip to assembly next to the object
and fails the build on a violation.  The producer's prologue is read once, its loop body twice (the second lap covers the
window that crosses the back edge).
"""
import re


def _regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[:])v(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def _code(lines):
    return [ln for ln in lines if ln.strip() and not ln.strip().startswith(';') and not ln.startswith('.')]


def check_kernel(body, fq):
    """`body` = assembly lines of ONE walk2 instantiation.  -> (summary dict, [(why, instruction)])"""
    heads = [i for i, ln in enumerate(body) if 'Loop Header: Depth=1' in ln]
    lo = heads[-1]                  # the producer loop: the kernel's last outermost loop
    ends = [i for i, ln in enumerate(body) if 's_endpgm' in ln and i > lo]
    hi = ends[0] if ends else len(body) - 1
    loop = _code(body[lo:hi + 1])
    pro0 = next((i for i, ln in enumerate(body[:lo]) if 'global_load_dwordx3' in ln), lo)
    prologue = _code(body[pro0:lo])
    inflight, bad, comp_waits, hand_waits, nloads = [], [], [], 0, 0
    for lap in range(-1, 2):
        for ln in (prologue if lap < 0 else loop):
            ins = ln.split(';')[0].strip()
            op = ins.split()[0]
            if op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', ins)
                if not m:
                    continue
                n = int(m.group(1))
                if n in (3 * fq, 3 * fq + 3) and 'lgkmcnt' not in ins:
                    hand_waits += lap == 0
                elif lap == 0:
                    comp_waits.append(ins)
                inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight       # all but the n youngest are complete
                continue
            if op.startswith('buffer_load_dwordx4') or op.startswith('global_load'):
                dst = _regs(ins.split(',')[0])
                if any(dst & d for d, _ in inflight):
                    bad.append(('load into a register that is still in flight', ins))
                if any(_regs(','.join(ins.split(',')[1:])) & d for d, _ in inflight):
                    bad.append(('address register is in flight', ins))
                nloads += lap == 0 and op.startswith('buffer_load')
                inflight.append((dst, ins))
                continue
            if op.startswith('s_'):
                continue
            touched = [t for d, t in inflight if _regs(ins) & d]
            if touched:
                bad.append((f'touches the destination of `{touched[0]}`', ins))
    if nloads != 6 * fq or hand_waits != 2:
        bad.append((f'expected {6 * fq} tap loads and 2 hand-written waits in the producer loop, found {nloads} / {hand_waits}', ''))
    return dict(loop_instructions=len(loop), tap_loads=nloads, hand_waits=hand_waits, compiler_vmcnt_waits=comp_waits), bad


def check_walk2_asm(asm_text):
    """Every triplane_walk2_kernel instantiation in a field.hip assembly listing.  -> {mangled name: (summary, violations)}"""
    lines = asm_text.splitlines()
    out = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(_ZN\S*triplane_walk2_kernelILi(\d+)ELi(\d+)ELb([01])E\S*):', ln)
        if not m:
            continue
        end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel')), len(lines))
        out[m.group(1)] = check_kernel(lines[i:end], int(m.group(2)))
    return out
