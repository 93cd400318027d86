#!/usr/bin/env python3
"""ISA check of triplane_walk2_kernel's hand-issued tap loads (field_walk2.inc, TDGP_WALK2_ASMLOAD): between a buffer_load_dwordx4 into
v[a:b] and the hand-written `s_waitcnt vmcnt(3 FQ | 3 FQ + 3)` that covers it, NO instruction may read or write any of v[a:b] -- the compiler
believes those registers hold their values from the moment the asm statement ends.  Walks the producer loop body twice (the second lap
covers the window that crosses the back edge).  Also lists every compiler-inserted vmcnt wait inside the loop.
    python tools/dev/check_walk2_isa.py [FQ MT TAPS]     (compiles 3dgp_amd/csrc/field.hip to assembly with the build's flags)"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-fvisibility=hidden', '-Wno-unused-result']


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[:])v(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def main():
    fq, mt, taps = (sys.argv[1:4] + ['8', '4', '0'])[:3] if len(sys.argv) > 1 else ('8', '4', '0')
    name = f'triplane_walk2_kernelILi{fq}ELi{mt}ELb{taps}E'
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'f.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + sys.argv[4:] + ['-S', '--cuda-device-only', '-o', out, os.path.join(REPO, '3dgp_amd', 'csrc', 'field.hip')],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith('_ZN') and name in ln and ln.rstrip().endswith(':') is False and ':' in ln)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i] and i > start + 2000) if False else next(
        i for i in range(start, len(lines)) if lines[i].strip().startswith('.amdhsa_kernel'))
    body = lines[start:end]
    # the producer loop: from the header of the kernel's last outermost loop to the end of the kernel, read as one linear body
    heads = [i for i, ln in enumerate(body) if 'Loop Header: Depth=1' in ln]
    lo = heads[-1]
    ends = [i for i, ln in enumerate(body) if 's_endpgm' in ln and i > lo]
    hi = ends[0] if ends else len(body) - 1
    loop = [ln for ln in body[lo:hi + 1] if ln.strip() and not ln.strip().startswith(';') and not ln.startswith('.')]
    # the producer's prologue (first hand-issued request up to the loop) is read once in front of the two laps
    pro0 = next((i for i, ln in enumerate(body[:lo]) if 'global_load_dwordx3' in ln), lo)
    prologue = [ln for ln in body[pro0:lo] if ln.strip() and not ln.strip().startswith(';') and not ln.startswith('.')]
    inflight = []         # [(set of destination registers, text)] in issue order
    bad, comp_waits, hand_waits, nloads = [], [], 0, 0
    for lap in range(-1, 2):
        for ln in (prologue if lap < 0 else loop):
            ins = ln.split(';')[0].strip()
            op = ins.split()[0]
            if op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', ins)
                if not m:
                    continue
                n = int(m.group(1))
                if n in (3 * int(fq), 3 * int(fq) + 3) and 'lgkmcnt' not in ins:
                    hand_waits += lap == 0
                else:
                    comp_waits += [ins] if lap == 0 else []
                inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight       # all but the n youngest are complete
                continue
            if op.startswith('buffer_load_dwordx4') or op.startswith('global_load'):
                dst = regs(ins.split(',')[0])
                if any(dst & d for d, _ in inflight):
                    bad.append(('load into a register that is still in flight', ins))
                if any(regs(','.join(ins.split(',')[1:])) & d for d, _ in inflight):
                    bad.append(('address register is in flight', ins))
                nloads += lap == 0 and op.startswith('buffer_load')
                inflight.append((dst, ins))
                continue
            if op.startswith('s_'):
                continue
            touched = [t for d, t in inflight if regs(ins) & d]
            if touched:
                bad.append((f'touches the destination of `{touched[0]}`', ins))
    print(f'{name}: producer loop {len(loop)} instructions, {nloads} tap loads, {hand_waits} hand-written waits, compiler vmcnt waits in the loop: {comp_waits}')
    if bad:
        for why, ins in bad[:20]:
            print('  VIOLATION:', why, '|', ins)
        sys.exit(1)
    print('  ok: no instruction touches a tap register between its load and its wait')


if __name__ == '__main__':
    main()
