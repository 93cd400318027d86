"""Build-time ISA check of triplane_walk2_kernel's hand-issued tap loads (csrc/field_walk2.inc, TDGP_WALK2_ASMLOAD).

Between a `buffer_load_dwordx4` into v[a:b] and the hand-written `s_waitcnt vmcnt(3 FQ | 3 FQ + 3)` that covers it, NO instruction may
read or write any of v[a:b]: the compiler believes those registers hold their values from the moment the asm statement ends.  A hipcc
upgrade or a flag change could break that silently; `build.build_native()` therefore compiles field.hip to assembly next to the object
and fails the build on a violation.  The producer's prologue is read once, its loop body twice (the second lap covers the
window that crosses the back edge).
"""
import re


class IsaListingError(RuntimeError):
    """The assembly listing does not have the shape the checker parses (not a violation of the property it checks)."""


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
    if not heads:
        raise IsaListingError("no 'Loop Header: Depth=1' comment in the listing of triplane_walk2_kernel: this hipcc does not annotate loops "
                              '(asm-verbose off, or a different compiler) -- the check cannot locate the producer loop')
    # the producer loop: the outermost loop that holds the hand-written tap-buffer waits (round 5: no longer "the last loop before s_endpgm" --
    # the bounded waits now end the wave in place, so s_endpgm also appears inside the loops)
    hand = [i for i, ln in enumerate(body) if re.search(r's_waitcnt vmcnt\((%d|%d)\)\s*$' % (3 * fq, 3 * fq + 3), ln.split(';')[0])]
    cand = [h for h in heads if hand and h < hand[-1]]
    lo = cand[-1] if cand else heads[-1]
    m = re.match(r'^\.L(BB\d+_\d+):', body[lo])
    label = m.group(1) if m else None
    members = [i for i, ln in enumerate(body) if label and re.search(r'Header=%s\b' % re.escape(label), ln)]
    if members:
        hi = next((j - 1 for j in range(members[-1] + 1, len(body)) if re.match(r'^\.LBB\d+_\d+:', body[j]) and not re.search(r'Header=%s\b' % re.escape(label), body[j])),
                  len(body) - 1)
    else:
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
    if nloads == 0 and hand_waits == 0:
        # the hand-issued path is compiled out (TDGP_WALK2_ASMLOAD=0 A/B builds): the compiler keeps its own wait counts, nothing to verify
        return dict(loop_instructions=len(loop), tap_loads=0, hand_waits=0, compiler_vmcnt_waits=comp_waits, skipped='no hand-issued loads'), []
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


# ------------------------------------------------------------------------------------------------------------------------------------
# conv3_wino4_kernel (csrc/modconv_wino4.inc): hand-issued LDS-direct loads (`s_mov_b32 m0, ..; s_nop 0; buffer_load_dwordx4 .. lds`) and
# the hand-counted `s_waitcnt vmcnt(4)` of the pair form's K loop.  Properties checked in the ISA of every instantiation:
#   1. m0 is written only by `s_mov_b32 m0, <sgpr>` and each such write is followed by `s_nop 0` and an LDS-direct buffer load -- the
#      compiler keeps nothing of its own in m0 (it cannot be told about the write: reserved registers are rejected on clobber lists);
#   2. in the K loop that ends in `vmcnt(4)`: every vector-memory instruction is an LDS-direct `buffer_load_dwordx4`, there are exactly
#      8 issue sites (3 of U, then 5 of V: the wait lets the <= 4 youngest -- V pieces of chunk c + 2 -- fly, LDS-direct loads complete
#      in issue order among themselves), the U sites precede the V sites, and `vmcnt(4)` is the loop's only vector-memory wait.
# ------------------------------------------------------------------------------------------------------------------------------------
def check_wino4_kernel(body):
    code = [(i, ln.split(';')[0].strip()) for i, ln in enumerate(body)]
    code = [(i, c) for i, c in code if c and not c.startswith('.') or re.match(r'^\.LBB\d+_\d+:', c or '')]
    bad = []
    ins_only = [(i, c) for i, c in code if not c.endswith(':')]
    # 1. m0
    m0_writes = 0
    for k, (i, c) in enumerate(ins_only):
        if not re.search(r'\bm0\b', c):
            continue
        if not re.match(r'^s_mov_b32 m0, s\d+$', c):
            bad.append(('m0 is touched by an instruction other than the hand-written `s_mov_b32 m0, <sgpr>`', c))
            continue
        m0_writes += 1
        nxt = [x for _, x in ins_only[k + 1:k + 3]]
        if len(nxt) < 2 or nxt[0] != 's_nop 0' or not (nxt[1].startswith('buffer_load_dwordx4') and nxt[1].endswith(' lds')):
            bad.append(('`s_mov_b32 m0` is not followed by `s_nop 0` + an LDS-direct buffer load', c + ' | ' + ' | '.join(nxt)))
    # 2. the pair form's K loop
    waits = [k for k, (_, c) in enumerate(code) if re.match(r'^s_waitcnt vmcnt\(4\)$', c)]
    summary = dict(m0_writes=m0_writes, pair_loops=len(waits))
    k_loops = 0
    for w in waits:
        lab = next((code[k][1][:-1] for k in range(w, -1, -1) if code[k][1].endswith(':')), None)
        back = [k for k in range(w, len(code)) if re.match(r'^s_c?branch\w* ' + re.escape(lab or '?') + '$', code[k][1])]
        if lab is None or not back:
            continue                                # (a compiler-placed vmcnt(4) in straight-line code -- e.g. the output stage's noise loads: not the K loop)
        k0 = next(k for k in range(w, -1, -1) if code[k][1] == lab + ':')
        loop = [c for _, c in code[k0:back[-1] + 1] if not c.endswith(':')]
        if not any(c.startswith('v_mfma') for c in loop):
            continue                                # a loop without matrix instructions is not the K loop either
        k_loops += 1
        vmem = [c for c in loop if re.match(r'^(buffer_|global_|flat_|scratch_)', c)]
        notlds = [c for c in vmem if not (c.startswith('buffer_load_dwordx4') and c.endswith(' lds'))]
        for c in notlds:
            bad.append(('a vector-memory instruction other than an LDS-direct load inside the K loop (the hand-counted vmcnt(4) does not know it)', c))
        if len(vmem) - len(notlds) != 8:
            bad.append((f'expected 8 LDS-direct issue sites (3 U + 5 V) in the K loop, found {len(vmem) - len(notlds)}', ''))
        descs = [re.search(r'(s\[\d+:\d+\])', c).group(1) for c in vmem if c not in notlds]
        if len(set(descs)) != 2 or descs != [descs[0]] * descs.count(descs[0]) + [descs[-1]] * descs.count(descs[-1]) or descs.count(descs[0]) != 3:
            bad.append(('the U loads (3 sites, one descriptor) must be issued before the V loads (5 sites, the other descriptor)', ' '.join(descs)))
        vwaits = [c for c in loop if c.startswith('s_waitcnt') and 'vmcnt' in c]
        if vwaits != ['s_waitcnt vmcnt(4)']:
            bad.append(('the K loop must hold exactly one vector-memory wait, the hand-written vmcnt(4)', ' | '.join(vwaits)))
        summary.update(loop_instructions=len(loop), lds_direct_sites=len(vmem) - len(notlds), mfma=sum(c.startswith('v_mfma') for c in loop))
    summary['pair_loops'] = k_loops
    return summary, bad


def check_wino4_asm(asm_text):
    """Every conv3_wino4_kernel instantiation in a modconv.hip assembly listing.  -> {mangled name: (summary, violations)}"""
    lines = asm_text.splitlines()
    out = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(_ZN\S*conv3_wino4_kernelILb([01])ELb([01])E\S*):', ln)
        if not m:
            continue
        end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel')), len(lines))
        out[m.group(1)] = check_wino4_kernel(lines[i:end])
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
# conv3_wino4f_kernel (csrc/modconv_wino4f.inc, round 6): EVERY vector-memory read of the item loop is hand-issued -- U pieces, window rows, halo quads,
# styles and noise patch as LDS-direct loads, the ticket as a returning atomic whose result register is only read after the K loop --, and the only waits are
# the hand-written `s_waitcnt vmcnt(0)` (item top, chunk ends).  The compiler's own wait counts do not know any of them, so the properties are checked here:
#   1. m0 only by `s_mov_b32 m0, <sgpr>` + `s_nop 0` + an LDS-direct buffer load (dwordx4 or dword);
#   2. no scratch access anywhere (a spill reload is a vector-memory load + `vmcnt(0)`: it would wait for the requests in flight);
#   3. the K loop (the loop holding the 144 MFMAs): its vector-memory instructions are LDS-direct loads only, its vector-memory waits `vmcnt(0)` only;
#   4. the destination of a `global_atomic_add` is not read before the next `s_waitcnt vmcnt(0)`;
#   5. from the K loop's last MFMA to the item-end barrier the only vector-memory loads are LDS-direct ones and the only `vmcnt` waits those that follow an
#      atomic (the ticket retry path) -- the output stage waits for nothing the next item requested.
# ------------------------------------------------------------------------------------------------------------------------------------
def check_wino4f_kernel(body):
    code = [(i, ln.split(';')[0].strip()) for i, ln in enumerate(body)]
    code = [c for _, c in code if c and (not c.startswith('.') or re.match(r'^\.LBB\d+_\d+:', c))]
    ins = [c for c in code if not c.endswith(':')]
    bad = []
    lds_direct = lambda c: re.match(r'^buffer_load_dword(x4)? ', c) and c.endswith(' lds')      # noqa: E731
    m0_writes = 0
    for k, c in enumerate(ins):
        if not re.search(r'\bm0\b', c):
            continue
        if not re.match(r'^s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)$', c):
            bad.append(('m0 is touched by an instruction other than the hand-written `s_mov_b32 m0, <sgpr>`', c))
            continue
        m0_writes += 1
        nxt = ins[k + 1:k + 3]
        if len(nxt) < 2 or nxt[0] != 's_nop 0' or not lds_direct(nxt[1]):
            bad.append(('`s_mov_b32 m0` is not followed by `s_nop 0` + an LDS-direct buffer load', c + ' | ' + ' | '.join(nxt)))
    for c in ins:
        if c.startswith('scratch_'):
            bad.append(('scratch access (a spill: its reload waits with vmcnt(0) for every LDS-direct load in flight)', c))
            break
    mf = [k for k, c in enumerate(code) if c.startswith('v_mfma')]
    if len(mf) != 144:
        raise IsaListingError(f'conv3_wino4f_kernel: expected 144 MFMAs (4 chunks x 36), found {len(mf)}')
    # the K loop: the nearest label in front of the first MFMA that a branch behind the last MFMA jumps back to
    k0 = back = None
    for k in range(mf[0], -1, -1):
        if code[k].endswith(':'):
            bk = [j for j in range(mf[-1], len(code)) if re.match(r'^s_c?branch\w* ' + re.escape(code[k][:-1]) + '$', code[j])]
            if bk:
                k0, back = k, bk[0]
                break
    if k0 is None:
        raise IsaListingError('conv3_wino4f_kernel: K loop not found (no back edge behind the last MFMA to a label in front of the first)')
    loop = [c for c in code[k0:back + 1] if not c.endswith(':')]
    vmem = [c for c in loop if re.match(r'^(buffer_|global_|flat_|scratch_)', c)]
    for c in vmem:
        if not lds_direct(c):
            bad.append(('a vector-memory instruction other than an LDS-direct load inside the K loop', c))
    for c in loop:
        if c.startswith('s_waitcnt') and 'vmcnt' in c and c != 's_waitcnt vmcnt(0)':
            bad.append(('a vector-memory wait other than the hand-written vmcnt(0) inside the K loop', c))
    # 4. atomics
    atomics = 0
    for k, c in enumerate(ins):
        m = re.match(r'^global_atomic_add (v\d+),', c)
        if not m:
            continue
        atomics += 1
        for c2 in ins[k + 1:]:
            if c2 == 's_waitcnt vmcnt(0)':
                break
            if re.search(r'\b' + m.group(1) + r'\b', c2):
                bad.append((f'{m.group(1)}, the destination of a returning atomic, is touched before the next vmcnt(0)', c2))
                break
    # 5. behind the K loop, up to the item loop's closing branch.  Not scanned: the block a branch IN FRONT of the K loop jumps to behind it -- the
    #    zero-chunk path around the loop (never taken: Cin >= 16), where the compiler may wait for the bias load it placed in front of the loop
    item_back = next((k for k in range(len(code) - 1, back, -1) if re.match(r'^s_c?branch\w* \.LBB\d+_\d+$', code[k])), len(code) - 1)
    around = {code[k].split()[-1] for k in range(k0) if re.match(r'^s_c?branch\w* \.LBB\d+_\d+$', code[k])}
    tail, skip = [], False
    for c in code[back + 1:item_back]:
        if c.endswith(':'):
            skip = c[:-1] in around
            continue
        if not skip:
            tail.append(c)
    for k, c in enumerate(tail):
        if re.match(r'^(buffer_load|global_load|flat_load)', c) and not lds_direct(c):
            bad.append(('a vector-memory load of the compiler\'s behind the K loop (its wait would also wait for the next item\'s requests)', c))
        if c.startswith('s_waitcnt') and 'vmcnt' in c:
            prev = tail[max(0, k - 4):k]
            if not any(x.startswith('global_atomic_add') for x in prev):
                bad.append(('a vector-memory wait behind the K loop that does not belong to the ticket retry path', c))
    return dict(m0_writes=m0_writes, mfma=len(mf), loop_instructions=len(loop), lds_direct_sites=len(vmem), atomics=atomics, stores_behind_loop=sum(c.startswith('global_store') for c in tail)), bad


def check_wino4f_asm(asm_text):
    """conv3_wino4f_kernel in a modconv.hip assembly listing.  -> {mangled name: (summary, violations)}"""
    lines = asm_text.splitlines()
    out = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(_Z\S*conv3_wino4f_kernel\S*):', ln)
        if not m:
            continue
        end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel')), len(lines))
        out[m.group(1)] = check_wino4f_kernel(lines[i:end])
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
# torgb_mfma_kernel<.., FAST = true> / torgb_ws_kernel (csrc/modconv.hip: rgb_output_skip_pipelined, round 6): the skip taps are hand-issued `global_load_dwordx4`
# in batches of 4 with A batches of look-ahead (A = 2 .. 5), the waits hand-counted `vmcnt(4 A)` (.. fewer for the last A batches, `vmcnt(0)` for the last): the count
# holds only if, in program order, every batch is exactly 4 tap loads + one store and NOTHING else of vector memory stands between the waits of a stage (a compiler-side
# load -- e.g. the FIR weight fetched by a per-lane index -- would be counted by the hardware and not by the hand).  Checked per kernel, per stage instance (the
# batches up to a hand-written `vmcnt(0)`): wait counts = 4 x min(A, batches left); four hand loads per wait, interleaved as issued; one `global_store_dwordx4` per
# batch; no other buffer_ / global_ / flat_ / scratch_ instruction inside; the registers of a load in flight untouched before its wait; no scratch access in the kernel.
# ------------------------------------------------------------------------------------------------------------------------------------
def check_torgb_kernel(body, mt):
    # (hand-written statements stand between `;;#ASMSTART` / `;;#ASMEND` in the listing: a compiler-placed `vmcnt(8)` elsewhere in the kernel is not one of the counted waits)
    code, hand, in_asm = [], [], False
    for ln in body:
        t = ln.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        c = ln.split(';')[0].strip()
        if c and not c.startswith('.') and not c.endswith(':'):
            code.append(c)
            hand.append(in_asm)
    bad = []
    if any(c.startswith('scratch_') for c in code):
        bad.append(('scratch access in a ToRGB kernel with hand-counted tap waits', next(c for c in code if c.startswith('scratch_'))))
    hw = [k for k, c in enumerate(code) if hand[k] and re.match(r'^s_waitcnt vmcnt\(\d+\)$', c)]
    hl = [k for k, c in enumerate(code) if hand[k] and c.startswith('global_load_dwordx4')]
    # one instance of the stage = the batches up to a hand-written `vmcnt(0)` (the one-role kernels hold one of 4 MT batches; the two-role kernel four: the tile loop's
    # and the last tile's, each shared between the memory waves and the multiplying waves)
    ends = [n for n, k in enumerate(hw) if code[k] == 's_waitcnt vmcnt(0)']
    if not hw or not ends or ends[-1] != len(hw) - 1 or len(hl) != 4 * len(hw) or len(hw) % (4 * mt):
        bad.append((f'expected stages of hand-counted waits ending in vmcnt(0), 4 MT batches per tile and four tap loads per wait; found {len(hw)} waits, {len(hl)} loads', ''))
        return dict(batches=len(hw), tap_loads=len(hl), hand_waits=len(hw)), bad
    bounds = [0] + [n + 1 for n in ends]
    stages = len(ends)
    for st in range(stages):
        nb = bounds[st + 1] - bounds[st]
        w, ld_ = hw[bounds[st]:bounds[st + 1]], hl[4 * bounds[st]:4 * bounds[st + 1]]
        counts = [int(re.search(r'\((\d+)\)', code[k]).group(1)) for k in w]
        ahead = counts[0] // 4
        want = [4 * min(ahead, nb - 1 - bi) for bi in range(nb)]
        if counts != want or counts[0] % 4 or not (1 <= ahead <= 5 or (ahead == 0 and nb == 1)):
            bad.append((f'wait counts of a stage must be 4 x min(look-ahead, batches left): {want}', str(counts)))
            continue
        if not (ld_[0] < w[0] and all(ld_[4 * (bi + ahead)] > w[bi - 1] for bi in range(1, nb - ahead)) and ld_[-1] < w[nb - ahead - 1 if nb > ahead else 0] + 10 ** 9):
            bad.append(('tap loads and waits of a stage are not interleaved as issued', ''))
        # between the stage's first tap load and its last wait: the hand-issued loads, one compiler store per batch, nothing else of vector memory
        seg = [(k, c) for k, c in enumerate(code[ld_[0]:w[-1] + 1], ld_[0]) if re.match(r'^(buffer_|global_|flat_|scratch_)', c)]
        other = [c for k, c in seg if not (hand[k] and c.startswith('global_load_dwordx4')) and not c.startswith('global_store_dwordx4')]
        if other:
            bad.append(('a vector-memory instruction of the compiler\'s between the hand-counted waits of a stage (the hardware counts it, the hand does not)', other[0]))
        for bi in range(nb - 1):
            st_ = [c for c in code[w[bi]:w[bi + 1]] if c.startswith('global_store_dwordx4')]
            if len(st_) != 1:
                bad.append((f'{len(st_)} stores between two hand-counted waits (one per batch)', ''))
                break
        # the registers a tap load of batch i writes are touched by nothing before the i-th wait of the stage
        for n, k in enumerate(ld_):
            m = re.match(r'^global_load_dwordx4 v\[(\d+):(\d+)\]', code[k])
            if not m:
                bad.append(('a hand-issued tap load the checker cannot read', code[k]))
                break
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
            hit = None
            for c in code[k + 1:w[n // 4]]:
                ops = c.split(None, 1)[1] if ' ' in c else ''
                used = set()
                for tok in re.findall(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', ops):
                    used |= set(range(int(tok[0]), int(tok[1]) + 1)) if tok[0] else {int(tok[2])}
                if used & regs:
                    hit = c
                    break
            if hit:
                bad.append((f'registers of a tap load in flight are touched before its wait ({code[k].split(",")[0]})', hit))
                break
    return dict(batches=len(hw), tap_loads=len(hl), hand_waits=len(hw), stages=stages), bad


def check_torgb_asm(asm_text):
    """Every FAST fp32 instantiation of torgb_mfma_kernel, and torgb_ws_kernel, in a modconv.hip listing.  -> {mangled name: (summary, violations)}"""
    lines = asm_text.splitlines()
    out = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(_ZN\S*torgb_mfma_kernelILi(\d)ELb[01]ELb1ELb0E\S*):', ln) or re.match(r'^(_ZN\S*torgb_ws_kernelILi(\d)E\S*):', ln)
        if not m:
            continue
        end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith('.amdhsa_kernel')), len(lines))
        out[m.group(1)] = check_torgb_kernel(lines[i:end], int(m.group(2)))
    return out
