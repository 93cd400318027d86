#!/usr/bin/env python3
"""asm_mix.py <file.s> <kernel-substring> : instruction mix of every loop in a kernel of a hipcc -S dump"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(txt) if re.match(r'^_Z\S*' + re.escape(sys.argv[2]) + r'\S*:', l))
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
lines = txt[start:end]
labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_pk'): return 'valu_pk'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'): return 'vmem'
    return 'other'
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for a, b in loops:
    if b - a < minlen: continue
    cnt, ops = collections.Counter(), collections.Counter()
    for l in lines[a:b + 1]:
        l = l.strip()
        if not l or l[0] in ';.' or l.endswith(':'): continue
        op = l.split()[0]; cnt[cls(op)] += 1; ops[op] += 1
    print(a, b, dict(cnt)); print('   ', ops.most_common(30))
