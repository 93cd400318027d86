#!/bin/bash
# device assembly of modconv.hip into /tmp/modconv.s + the fused F(4x4) kernel's resources and its vector-memory waits in front of the K loop
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-result -S --cuda-device-only -o /tmp/modconv.s /root/repo/3dgp_amd/csrc/modconv.hip "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|wino4f.inc:[0-9]+:[0-9]+: remark:     (VGPRs|Scratch|VGPRs Spill)" | head
python3 - <<'PY'
import re
s=open('/tmp/modconv.s').read()
i=s.index('conv3_wino4f_kernel')
i=s.index('conv3_wino4f_kernel',i+10)
e=s.index('.Lfunc_end',i)
k=s[i:e].split('\n')
m=[n for n,l in enumerate(k) if 'v_mfma' in l]
print('lines',len(k),'mfma',m[0],m[-1])
for n,l in enumerate(k):
    if re.search(r'scratch_|global_load|s_waitcnt vmcnt|s_barrier', l) and n < m[0]+50: print(n,l.strip()[:90])
for a in [n for n,l in enumerate(k) if 'global_atomic_add' in l]:
    print(a, k[a].strip())
    reg=re.search(r'global_atomic_add (v\d+)',k[a]).group(1)
    for n in range(a+1,len(k)):
        if 's_waitcnt vmcnt(0)' in k[n]:
            print('  next vmcnt(0) at',n, '(+%d)'%(n-a)); break
        if re.search(r'\b'+reg+r'\b',k[n]) : print('  USE before wait:',n,k[n].strip()[:100])
PY
