"""Per-kernel times of the stride-1 3x3 layers of C3 / C4 under the default arithmetic (library profiler): python tools/dev/bench_wino4.py [B] [c3|c4]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
L = tdgp._lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = dict(c3=[(512, 32), (512, 64), (256, 128), (128, 256), (64, 512)], c4=[(1024, 32), (1024, 64), (512, 128), (256, 256), (128, 512)], low=[(512, 4), (512, 8), (512, 16), (512, 32)])[sys.argv[2] if len(sys.argv) > 2 else 'c3']
dev = torch.device('cuda')
torch.manual_seed(0)
out = []
for (C, R) in shapes:
    x = torch.randn(B, C, R, R, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev)
    s = torch.randn(B, C, device=dev) * 0.5 + 1.0
    nz = torch.randn(R, R, device=dev) * 0.1
    bias = torch.randn(C, device=dev) * 0.1
    pk = M._packed(w)
    for _ in range(2):
        M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
    torch.cuda.synchronize()
    L.profile_enable(True)
    for _ in range(5):
        M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
    torch.cuda.synchronize()
    r = L.profile_report()
    L.profile_enable(False)
    gf = 2.0 * B * R * R * C * C * 9 / 1e9
    ks = {k.replace('_kernel', ''): round(v['avg_ms'], 3) for k, v in r.items() if v['avg_ms'] > 0.01}
    tot = sum(v['total_ms'] for v in r.values()) / 5
    out.append(f'C={C} R={R}: {tot:.3f} ms ({gf / tot:.0f} TF-eq) {ks}')
    del x, w, pk
print(os.path.basename(L.LIB_PATH), ' | '.join(out))
