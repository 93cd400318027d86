"""Stress (development): repeat F(4x4) layers many times and compare every output with the first, bit for bit -- the 8-wave form's waits count on
LDS-direct loads completing in issue order; a violation would show up as run-to-run differences.   python tools/dev/stress_w4.py [repeats]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
U = importlib.import_module('3dgp_amd.ops.upfirdn2d')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda')
torch.manual_seed(1)
fir = M.fir_host_array(U.setup_filter([1, 3, 3, 1]))
bad = 0
for (B, Ci, Co, R, up) in [(16, 512, 512, 64, 1), (16, 128, 128, 256, 1), (16, 64, 64, 512, 1), (4, 512, 512, 32, 1), (16, 256, 128, 128, 2), (16, 512, 512, 32, 2), (5, 128, 200, 64, 1)]:
    x = torch.randn(B, Ci, R, R, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev)
    s = torch.randn(B, Ci, device=dev) * 0.5 + 1.0
    bias = torch.randn(Co, device=dev) * 0.1
    pk = M._packed(w)
    f = (lambda: M.modconv_forward(x, pk, s, bias=bias, act='lrelu')) if up == 1 else (lambda: M.modconv_forward(x, pk, s, bias=bias, up=2, fir=fir, act='lrelu'))
    ref = f().clone()
    other = torch.randn(64 << 20, device=dev)          # something else churning the caches between repeats
    nb = 0
    for i in range(N):
        if i % 7 == 0:
            other.mul_(1.0001)
        y = f()
        if not torch.equal(y, ref):
            nb += 1
    print(f'B={B} {Ci}->{Co} @{R} up={up}: {nb} of {N} repeats differ')
    bad += nb
    del x, w, pk, ref, other
print('TOTAL differing repeats:', bad)
