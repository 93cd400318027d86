"""Fused-input-transform F(4x4) kernel (modconv_wino4f.inc, conv arith 0) vs the two-kernel path (conv arith 4): same bits, time per layer.
   python tools/dev/check_w4f.py [B]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
L = tdgp._lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')
torch.manual_seed(0)
for (Ci, Co, R) in [(64, 64, 512), (128, 128, 256), (64, 128, 256), (128, 64, 128), (32, 64, 128), (48, 64, 64)]:
    b = B if Ci * R * R * B < (1 << 30) else B // 2
    x = torch.randn(b, Ci, R, R, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev)
    s = torch.randn(b, Ci, device=dev) * 0.5 + 1.0
    nz = torch.randn(R, R, device=dev) * 0.1
    bias = torch.randn(Co, device=dev) * 0.1
    pk = M._packed(w)
    out = {}
    for mode in (4, 0):
        L.set_conv_arith(mode)
        L.profile_enable(True)
        y = M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
        torch.cuda.synchronize()
        names = sorted(L.profile_report().keys())
        L.profile_enable(False)
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            y = M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
        torch.cuda.synchronize()
        out[mode] = (y, (time.perf_counter() - t0) / n * 1e3, names)
    L.set_conv_arith(0)
    (y4, t4, n4), (y0, t0_, n0) = out[4], out[0]
    same = torch.equal(y4, y0)
    err = (y4 - y0).abs().max().item() / y4.abs().max().item()
    print(f'Cin={Ci:4d} Cout={Co:4d} R={R:4d} B={b}: two-kernel {t4:7.3f} ms {n4}   fused {t0_:7.3f} ms {n0}   same bits: {same}  (max|d|/max|y| = {err:.2e})', flush=True)
