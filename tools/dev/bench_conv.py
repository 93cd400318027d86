"""Stride-1 3x3 layer: Winograd (arith 0) vs direct (arith 2) -- agreement and time per layer shape.  python tools/dev/bench_conv.py [B]"""
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
for (C, R) in [(512, 32), (512, 64), (256, 128), (128, 256), (64, 512)]:
    x = torch.randn(B, C, R, R, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev)
    s = torch.randn(B, C, device=dev) * 0.5 + 1.0
    nz = torch.randn(R, R, device=dev) * 0.1
    bias = torch.randn(C, device=dev) * 0.1
    pk = M._packed(w)
    out = {}
    for mode in (2, 0):
        L.set_conv_arith(mode)
        y = M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            y = M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
        torch.cuda.synchronize()
        out[mode] = (y, (time.perf_counter() - t0) / n * 1e3)
    L.set_conv_arith(0)
    yd, td = out[2]
    yw, tw = out[0]
    scale = yd.abs().max().item()
    err = (yd - yw).abs().max().item() / scale
    gf = 2.0 * B * R * R * C * C * 9 / 1e9
    print(f'C={C:4d} R={R:4d}: direct {td:7.3f} ms ({gf / td:6.1f} TF)  wino {tw:7.3f} ms ({gf / tw:6.1f} TF-equivalent)  max|d|/max|y| = {err:.2e}', flush=True)
