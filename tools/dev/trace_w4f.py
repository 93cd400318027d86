"""Cycles per phase of the fused F(4x4) kernel's item loop (a build with -DTDGP_W4F_TRACE=1: tools/dev/build_variant.sh w4ftrace modconv -DTDGP_W4F_TRACE=1).
   python tools/dev/with_lib.py tools/dev/variants/w4ftrace.so tools/dev/trace_w4f.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
dev = torch.device('cuda')
torch.manual_seed(0)
for (Ci, Co, R, b) in [(64, 64, 512, 16), (128, 128, 256, 16)]:
    x = torch.randn(b, Ci, R, R, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev)
    s = torch.randn(b, Ci, device=dev) * 0.5 + 1.0
    nz = torch.randn(R, R, device=dev) * 0.1
    bias = torch.randn(Co, device=dev) * 0.1
    pk = M._packed(w)
    for _ in range(3):
        y = M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu')
    torch.cuda.synchronize()
    t = y.flatten()[:256 * 16].view(torch.int32).reshape(256, 16).cpu().double()
    items = t[:, 11]
    names = ['prologue: 2nd barrier', 'K loop', 'begin_item', 'output stage', 'item-end barrier', 'prologue: ticket draw, bias / demod loads', 'prologue: wait for the requests',
             'prologue: 1st barrier', 'prologue: transform', 'ticket hand-over (2 barriers) + its arithmetic']
    tot = t[:, :10].sum(1)
    print(f'Cin={Ci} Cout={Co} R={R} B={b}: items/block {items.min():.0f}..{items.max():.0f}, cycles/block {tot.mean():.0f} (s_memtime ticks)')
    for i, n in enumerate(names):
        print(f'   {n:42s} {t[:, i].sum() / items.sum():9.0f} ticks/item  {100 * t[:, i].sum() / tot.sum():5.1f} %')
