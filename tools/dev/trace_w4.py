"""Cycles per phase of conv3_wino4_kernel's item loop, x2 and plain layers of C3 (a build with -DTDGP_W4_TRACE=1: tools/dev/build_variant.sh w4trace modconv -DTDGP_W4_TRACE=1).
   python tools/dev/with_lib.py tools/dev/variants/w4trace.so tools/dev/trace_w4.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
U = importlib.import_module('3dgp_amd.ops.upfirdn2d')
dev = torch.device('cuda')
torch.manual_seed(0)
fir = M.fir_host_array(U.setup_filter([1, 3, 3, 1]))
b = 16
for (Ci, Co, R, up) in [(128, 64, 256, 2), (256, 128, 128, 2), (512, 256, 64, 2), (512, 512, 32, 2), (256, 256, 128, 1), (512, 512, 64, 1)]:
    x = torch.randn(b, Ci, R, R, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev)
    s = torch.randn(b, Ci, device=dev) * 0.5 + 1.0
    nz = torch.randn(up * R, up * R, device=dev) * 0.1
    bias = torch.randn(Co, device=dev) * 0.1
    pk = M._packed(w)
    kw = dict(up=2, fir=fir) if up == 2 else {}
    for _ in range(3):
        y = M.modconv_forward(x, pk, s, noise=nz, bias=bias, act='lrelu', **kw)
    torch.cuda.synchronize()
    t = y.flatten()[:256 * 8].view(torch.int32).reshape(256, 8).cpu().double()
    items = t[:, 7]
    names = ['item top: ticket draw, bias / demod loads', 'wait for the first chunk + barrier', 'K loop', 'ticket hand-over', 'first chunk(s) of the next item requested', 'output stage', 'item-end barrier']
    tot = t[:, :7].sum(1)
    nch = Ci // 4
    print(f'Cin={Ci} Cout={Co} R={R} x{up} B={b}: items/block {items.min():.0f}..{items.max():.0f}, ticks/block {tot.mean():.0f}, K loop {t[:, 2].sum() / items.sum() / nch:.0f} ticks per chunk')
    for i, n in enumerate(names):
        print(f'   {n:46s} {t[:, i].sum() / items.sum():9.0f} ticks/item  {100 * t[:, i].sum() / tot.sum():5.1f} %')
