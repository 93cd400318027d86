"""Stand-alone upfirdn2d at the generator's hot sizes: achieved GB/s of algorithmic traffic (input once + output once).
   python tools/dev/bench_upfirdn.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
u = t.ops.upfirdn2d
f = u.setup_filter([1, 3, 3, 1]).cuda()
for name, shape, kw in (('F1 64ch 513^2 -> 512^2', (8, 64, 513, 513), dict(padding=[1, 1, 1, 1], gain=4)),
                        ('F1 128ch 257^2 -> 256^2', (8, 128, 257, 257), dict(padding=[1, 1, 1, 1], gain=4)),
                        ('F2 96ch 256^2 -> 512^2', (8, 96, 256, 256), dict(up=2, padding=[2, 1, 2, 1], gain=4)),
                        ('F2 96ch 128^2 -> 256^2', (8, 96, 128, 128), dict(up=2, padding=[2, 1, 2, 1], gain=4))):
    x = torch.randn(*shape, device='cuda')
    for _ in range(3):
        y = u.upfirdn2d(x, f, **kw)
    torch.cuda.synchronize()
    t._lib.profile_enable(True)
    for _ in range(10):
        y = u.upfirdn2d(x, f, **kw)
    torch.cuda.synchronize()
    r = t._lib.profile_report()['upfirdn2d_4x4']
    t._lib.profile_enable(False)
    nbytes = (x.numel() + y.numel()) * 4
    print(f'{name}: {r["avg_ms"] * 1e3:.1f} us, {nbytes / r["avg_ms"] / 1e6:.0f} GB/s of algorithmic traffic ({nbytes / 1e6:.0f} MB)')
