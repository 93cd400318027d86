"""Per-layer times of the ToRGB layers of C3 (channel-last planes, fused x2 skip) from the library profiler: python tools/dev/bench_torgb.py [B]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
U = importlib.import_module('3dgp_amd.ops.upfirdn2d')
L = tdgp._lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')
torch.manual_seed(0)
fir = M.fir_host_array(U.setup_filter([1, 3, 3, 1]))
out = []
tot_all = 0.0
for (C, R) in [(512, 4), (512, 8), (512, 16), (512, 32), (512, 64), (256, 128), (128, 256), (64, 512)]:
    x = torch.randn(B, C, R, R, device=dev)
    w = torch.randn(96, C, 1, 1, device=dev)
    s = torch.randn(B, C, device=dev) * 0.05
    bias = torch.randn(96, device=dev) * 0.1
    skip = None if R == 4 else torch.randn(B, 3, R // 2, R // 2, 32, device=dev)
    pk = M._packed(w)
    f = lambda: M.modconv_forward(x, pk, s, bias=bias, demodulate=False, act='linear', gain=1.0, skip=skip, fir=fir if skip is not None else None, out_layout=1, out_feat=32)   # noqa: E731
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    L.profile_enable(True)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    r = L.profile_report()
    L.profile_enable(False)
    tot = sum(v['total_ms'] for v in r.values()) / 5
    tot_all += tot
    gb = B * R * R * (C * 4 + 96 * 4 + (96 if skip is not None else 0)) / 1e9
    out.append(f'C={C} R={R}: {tot * 1e3:.0f} us ({gb / tot:.2f} TB/s, {2.0 * B * R * R * C * 96 / tot / 1e9:.0f} TF)')
    del x, w, pk, skip
print(os.path.basename(L.LIB_PATH), f'B={B} total {tot_all:.3f} ms |', ' | '.join(out))
