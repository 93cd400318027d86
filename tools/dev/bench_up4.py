"""Per-kernel times of the x2 layers of C3 / C4 (folded F(4x4) path) from the library profiler: python tools/dev/bench_up4.py [B] [c3|c4]"""
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
shapes = dict(c3=[(512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256)],
              c4=[(1024, 1024, 32), (1024, 512, 64), (512, 256, 128), (256, 128, 256)])[sys.argv[2] if len(sys.argv) > 2 else 'c3']
dev = torch.device('cuda')
torch.manual_seed(0)
fir = M.fir_host_array(U.setup_filter([1, 3, 3, 1]))
out = []
for (Ci, Co, R) in shapes:
    x = torch.randn(B, Ci, R, R, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev)
    s = torch.randn(B, Ci, device=dev) * 0.5 + 1.0
    nz = torch.randn(2 * R, 2 * R, device=dev) * 0.1
    bias = torch.randn(Co, device=dev) * 0.1
    pk = M._packed(w)
    f = lambda: M.modconv_forward(x, pk, s, noise=nz, bias=bias, up=2, fir=fir, act='lrelu')      # noqa: E731
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    L.profile_enable(True)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    r = L.profile_report()
    L.profile_enable(False)
    gf = 2.0 * B * R * R * Ci * Co * 9 / 1e9
    ks = {k.replace('_kernel', ''): round(v['avg_ms'], 3) for k, v in r.items() if v['avg_ms'] > 0.01}
    tot = sum(v['total_ms'] for v in r.values()) / 5
    out.append(f'{Ci}->{Co} @{R}->{2 * R}: {tot:.3f} ms ({gf / tot:.0f} TF) {ks}')
    del x, w, pk
print(os.path.basename(L.LIB_PATH), ' | '.join(out))
