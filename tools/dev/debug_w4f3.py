import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
L = tdgp._lib
dev = torch.device('cuda')
torch.manual_seed(0)
Ci, Co, R, b = 64, 64, 128, 16
x = torch.randn(b, Ci, R, R, device=dev)
w = torch.randn(Co, Ci, 3, 3, device=dev)
s = torch.ones(b, Ci, device=dev)
pk = M._packed(w)
out = {}
for mode in (4, 0):
    L.set_conv_arith(mode)
    out[mode] = M.modconv_forward(x, pk, s, noise=None, bias=None, act='linear', demodulate=False)
    torch.cuda.synchronize()
L.set_conv_arith(0)
d = (out[4] - out[0]).abs()
print(L.LIB_PATH)
print(' per sample', [round(v, 2) for v in d.amax(dim=(1, 2, 3)).tolist()])
for bb in range(b):
    if d[bb].max() > 0:
        print(' sample', bb, 'per gy', [round(d[bb, :, 8 * g:8 * g + 8, :].max().item(), 1) for g in range(R // 8)], 'rows of bad group', [round(d[bb, :, r, :].max().item(), 1) for r in range(R) if d[bb, :, r, :].max() > 0][:10], [r for r in range(R) if d[bb, :, r, :].max() > 0][:10])
        print('   per out channel %16', [round(d[bb, o::16].max().item(), 1) for o in range(16)])
