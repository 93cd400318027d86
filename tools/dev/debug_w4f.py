import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
L = tdgp._lib
dev = torch.device('cuda')
torch.manual_seed(0)
for (Ci, Co, R, b, kind) in [(64, 64, 128, 16, 'rand'), (64, 64, 128, 16, 'rand'), (64, 64, 256, 4, 'rand')]:
    x = torch.randn(b, Ci, R, R, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev)
    if kind == 'ch':      # only input channel 5 non-zero, centre tap only: y[o] = w[o,5,1,1] * x[5]
        w2 = torch.zeros_like(w); w2[:, 5, 1, 1] = w[:, 5, 1, 1]; w = w2
    s = torch.ones(b, Ci, device=dev)
    pk = M._packed(w)
    out = {}
    for mode in (4, 0):
        L.set_conv_arith(mode)
        L.profile_enable(True)
        y = M.modconv_forward(x, pk, s, noise=None, bias=None, act='linear', demodulate=False)
        torch.cuda.synchronize()
        names = sorted(L.profile_report().keys())
        L.profile_enable(False)
        out[mode] = y
        print(mode, names)
    L.set_conv_arith(0)
    d = (out[4] - out[0]).abs()
    print(Ci, Co, R, kind, 'max err', d.max().item(), 'ref max', out[4].abs().max().item())
    print(' per sample', [round(v, 3) for v in d.amax(dim=(1, 2, 3)).tolist()])
    print(' per out-channel (first 16, every 4)', [round(v, 3) for v in d.amax(dim=(0, 2, 3)).tolist()][:64:4])
    print(' per row%8', [round(d[:, :, r::8, :].max().item(), 3) for r in range(8)])
    print(' per col%4 / col//4 %16', [round(d[:, :, :, c::4].max().item(), 3) for c in range(4)], [round(d[:, :, :, 4 * c:4 * c + 4].max().item(), 3) for c in range(16)])
    print(' sample 0: per gy', [round(d[0, :, 8 * g:8 * g + 8, :].max().item(), 2) for g in range(R // 8)])
    print(' sample 0: per gx', [round(d[0, :, :, 64 * g:64 * g + 64].max().item(), 2) for g in range(R // 64)])
    print(' sample 0: per row of the first 8', [round(d[0, :, r, :].max().item(), 2) for r in range(8)])
    print(' sample 0: err by in-channel probe: see below')
    if kind == 'ch':
        print(' fused y[0,0,:6,:6]'); print(out[0][0, 0, :6, :6]); print(' ref'); print(out[4][0, 0, :6, :6])
