import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
M = importlib.import_module('3dgp_amd.ops.modconv')
L = tdgp._lib
dev = torch.device('cuda')
torch.manual_seed(0)
Ci, Co, R, b = 64, 64, 128, 16
for cin_probe in (0, 1, 4, 5, 8):
    x = torch.zeros(b, Ci, R, R, device=dev)
    # every pixel of every channel carries its own code: channel * 1000 + row * 1 + col * 0.001 (sample 1: negative)
    rr = torch.arange(R, device=dev).float()
    for c in range(Ci):
        x[0, c] = c * 1000 + rr[:, None] + rr[None, :] * 0.001
        for bb in range(1, b):
            x[bb, c] = -(bb * 100000 + c * 1000 + rr[:, None] + rr[None, :] * 0.001)
    w = torch.zeros(Co, Ci, 3, 3, device=dev)
    w[:, cin_probe, 0, 1] = 1.0          # y[r, c] = x[cin_probe][r - 1, c]
    s = torch.ones(b, Ci, device=dev)
    pk = M._packed(w)
    for mode in (4, 0):
        L.set_conv_arith(mode)
        y = M.modconv_forward(x, pk, s, noise=None, bias=None, act='linear', demodulate=False)
        torch.cuda.synchronize()
        L.profile_enable(True)
        y = M.modconv_forward(x, pk, s, noise=None, bias=None, act='linear', demodulate=False)
        torch.cuda.synchronize()
        print(sorted(L.profile_report().keys()))
        L.profile_enable(False)
        print('probe', cin_probe, 'mode', mode, 'y[0,0,0:2,0:6]', y[0, 0, 0:2, 0:6].tolist(), ' y[0,0,0,60:68]', [round(v, 3) for v in y[0, 0, 0, 60:68].tolist()], ' y[1,0,0,0:3]', y[1, 0, 0, 0:3].tolist())
    L.set_conv_arith(0)
