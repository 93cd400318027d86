import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
B = 8
for cin, cout, H in ((128, 128, 256), (64, 64, 512)):
    x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(cout,cin,3,3,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
    pk = mc.PackedConv(w); bias = torch.randn(cout, device='cuda')
    for _ in range(2):
        y = mc.modconv_forward(x,pk,s,bias=bias,up=1,demodulate=True,act='lrelu')
        torch.cuda.synchronize()
