import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
B = 8
f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
for cin, H in ((64, 512), (128, 256)):
    x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(96,cin,1,1,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
    pk = mc.PackedConv(w); bias = torch.randn(96, device='cuda'); fir = mc.fir_host_array(f)
    sk = torch.randn(B,3,H//2,H//2,32,device='cuda')
    for _ in range(2):
        y = mc.modconv_forward(x,pk,s,bias=bias,demodulate=False,skip=sk,fir=fir,out_layout=1,out_feat=32)
        torch.cuda.synchronize()
