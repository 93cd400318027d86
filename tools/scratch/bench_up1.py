import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
B = 8; cin, cout, H = 256, 128, 128
f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(cout,cin,3,3,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
pk = mc.PackedConv(w); fir = mc.fir_host_array(f)
for _ in range(2):
    y = mc.modconv_forward(x,pk,s,bias=None,up=2,demodulate=True,act='lrelu',fir=fir)
    torch.cuda.synchronize()
