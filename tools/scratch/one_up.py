import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
t._lib.set_conv_arith(1)
B, cin, cout, H = [int(a) for a in sys.argv[1:5]]
f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
x = torch.randn(B, cin, H, H, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda'); s = torch.rand(B, cin, device='cuda') + 0.5
pk = mc.PackedConv(w)
y = mc.modconv_forward(x, pk, s, bias=torch.randn(cout, device='cuda'), demodulate=True, act='lrelu', up=2, fir=mc.fir_host_array(f))
torch.cuda.synchronize()
