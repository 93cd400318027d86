import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
if os.environ.get('TDGP_ARITH') == 'split': t._lib.set_conv_arith(1)
B, c, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B, c, H, H, device='cuda'); w = torch.randn(c, c, 3, 3, device='cuda'); s = torch.rand(B, c, device='cuda') + 0.5
pk = mc.PackedConv(w)
y = mc.modconv_forward(x, pk, s, bias=torch.randn(c, device='cuda'), demodulate=True, act='lrelu')
torch.cuda.synchronize()
