import sys, os, importlib, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
def run(B,cin,cout,H,k,up,reps=10):
    x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(cout,cin,k,k,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
    bias = torch.randn(cout,device='cuda'); f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
    pk = mc.PackedConv(w); fir = mc.fir_host_array(f)
    for _ in range(3): y = mc.modconv_forward(x,pk,s,bias=bias,up=up,demodulate=(k==3),act='lrelu',fir=fir)
    torch.cuda.synchronize(); t._lib.profile_enable(True)
    for _ in range(reps): y = mc.modconv_forward(x,pk,s,bias=bias,up=up,demodulate=(k==3),act='lrelu',fir=fir)
    torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    fl = 2*cin*cout*k*k*H*H*B
    return {k2: round(v['avg_ms']*1e3,1) for k2,v in r.items()}, fl
for shape in [(4,128,128,256,3,1),(4,512,512,64,3,1),(4,64,64,512,3,1),(4,256,128,128,3,2)]:
    r, fl = run(*shape)
    us = r['conv_mfma_kernel']
    print(os.environ.get('TDGP_CONV_DBG','0'), shape, r, 'conv TF/s', round(fl/us/1e6,1))
