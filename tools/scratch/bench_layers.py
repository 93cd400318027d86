import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if os.environ.get('TDGP_ARITH') == 'split': t._lib.set_conv_arith(1)
f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
def run(cin,cout,H,k,up,reps=5):
    x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(cout,cin,k,k,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
    bias = torch.randn(cout,device='cuda')
    pk = mc.PackedConv(w); fir = mc.fir_host_array(f)
    kw = dict(bias=bias,up=up,demodulate=(k==3),act='lrelu' if k==3 else 'linear',fir=fir)
    for _ in range(2): y = mc.modconv_forward(x,pk,s,**kw)
    torch.cuda.synchronize(); t._lib.profile_enable(True)
    for _ in range(reps): y = mc.modconv_forward(x,pk,s,**kw)
    torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    fl = 2*cin*cout*k*k*H*H*B
    return {k2: round(v['avg_ms']*1e3,1) for k2,v in r.items()}, fl
ch = {4:512,8:512,16:512,32:512,64:512,128:256,256:128,512:64}
tot = 0; totfl = 0
for r_ in [4,8,16,32,64,128,256,512]:
    layers = []
    if r_ > 4: layers.append(('conv0', ch[r_//2], ch[r_], r_//2, 3, 2))
    layers.append(('conv1', ch[r_], ch[r_], r_, 3, 1))
    layers.append(('torgb', ch[r_], 96, r_, 1, 1))
    for name,cin,cout,H,k,up in layers:
        r, fl = run(cin,cout,H,k,up)
        us = sum(r.values()); tot += us; totfl += fl
        print(f'b{r_:<4}{name} {cin:>4}->{cout:<4}@{H:<4} up{up}', r, 'conv TF/s', round(fl/r.get('conv_mfma_kernel', r.get('upconv_mfma_kernel'))/1e6,1), 'all-in TF/s', round(fl/us/1e6,1))
print('total us', round(tot), 'TF/s', round(totfl/tot/1e6,1))
