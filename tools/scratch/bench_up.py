import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
def run(cin,cout,H,reps=5):
    x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(cout,cin,3,3,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
    bias = torch.randn(cout,device='cuda')
    pk = mc.PackedConv(w); fir = mc.fir_host_array(f)
    kw = dict(bias=bias,up=2,demodulate=True,act='lrelu',fir=fir)
    for _ in range(2): y = mc.modconv_forward(x,pk,s,**kw)
    torch.cuda.synchronize(); t._lib.profile_enable(True)
    for _ in range(reps): y = mc.modconv_forward(x,pk,s,**kw)
    torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    return {k2: round(v['avg_ms']*1e3,1) for k2,v in r.items()}, 2*cin*cout*9*H*H*B
out = []
for cin,cout,H in [(512,512,16),(512,512,32),(512,256,64),(256,128,128),(128,64,256)]:
    r, fl = run(cin,cout,H)
    out.append(f"{cin}->{cout}@{H}: up {r['upconv_mfma_kernel']:.0f} fir {r['fir_act_kernel']:.0f} TF {fl/r['upconv_mfma_kernel']/1e6:.1f}")
print(os.environ.get('TDGP_UP_CFG','-'), os.environ.get('TDGP_UP_KS','-'), ' | '.join(out))
