import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
def run(B,cin,H,skip,layout,reps=5):
    x = torch.randn(B,cin,H,H,device='cuda'); w = torch.randn(96,cin,1,1,device='cuda'); s = torch.rand(B,cin,device='cuda')+0.5
    bias = torch.randn(96,device='cuda'); f = np.outer([1,3,3,1],[1,3,3,1]).astype(np.float32)/64
    pk = mc.PackedConv(w); fir = mc.fir_host_array(f)
    sk = None
    if skip: sk = torch.randn(B,3,H//2,H//2,32,device='cuda') if layout else torch.randn(B,96,H//2,H//2,device='cuda')
    kw = dict(bias=bias, demodulate=False, skip=sk, fir=fir if skip else None, out_layout=layout, out_feat=32 if layout else 0)
    for _ in range(2): y = mc.modconv_forward(x,pk,s,**kw)
    torch.cuda.synchronize(); t._lib.profile_enable(True)
    for _ in range(reps): y = mc.modconv_forward(x,pk,s,**kw)
    torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    return {k: round(v['avg_ms']*1e3,1) for k, v in r.items()}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for H,cin in ((512,64),(256,128),(128,256),(64,512),(32,512)):
    for skip in (0,1):
        r = run(B,cin,H,skip,1)
        byt = B*H*H*4*(cin + 96 + (24 if skip else 0))
        print('H',H,'cin',cin,'skip',skip, r, 'GB/s', round(byt/list(r.values())[0]/1e3), 'TF', round(2*B*H*H*cin*96/list(r.values())[0]/1e6,1))
