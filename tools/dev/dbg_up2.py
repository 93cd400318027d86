import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
import oracle
T = lambda a: torch.as_tensor(a).cuda()
f = oracle.setup_filter([1, 3, 3, 1])
B, cin, cout, H = 16, 64, 64, 64
rs = np.random.RandomState(1)
x = rs.randn(B, cin, H, H).astype(np.float32); w = rs.randn(cout, cin, 3, 3).astype(np.float32)
s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32)
bias = T((0.2 * rs.randn(cout)).astype(np.float32)); noise = T((0.3 * rs.randn(2 * H, 2 * H)).astype(np.float32))
pk = mc.PackedConv(T(w))
for kw in (dict(act='lrelu'), dict(act='linear', bias=bias), dict(act='linear', noise=noise), dict(act='lrelu', bias=bias, noise=noise)):
    mc.FOLD_UP2 = False
    ref = mc.modconv_forward(T(x), pk, T(s), demodulate=True, up=2, fir=mc.fir_host_array(f), **kw)
    mc.FOLD_UP2 = True
    y = mc.modconv_forward(T(x), pk, T(s), demodulate=True, up=2, fir=mc.fir_host_array(f), **kw)
    e = ((y - ref).abs().amax((1, 2, 3)) / ref.abs().amax((1, 2, 3))).tolist()
    print({k: (v if isinstance(v, str) else 'T') for k, v in kw.items()}, 'err per sample', [round(v, 6) for v in e[:5]])
    if e[0] > 1e-3:
        dd = (y[0] - ref[0])
        print('   diff sample 0: ch0 mean', float(dd[0].mean()), 'std', float(dd[0].std()), ' corr with noise', float((dd[0] * noise).mean() / (noise * noise).mean()) if 'noise' in kw else '-')
