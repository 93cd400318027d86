import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
t = importlib.import_module('3dgp_amd'); mc = t.ops.modconv
import oracle
T = lambda a: torch.as_tensor(a).cuda()
f = oracle.setup_filter([1, 3, 3, 1])
for (B, cin, cout, H) in [(16, 128, 64, 64), (16, 64, 64, 64), (4, 512, 256, 64)]:
    rs = np.random.RandomState(1)
    x = rs.randn(B, cin, H, H).astype(np.float32); w = rs.randn(cout, cin, 3, 3).astype(np.float32)
    s = (1 + 0.5 * rs.randn(B, cin)).astype(np.float32); bias = (0.2 * rs.randn(cout)).astype(np.float32)
    noise = (0.3 * rs.randn(2 * H, 2 * H)).astype(np.float32)
    pk = mc.PackedConv(T(w))
    mc.FOLD_UP2 = False
    ref = mc.modconv_forward(T(x), pk, T(s), noise=T(noise), bias=T(bias), demodulate=True, act='lrelu', up=2, fir=mc.fir_host_array(f))
    mc.FOLD_UP2 = True
    bad = 0
    for it in range(12):
        y = mc.modconv_forward(T(x), pk, T(s), noise=T(noise), bias=T(bias), demodulate=True, act='lrelu', up=2, fir=mc.fir_host_array(f))
        d = (y - ref).abs()
        e = float(d.max() / ref.abs().max())
        if e > 1e-4:
            bad += 1
            idx = (d > 1e-3 * ref.abs().max()).nonzero()
            print(' iteration', it, 'err', e, 'bad elements', idx.shape[0], 'samples', idx[:, 0].unique().tolist()[:8], 'channels', idx[:, 1].unique().tolist()[:12],
                  'rows', idx[:, 2].unique().tolist()[:12], 'cols', idx[:, 3].unique().tolist()[:12])
    print((B, cin, cout, H), 'bad iterations', bad, 'last err', e)
