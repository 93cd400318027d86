"""Repeat every hot-path entry point on identical inputs and report anything that is not bit-reproducible."""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
T = lambda a: torch.as_tensor(a).cuda()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30

def rep(name, fn, n=N):
    ref = fn(); bad = 0
    ref = [r.clone() for r in (ref if isinstance(ref, (list, tuple)) else [ref])]
    for i in range(n):
        out = fn(); out = out if isinstance(out, (list, tuple)) else [out]
        for k, (a, b) in enumerate(zip(out, ref)):
            if not torch.equal(a, b):
                d = (a - b).abs(); bad += 1
                print(f'  {name}: run {i} output {k} differs: max {d.max().item():.3e} (scale {b.abs().max().item():.3e}) at {int((d > 0).sum())} elements')
                break
    print(f'{name}: {bad} bad of {n}')

for cfgname, B in (('c3', 2), ('c3', 8), ('c4', 1)):
    cfg = getattr(t.config, 'config_' + cfgname)()
    G = t.generator.Generator(cfg); G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=3)); G = G.cuda()
    inp = t.weights.synthetic_inputs(cfg, batch=B, seed=4)
    cam = {k: T(v) for k, v in inp['camera'].items()}
    kw = dict(noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
    z, c = T(inp['z']), T(inp['c'])
    rep(f'G forward {cfgname} B={B}', lambda: G(z, c, cam, **kw))
    if cfgname == 'c3' and B == 2:
        t._lib.set_conv_arith(1)
        rep('G forward c3 B=2 split', lambda: G(z, c, cam, **kw))
        t._lib.set_conv_arith(0)
        for p in G.parameters(): p.requires_grad_(True)
        ps = [p for p in G.parameters()]
        def fb():
            img = G.forward_autograd(z, c, cam, **kw)
            loss = torch.nn.functional.softplus(-img).mean()
            g = torch.autograd.grad(loss, ps, allow_unused=True)
            return [x for x in g if x is not None]
        names = [n for n, p in G.named_parameters()]
        ref = fb(); ref = [r.clone() for r in ref]
        for i in range(6):
            out = fb()
            diffs = [(names[k], ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()) for k, (a, b) in enumerate(zip(out, ref)) if not torch.equal(a, b)]
            big = [d for d in diffs if d[1] > 1e-4]
            print(f'G backward run {i}: {len(diffs)} tensors not bit-equal (atomics in the plane scatter), {len(big)} beyond 1e-4 rel', big[:4])
        for p in G.parameters(): p.requires_grad_(False)
    del G
    torch.cuda.empty_cache()
