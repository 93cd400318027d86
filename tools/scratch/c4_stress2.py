import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
cfg = getattr(tdgp.config, 'config_' + (sys.argv[2] if len(sys.argv) > 2 else 'c4'))()
G = tdgp.generator.Generator(cfg); G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=3)); G = G.cuda()
inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=4)
T = lambda a: torch.as_tensor(a).cuda()
ws = G.mapping(T(inp['z']), T(inp['c']))
dec = G.synthesis.tri_plane_decoder
rec = {}
def mk(res):
    def hook(mod, args, out):
        rec[res] = (out[0].clone(), out[1].clone())
    return hook
for res in dec.block_resolutions:
    getattr(dec, f'b{res}').register_forward_hook(mk(res))
def mk2(name):
    def hook(mod, args, out):
        rec[name] = out.clone()
    return hook
dec.b256.conv0.register_forward_hook(mk2('c0')); dec.b256.conv1.register_forward_hook(mk2('c1'))
dec(ws, noise_mode='const', hwc=True); ref = dict(rec)
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    rec.clear()
    dec(ws, noise_mode='const', hwc=True)
    for nm in ('c0', 'c1'):
        d = (rec[nm] - ref[nm]).abs()
        if d.max().item() > 0:
            idx = (d > 0).nonzero()
            print('run', i, nm, 'max', d.max().item(), 'n', idx.shape[0], 'ch', idx[:,1].min().item(), idx[:,1].max().item(), 'rows', idx[:,2].min().item(), idx[:,2].max().item(), 'cols', idx[:,3].min().item(), idx[:,3].max().item())
    for res in dec.block_resolutions:
        dx = (rec[res][0] - ref[res][0]).abs().max().item(); di = (rec[res][1] - ref[res][1]).abs()
        if dx > 0 or di.max().item() > 0:
            idx = (di > 0).nonzero()
            print('run', i, 'res', res, 'x diff', dx, 'img diff', di.max().item(), 'img scale', ref[res][1].abs().max().item(), 'n', idx.shape[0],
                  'first', idx[0].tolist() if idx.shape[0] else None, 'last', idx[-1].tolist() if idx.shape[0] else None)
            bad += 1
            break
print('bad runs', bad)
