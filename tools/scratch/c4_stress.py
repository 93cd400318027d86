import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tdgp = importlib.import_module('3dgp_amd')
cfg = tdgp.config.config_c4()
G = tdgp.generator.Generator(cfg); G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=3)); G = G.cuda()
inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=4)
T = lambda a: torch.as_tensor(a).cuda()
ws = G.mapping(T(inp['z']), T(inp['c']))
dec = G.synthesis.tri_plane_decoder
ref = dec(ws, noise_mode='const').clone()
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    if i % 2 == 0:
        y = dec(ws, noise_mode='const')
    else:
        y = dec(ws, noise_mode='const', hwc=True).t.permute(0, 1, 4, 2, 3).reshape(ref.shape)
    d = (y - ref).abs().max().item() / ref.abs().max().item()
    if d > 5e-6:
        bad += 1
        diff = (y - ref).abs()
        idx = (diff > 1e-3 * ref.abs().max()).nonzero()
        print('run', i, 'hwc' if i % 2 else 'nchw', 'rel', d, 'n bad', idx.shape[0], 'first', idx[0].tolist(), 'last', idx[-1].tolist())
print('bad runs', bad)
