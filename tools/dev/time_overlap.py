import importlib, os, sys, time, torch
sys.path.insert(0, '/root/repo')
t = importlib.import_module('3dgp_amd')
cfg = t.config.config_c3()
G = t.generator.Generator(cfg); G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0)); G = G.cuda()
inp = t.weights.synthetic_inputs(cfg, batch=16, seed=0)
T = lambda a: torch.as_tensor(a).cuda()
z, c, cam = T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}
dec = G.synthesis.tri_plane_decoder
for rep in range(2):
    for ov in (True, False):
        dec.overlap_torgb = ov
        for _ in range(3): G(z, c, cam, noise_mode='const')
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): G(z, c, cam, noise_mode='const')
        torch.cuda.synchronize(); print('overlap', ov, round((time.perf_counter() - t0) / 10 * 1e3, 3), 'ms')
