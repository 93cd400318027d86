import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
cfg = t.config.config_c3()
G = t.generator.Generator(cfg); G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0)); G = G.cuda()
inp = t.weights.synthetic_inputs(cfg, batch=8, seed=0)
T = lambda a: torch.as_tensor(a).cuda()
G(T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}, noise_mode='const', u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))
torch.cuda.synchronize()
