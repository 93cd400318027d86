import sys, os, importlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
t = importlib.import_module('3dgp_amd')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = t.config.config_c3()
G = t.generator.Generator(cfg); G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0)); G = G.cuda()
inp = t.weights.synthetic_inputs(cfg, batch=B, seed=0)
T = lambda a: torch.as_tensor(a).cuda()
z, c = T(inp['z']), T(inp['c']); cam = {k: T(v) for k, v in inp['camera'].items()}; u1, u2 = T(inp['u_coarse']), T(inp['u_fine'])
for _ in range(2): G(z, c, cam, noise_mode='const', u_coarse=u1, u_fine=u2)
torch.cuda.synchronize(); t._lib.profile_enable(True)
for _ in range(3): G(z, c, cam, noise_mode='const', u_coarse=u1, u_fine=u2)
torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
print(os.path.basename(os.environ.get('TDGP_LIB_PATH', 'default')), {k: round(v['avg_ms'], 3) for k, v in r.items() if k in ('triplane_field_kernel', 'merge_composite_kernel', 'importance_from_coarse_kernel')})
