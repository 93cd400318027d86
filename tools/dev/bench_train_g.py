"""Generator forward + backward on the differentiable path (G.forward_autograd) at BASELINE configs[2], per-kernel times from the library profiler:
   python tools/dev/bench_train_g.py [B=4] [steps=5]     (development tool)"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = t.config.config_c3()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0))
G = G.cuda()
for p in G.parameters():
    p.requires_grad_(True)
ps = list(G.parameters())
inp = t.weights.synthetic_inputs(cfg, batch=B, seed=0)
T = lambda a: torch.as_tensor(a).cuda()          # noqa: E731
z, c, cam = T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}


def step():
    img = G.forward_autograd(z, c, cam, noise_mode='random')
    loss = torch.nn.functional.softplus(-img).mean()
    return torch.autograd.grad(loss, ps, allow_unused=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
t._lib.profile_enable(True)
for _ in range(2):
    step()
torch.cuda.synchronize()
rep = t._lib.profile_report()
t._lib.profile_enable(False)
ks = sorted(((v['total_ms'] / 2, k, v['launches'] // 2 if 'launches' in v else 0) for k, v in rep.items()), reverse=True)
print(f'{os.path.basename(t._lib.LIB_PATH)} B={B}: {ms:.1f} ms per generator forward+backward = {B / ms * 1e3:.1f} img/s, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; '
      f'HIP kernels {sum(k[0] for k in ks):.1f} ms: ' + ', '.join(f'{k[1].replace("_kernel", "")} {k[0]:.2f}' for k in ks[:14]))
