"""Per-kernel ms of one forward at batch 4 next to a quarter of the batch-16 figure (library profiler): python tools/dev/bench_b4.py [config]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
cfg = getattr(t.config, 'config_' + (sys.argv[1] if len(sys.argv) > 1 else 'c3'))()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0))
G = G.cuda()
G.synthesis.tri_plane_decoder.overlap_torgb = False
res = {}
for B in (16, 4):
    inp = t.weights.synthetic_inputs(cfg, batch=B, seed=0)
    T = lambda a: torch.as_tensor(a).cuda()          # noqa: E731
    z, c, cam = T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}
    for _ in range(3):
        G(z, c, cam, noise_mode='const')
    torch.cuda.synchronize()
    t._lib.profile_enable(True)
    for _ in range(4):
        G(z, c, cam, noise_mode='const')
    torch.cuda.synchronize()
    res[B] = {k: v['total_ms'] / 4 for k, v in t._lib.profile_report().items()}
    t._lib.profile_enable(False)
print('kernel                          B=4     B=16/4   excess')
tot = [0, 0]
for k in sorted(res[4], key=lambda k: -res[4][k]):
    a, b = res[4][k], res[16].get(k, 0) / 4
    tot[0] += a; tot[1] += b
    if a > 0.01:
        print(f'{k:30s} {a:7.3f} {b:7.3f} {a - b:+7.3f}')
print(f'{"sum":30s} {tot[0]:7.3f} {tot[1]:7.3f} {tot[0] - tot[1]:+7.3f}')
