"""Stress (development): N full C3 forwards (B = 16 and 4, ToRGB on the second stream) compared bit for bit with the first -- the hand-counted waits of the F(4x4), ToRGB and field
kernels and the two-role ToRGB kernel's barriers would show a violation as run-to-run differences.   python tools/dev/stress_forward.py [repeats]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = t.config.config_c3()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0))
G = G.cuda()
T = lambda a: torch.as_tensor(a).cuda()          # noqa: E731
for B in (16, 4, 1):
    inp = t.weights.synthetic_inputs(cfg, batch=B, seed=B)
    z, c, cam = T(inp['z']), T(inp['c']), {k: T(v) for k, v in inp['camera'].items()}
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1)
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    uc = torch.rand([B, R, S, 1], generator=gen, device='cuda')
    uf = torch.rand([B * R, S], generator=gen, device='cuda')
    first, bad = None, 0
    for i in range(n):
        img = G(z, c, cam, noise_mode='const', u_coarse=uc, u_fine=uf)
        if first is None:
            first = img.clone()
        elif not torch.equal(first, img):
            bad += 1
    torch.cuda.synchronize()
    t._lib.raise_on_device_fault('stress_forward')
    print(f'B={B}: {n} forwards, {bad} differ from the first; finite: {bool(torch.isfinite(first).all())}')
