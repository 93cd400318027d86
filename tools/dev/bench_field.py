"""Renderer-only timing at the C3 shape (random planes): per-kernel min/max ms through the library's event hooks.
   python tools/dev/bench_field.py [B]            (variant libraries: python tools/dev/with_lib.py <so> tools/dev/bench_field.py ...)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
R = t.renderer
cfg = t.config.config_c3()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inp = t.weights.synthetic_inputs(cfg, B, 0)
dev = lambda a: torch.as_tensor(a).cuda()      # noqa: E731
planes = R.HWCPlanes(torch.randn(B, 3, 512, 512, 32, device='cuda'))
mlp = R.TriPlaneMLP(32, 64).cuda()
c2w = R.compute_cam2world_matrix({k: dev(v) for k, v in inp['camera'].items()})
ro, rd = R.sample_rays(c2w, dev(inp['camera']['fov']), (256, 256))
rend = R.ImportanceRenderer('classical')
opts = dict(box_size=1.0, num_proposal_steps=64, num_fine_steps=64, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
            u_coarse=dev(inp['u_coarse']), u_fine=dev(inp['u_fine']), ray_grid_w=256)
for _ in range(2):
    rend(planes, mlp, ro, rd, opts)
torch.cuda.synchronize()
t._lib.profile_enable(True)
for _ in range(reps):
    rend(planes, mlp, ro, rd, opts)
torch.cuda.synchronize()
r = t._lib.profile_report()
t._lib.profile_enable(False)
print('B', B, t._lib.LIB_PATH, {k: (round(v['min_ms'], 3), round(v['avg_ms'], 3)) for k, v in r.items()})
