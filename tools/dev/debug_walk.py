"""Which points differ between the image walk and the linear-order kernel?  python tools/dev/with_lib.py <so> tools/dev/debug_walk.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
T = lambda a: torch.as_tensor(np.asarray(a, np.float32)).cuda()
B, h, w, S, F, hid = [int(x) for x in (sys.argv[1:7] if len(sys.argv) > 6 else (2, 16, 24, 16, 32, 64))]
rs = np.random.RandomState(S * 7 + h)
H = 48
planes = T(rs.randn(B, 3 * F, H, H))
mlp = t.renderer.TriPlaneMLP(F, hid).cuda()
cam = dict(angles=T(np.stack([rs.uniform(-1, 1, B), rs.uniform(1.0, 2.0, B), np.zeros(B)], 1)), radius=T(np.ones(B)), look_at=T(np.zeros((B, 3))))
ro, rd = t.renderer.sample_rays(t.renderer.compute_cam2world_matrix(cam), T(rs.uniform(15, 45, B)), (w, h))
R = h * w
tt = T(np.sort(rs.uniform(0.4, 1.6, (B, R, S)), axis=2))
hw = t.renderer.planes_to_hwc(planes)
mp = t.renderer._mlp_params(mlp)
for rep in range(3):
    lin = t.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=tt, ray_w=0).cpu().numpy().reshape(B, h, w, S, 4)
    img = t.renderer._field(hw, mp, 0.5, ray_o=ro, ray_d=rd, t=tt, ray_w=w).cpu().numpy().reshape(B, h, w, S, 4)
    bad = (lin != img).any(-1)
    print('rep', rep, 'bad points', int(bad.sum()), 'of', bad.size)
    if bad.any():
        b_, y_, x_, s_ = np.nonzero(bad)
        print('  images', sorted(set(b_)), 'rows', sorted(set(y_)), 'cols', sorted(set(x_)), 'samples', sorted(set(s_)))
        print('  per (b, patch row, patch col):', sorted({(int(a), int(b) // 8, int(c) // 8) for a, b, c in zip(b_, y_, x_)}))
        print('  bad per sample index:', [int(bad[..., k].sum()) for k in range(S)])
