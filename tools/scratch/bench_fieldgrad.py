import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); R = t.renderer
B, F, H, hid, P = 2, 32, 512, 64, 65536 * 64
planes = torch.randn(B, 3 * F, H, H, device='cuda')
mlp = R.TriPlaneMLP(F, hid, 3, 'classical').cuda()
coords = (torch.rand(B, P, 3, device='cuda') - 0.5)
d_rgb, d_sigma = torch.randn(B, P, 3, device='cuda'), torch.randn(B, P, 1, device='cuda')
hw = R.planes_to_hwc(planes)
for _ in range(2): R.simple_tri_plane_renderer_backward(hw, coords, mlp, d_rgb, d_sigma, scale=0.5)
torch.cuda.synchronize(); t._lib.profile_enable(True)
for _ in range(3):
    R.simple_tri_plane_renderer_backward(hw, coords, mlp, d_rgb, d_sigma, scale=0.5)
    R.simple_tri_plane_renderer(hw, coords, mlp, scale=0.5)
torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
print({n: round(v['avg_ms'], 3) for n, v in r.items()}, 'points', B * P)
# ray-march grad
S = 128; Rr = 8 * 65536
col = torch.rand(1, Rr, S, 3, device='cuda'); den = torch.randn(1, Rr, S, 1, device='cuda'); dep = torch.sort(torch.rand(1, Rr, S, 1, device='cuda'), dim=2)[0]
drgb = torch.randn(1, Rr, 3, device='cuda')
for _ in range(2): R.ray_march_backward(col, den, dep, {}, 'classical', drgb)
torch.cuda.synchronize(); t._lib.profile_enable(True)
for _ in range(3): R.ray_march_backward(col, den, dep, {}, 'classical', drgb)
torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
print({n: round(v['avg_ms'], 3) for n, v in r.items()}, 'rays', Rr, 'S', S)
