import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); R = t.renderer
B, F, H, hid, S, hw = 2, 32, 512, 64, 64, 256
planes = torch.randn(B, 3 * F, H, H, device='cuda')
mlp = R.TriPlaneMLP(F, hid, 3, 'classical').cuda()
cam = dict(angles=torch.tensor([[0.2, 1.4, 0.0]] * B, device='cuda'), radius=torch.ones(B, device='cuda'), look_at=torch.zeros(B, 3, device='cuda'))
ro, rd = R.sample_rays(R.compute_cam2world_matrix(cam), torch.full([B], 20.0, device='cuda'), (hw, hw))
tt = torch.linspace(0.75, 1.25, S, device='cuda').view(1, 1, S, 1).expand(B, hw * hw, S, 1)
coords = (ro.unsqueeze(-2) + tt * rd.unsqueeze(-2)).reshape(B, -1, 3).contiguous()
P = coords.shape[1]
d_rgb, d_sigma = torch.randn(B, P, 3, device='cuda'), torch.randn(B, P, 1, device='cuda')
hwc = R.planes_to_hwc(planes)
for pg in (True, False):
    for _ in range(2): R.simple_tri_plane_renderer_backward(hwc, coords, mlp, d_rgb, d_sigma, scale=0.5, planes_grad=pg)
    torch.cuda.synchronize(); t._lib.profile_enable(True)
    for _ in range(3): R.simple_tri_plane_renderer_backward(hwc, coords, mlp, d_rgb, d_sigma, scale=0.5, planes_grad=pg)
    torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    print('planes_grad', pg, {n: round(v['avg_ms'], 3) for n, v in r.items()}, 'points', B * P)
