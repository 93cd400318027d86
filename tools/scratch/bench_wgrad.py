import sys, os, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd'); cg = t.ops.conv2d_gradfix
for B, cin, cout, H, k in ((8, 512, 512, 64, 3), (8, 256, 256, 128, 3), (8, 128, 128, 256, 3), (8, 64, 64, 512, 3), (8, 64, 96, 512, 1)):
    x = torch.randn(B, cin, H, H, device='cuda'); dy = torch.randn(B, cout, H, H, device='cuda')
    for _ in range(2): cg.conv2d_weight_grad(x, dy, (cout, cin, k, k), 1, k // 2)
    torch.cuda.synchronize(); t._lib.profile_enable(True)
    for _ in range(3): cg.conv2d_weight_grad(x, dy, (cout, cin, k, k), 1, k // 2)
    torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    us = {n: round(v['avg_ms'] * 1e3, 1) for n, v in r.items()}
    fl = 2.0 * B * cin * cout * k * k * H * H
    print(f'wgrad {cin}->{cout} @{H} k{k}', us, 'TF/s', round(fl / sum(us.values()) / 1e6, 1))
