import sys, os, importlib, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = t.config.config_c3()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=1))
G = G.cuda()
for p in G.parameters(): p.requires_grad_(True)
z = torch.randn(B, cfg.z_dim, device='cuda'); c = torch.zeros(B, cfg.c_dim, device='cuda'); c[:, 0] = 1
cam = dict(angles=torch.tensor([[0.2, 1.4, 0.0]] * B, device='cuda'), fov=torch.full([B], 20.0, device='cuda'), radius=torch.ones(B, device='cuda'),
           look_at=torch.zeros(B, 3, device='cuda'))
def step():
    img = G.forward_autograd(z, c, cam, noise_mode='random')
    loss = torch.nn.functional.softplus(-img).mean()
    grads = torch.autograd.grad(loss, [p for p in G.parameters()], allow_unused=True)
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.time()
n = 3
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f'G forward+backward B={B}: {dt*1e3:.1f} ms / step = {B/dt:.1f} img/s; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
t._lib.profile_enable(True); step(); torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
tot = sum(v['total_ms'] for v in r.values())
for k, v in sorted(r.items(), key=lambda kv: -kv[1]['total_ms'])[:14]:
    print(f"  {k:34s} launches {v['launches']:>4} total {v['total_ms']:8.2f} ms")
print('  HIP kernels total', round(tot, 1), 'ms')
