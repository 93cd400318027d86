"""One full training iteration at BASELINE configs[2] sizes (patch-wise 64^2 training as in configs/training/base.yaml): phases
Gmain, Dmain, Dreg with Adam, all device work on the library's kernels.  python tools/scratch/bench_train_step.py [batch]"""
import sys, os, importlib, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = t.config.config_c3()
cfg.patch_resolution = 64
cfg.depth_adaptor = t.config.DepthAdaptorConfig()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=1))
G = G.cuda().train()
dcfg = t.discriminator.DiscriminatorConfig(c_dim=cfg.c_dim, patch_params_cond=True, hyper_mod=True)
D = t.discriminator.seeded_discriminator(dcfg, 64, 4, seed=2).cuda().train()
TR = t.training
loss = TR.StyleGAN2Loss(G, D, 'cuda', r1_gamma=1.0, patch_cfg=TR.PatchConfig(resolution=64, min_scale_trg=0.25), use_depth=True)
loss.progressive_update(1000)
optG = torch.optim.Adam(G.parameters(), lr=0.0025, betas=(0.0, 0.99), eps=1e-8)
optD = torch.optim.Adam(D.parameters(), lr=0.002, betas=(0.0, 0.99), eps=1e-8)
z = torch.randn(B, cfg.z_dim, device='cuda'); c = torch.zeros(B, cfg.c_dim, device='cuda'); c[:, 3] = 1
cam = t.metrics.sample_camera_params(t.metrics.camera_base(), B, 'cuda')
real = t.generator.TensorGroup(img=torch.randn(B, 3, 256, 256, device='cuda'), c=c, depth=torch.rand(B, 1, 256, 256, device='cuda'))
gen = t.generator.TensorGroup(z=z, c=c, camera_params=cam)

def phase(name, module, opt):
    for m in (G, D): m.requires_grad_(False)
    opt.zero_grad(set_to_none=True)
    module.requires_grad_(True)
    loss.accumulate_gradients(name, real, gen, gain=1, cur_nimg=1000000)
    module.requires_grad_(False)
    TR.optimizer_step(module, opt, world=1)

def iteration():
    phase('Gmain', G, optG); phase('Dmain', D, optD); phase('Dreg', D, optD)

for _ in range(2): iteration()
torch.cuda.synchronize(); t0 = time.time(); n = 3
for _ in range(n): iteration()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f'training iteration (Gmain + Dmain + Dreg, Adam) B={B}, 64^2 patches: {dt*1e3:.1f} ms = {B/dt:.1f} img/s; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
for name, module, opt in (('Gmain', G, optG), ('Dmain', D, optD), ('Dreg', D, optD)):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): phase(name, module, opt)
    torch.cuda.synchronize(); print(f'  {name}: {(time.time() - t0) / 3 * 1e3:.1f} ms')
print('  losses', {k: float(v.mean()) for k, v in loss.stats.items()})
for name, module, opt in (('Gmain', G, optG), ('Dmain', D, optD), ('Dreg', D, optD)):
    t._lib.profile_enable(True); phase(name, module, opt); torch.cuda.synchronize(); r = t._lib.profile_report(); t._lib.profile_enable(False)
    tot = sum(v['total_ms'] for v in r.values())
    print(f'  {name}: HIP kernels {tot:.1f} ms:', ', '.join(f"{k.replace('_kernel','')} {v['total_ms']:.1f}" for k, v in sorted(r.items(), key=lambda kv: -kv[1]['total_ms'])[:9]))
