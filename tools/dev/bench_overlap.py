"""A/B: ToRGB layers on a second stream (SynthesisBlocksSequence.overlap_torgb) vs one stream; C3, B = 16 and 4."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
cfg = t.config.config_c3()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0))
G = G.cuda()
for B in (16, 4):
    inp = t.weights.synthetic_inputs(cfg, batch=B, seed=0)
    T = lambda a: torch.as_tensor(a).cuda()
    x = dict(z=T(inp['z']), c=T(inp['c']), cam={k: T(v) for k, v in inp['camera'].items()}, uc=T(inp['u_coarse']), uf=T(inp['u_fine']))
    ref = None
    for rep in range(2):
        for ov in (False, 16, 32, True):
            G.synthesis.tri_plane_decoder.overlap_torgb = ov
            for _ in range(3):
                img = G(x['z'], x['c'], x['cam'], noise_mode='const', u_coarse=x['uc'], u_fine=x['uf'])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                img = G(x['z'], x['c'], x['cam'], noise_mode='const', u_coarse=x['uc'], u_fine=x['uf'])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            if ref is None:
                ref = img.clone()
            print(f'B {B} overlap {ov}: {dt * 1e3:.3f} ms/step, {B / dt:.1f} img/s, identical {bool(torch.equal(img, ref))}')
