"""Wall time of the tri-plane decoder alone vs the sum of its kernels (library profiler), folded x2 layers on / off."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
M = t.ops.modconv
cfg = t.config.config_c3()
G = t.generator.Generator(cfg); G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=0)); G = G.cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
inp = t.weights.synthetic_inputs(cfg, batch=B, seed=0)
ws = G.mapping(torch.as_tensor(inp['z']).cuda(), torch.as_tensor(inp['c']).cuda())
dec = G.synthesis.tri_plane_decoder
for fold in (True, False, True, False):
    M.FOLD_UP2 = fold
    for ov in (True, False):
        dec.overlap_torgb = ov
        for _ in range(3):
            dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
        t._lib.profile_enable(True)
        for _ in range(3):
            dec(ws[:, :dec.num_ws], noise_mode='const', hwc=True)
        torch.cuda.synchronize()
        r = t._lib.profile_report(); t._lib.profile_enable(False)
        ks = sum(v['total_ms'] for v in r.values()) / 3
        print(f'fold={fold} overlap={ov}: wall {wall:.3f} ms, kernel sum {ks:.3f} ms, launches {sum(v["launches"] for v in r.values()) // 3}')
