"""Per-layer kernel times of the C3 (or --config) backbone: every conv / ToRGB layer of the tri-plane decoder timed on its own through the
library's event hooks (a synchronisation around each layer: NOT the step time, the split of it).   python tools/dev/bench_layers.py [B] [config]"""
import collections
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t = importlib.import_module('3dgp_amd')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = getattr(t.config, 'config_' + (sys.argv[2] if len(sys.argv) > 2 else 'c3'))()
G = t.generator.Generator(cfg)
G.load_numpy_state_dict(t.weights.random_state_dict(cfg, seed=3))
G = G.cuda()
dec = G.synthesis.tri_plane_decoder
dec.overlap_torgb = False          # every layer on the main stream
inp = t.weights.synthetic_inputs(cfg, B, 0)
ws = G.mapping(torch.as_tensor(inp['z']).cuda(), torch.as_tensor(inp['c']).cuda())
acc = collections.defaultdict(lambda: collections.defaultdict(float))
armed = [False]


def wrap(name, mod):
    fwd = mod.forward

    def timed(*a, **k):
        if not armed[0]:
            return fwd(*a, **k)
        torch.cuda.synchronize()
        t._lib.profile_enable(True)
        out = fwd(*a, **k)
        torch.cuda.synchronize()
        for kn, v in t._lib.profile_report().items():
            acc[name][kn] += v['total_ms']
        t._lib.profile_enable(False)
        return out
    mod.forward = timed


for res in dec.block_resolutions:
    blk = getattr(dec, f'b{res}')
    for ln in ('conv0', 'conv1', 'torgb'):
        if hasattr(blk, ln) and getattr(blk, ln) is not None:
            wrap(f'b{res}.{ln}', getattr(blk, ln))
for _ in range(2):
    dec(ws[:, :dec.num_ws], hwc=True)
armed[0] = True
reps = 3
for _ in range(reps):
    dec(ws[:, :dec.num_ws], hwc=True)
tot = 0.0
for name, ks in acc.items():
    s = sum(ks.values()) / reps
    tot += s
    print(f'{name:14s} {s:7.3f} ms  ' + '  '.join(f'{k}={v / reps:.3f}' for k, v in sorted(ks.items(), key=lambda kv: -kv[1])))
print(f'total {tot:.3f} ms')
