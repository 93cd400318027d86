#!/usr/bin/env python3
"""N ranks with REAL HIP contexts on ONE GPU: what a 1-GPU lease can still say about the multi-process path.

The reference spawns one process per GPU (calc_metrics.py:144-149), every rank generates its own items (item i -> rank i % world,
metric_utils.py:154; per-rank seed seed * world + rank, training_loop.py:73-74) and the feature blocks are exchanged per batch
(metric_utils.py:145-155).  The CPU tests run that plumbing on gloo without a GPU; the RCCL tests need >= 2 GPUs.  This script is the
piece in between: every rank is a separate Python process with its own HIP context, allocator, streams and launch thread -- all mapped to
cuda:0 --, the process group is gloo (RCCL cannot put two ranks on one device), features are staged through the host.

  mode `check` (tests/test_distributed.py::test_four_ranks_share_one_gpu):
      each rank runs ROUNDS real forwards of BATCH items of its shard at the C3 shape, gathers the feature blocks, rank 0 re-generates EVERY
      rank's batches alone afterwards and compares the interleaved block bit for bit; per-rank host enqueue times (all ranks launching at once,
      released by a barrier) go into the JSON written to argv[2].
  mode `load`  (bench.py: host_launch.*_real_siblings):
      run forwards until killed; prints `ready` once warm.  The bench process measures its own enqueue time next to these siblings.

Launch: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (tests and bench.py start the processes by PID and end them by PID).
"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

ROUNDS, BATCH, SEED = 3, 4, 5


def item_inputs(tdgp, cfg, items, dev):
    """Inputs of the global items `items`, each a function of its own index only (so that any process can re-generate any item)."""
    z = np.stack([np.random.RandomState(1000 + i).randn(cfg.z_dim) for i in items]).astype(np.float32)
    c = np.zeros((len(items), cfg.c_dim), np.float32)
    if cfg.c_dim > 0:
        c[np.arange(len(items)), [i % cfg.c_dim for i in items]] = 1.0
    cams = [np.random.RandomState(2000 + i) for i in items]
    cam = dict(angles=np.stack([[g.uniform(-1.5, 1.5), g.uniform(0.8, 2.3), 0.0] for g in cams]).astype(np.float32),
               fov=np.array([g.uniform(10.0, 45.0) for g in cams], np.float32), radius=np.ones(len(items), np.float32),
               look_at=np.zeros((len(items), 3), np.float32))
    R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
    gen = torch.Generator(device=dev)
    uc, uf = [], []
    for i in items:                                  # the renderer's draws: device generator seeded by the item
        gen.manual_seed(3000 + i)
        uc.append(torch.rand([R, S], generator=gen, device=dev))
        uf.append(torch.rand([R, S], generator=gen, device=dev))
    T = lambda a: torch.as_tensor(a).to(dev)        # noqa: E731
    return dict(z=T(z), c=T(c), cam={k: T(v) for k, v in cam.items()}, u_coarse=torch.stack(uc).unsqueeze(-1), u_fine=torch.stack(uf).reshape(len(items) * R, S))


def rank_batches(rank, world):
    """Round j of rank r: the BATCH items (j * BATCH + k) * world + r -- in the gathered, interleaved block of that round they sit at k * world + r."""
    return [[(j * BATCH + k) * world + rank for k in range(BATCH)] for j in range(ROUNDS)]


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    tdgp = importlib.import_module('3dgp_amd')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    pin = tdgp.distributed.pin_rank(rank, world) if mode == 'check' else None        # (load mode: a lone process, placed by the OS like any sibling rank of another job)
    cfg = getattr(tdgp.config, 'config_' + os.environ.get('TDGP_ONE_GPU_CONFIG', 'c3'))()
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=SEED))
    G = G.to(dev)

    def forward(x):
        return G(x['z'], x['c'], x['cam'], noise_mode='const', u_coarse=x['u_coarse'], u_fine=x['u_fine'])

    if mode == 'load':
        batch = int(sys.argv[2]) if len(sys.argv) > 2 else BATCH
        x = item_inputs(tdgp, cfg, list(range(rank * batch, (rank + 1) * batch)), dev)
        for _ in range(3):
            forward(x)
        torch.cuda.synchronize()
        print('ready', flush=True)
        while True:                                  # ended by PID from the parent
            for _ in range(4):
                forward(x)
            torch.cuda.synchronize()

    assert mode == 'check'
    gather = tdgp.distributed.FeatureGatherer(side_stream=False)
    seed = tdgp.distributed.rank_seed(SEED, rank, world)
    torch.manual_seed(seed)
    batches = rank_batches(rank, world)
    xs = [item_inputs(tdgp, cfg, b, dev) for b in batches]
    for _ in range(2):                               # warm: weight packing, workspace allocation
        forward(xs[0])
    torch.cuda.synchronize()
    gathered, enqueue_ms, step_ms = [], [], []
    for x in xs:
        if world > 1:
            dist.barrier()                           # every rank launches its forward at the same moment: N launch threads, N HIP contexts, one GPU
        t0 = time.perf_counter()
        img = forward(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enqueue_ms.append((t1 - t0) * 1e3)
        step_ms.append((t2 - t0) * 1e3)
        feats = tdgp.distributed.stand_in_features(img, 2048).cpu()          # staged through the host: gloo
        gathered.append(gather.gather(feats))
    tdgp._lib.raise_on_device_fault('one_gpu_ranks')
    out = dict(rank=rank, world=world, seed=seed, pin=pin, enqueue_ms=enqueue_ms, step_ms=step_ms, device=torch.cuda.get_device_name(0),
               first_random=float(torch.rand(1).item()), mem_mb=round(torch.cuda.max_memory_allocated() / 2 ** 20, 1))
    got = torch.stack(gathered)                      # [ROUNDS, BATCH * world, 2048]
    if world > 1:                                    # every rank holds the same gathered block
        ref = got.clone()
        dist.broadcast(ref, 0)
        out['same_block_on_every_rank'] = bool(torch.equal(ref, got))
        dist.barrier()
    if rank == 0:
        # alone on the GPU now (the others wait at the barrier below): every rank's batches again, interleaved by hand
        want = torch.empty_like(got)
        for r in range(world):
            for j, b in enumerate(rank_batches(r, world)):
                f = tdgp.distributed.stand_in_features(forward(item_inputs(tdgp, cfg, b, dev)), 2048).cpu()
                want[j, r::world] = f
        out['equals_single_process'] = bool(torch.equal(want, got))
        out['max_abs_diff_vs_single_process'] = float((want - got).abs().max())
        out['items_in_order'] = [[(j * BATCH + k) * world + r for k in range(BATCH) for r in range(world)] == list(range(j * BATCH * world, (j + 1) * BATCH * world))
                                 for j in range(ROUNDS)]
    if world > 1:
        dist.barrier()
    with open(sys.argv[2] + f'.rank{rank}.json', 'w') as f:
        json.dump(out, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
