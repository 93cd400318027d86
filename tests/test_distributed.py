"""N > 1 path on CPU: world-size-2 gloo.  The generator forward never communicates; the only collective is the
FID-style feature all-gather, whose interleaved order must equal the reference's
`torch.stack(ys, dim=1).flatten(0, 1)` (metric_utils.py:145-155)."""
import os
import socket

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D = importlib.import_module('3dgp_amd').distributed
    r, w, _ = D.init_from_env('gloo')
    assert (r, w) == (rank, world)
    num_items, F = 10, 6
    mine = D.shard_items(num_items, rank, world)                  # global item i is produced by rank i % world
    feats = torch.stack([torch.full([F], float(i)) for i in mine])  # "features" of my items, in local order
    g = D.FeatureGatherer(side_stream=False)
    out = g.gather(feats)
    # the reference's exchange: world sequential broadcasts, then stack(dim=1).flatten(0,1)
    ys = []
    for src in range(world):
        y = feats.clone()
        dist.broadcast(y, src=src)
        ys.append(y)
    ref = torch.stack(ys, dim=1).flatten(0, 1)
    ok = torch.equal(out, ref) and torch.equal(out[:, 0], torch.arange(num_items, dtype=torch.float32))
    # async form returns the same block
    g.gather_async(feats)
    ok = ok and torch.equal(g.wait(), ref)
    # FeatureStats.append_torch (metric_utils.py:145-155): every rank accumulates the same interleaved block, max_items truncates
    M = importlib.import_module('3dgp_amd').metrics
    st = M.FeatureStats(capture_all=True, capture_mean_cov=True, max_items=13)
    st.append_torch(feats, num_gpus=world, rank=rank, gatherer=g)
    st.append_torch(feats + 100, num_gpus=world, rank=rank)            # gatherer built on demand; only 3 of these 10 rows fit
    ok = ok and st.num_items == 13 and st.is_full() and bool((st.get_all() == torch.cat([ref, ref[:3] + 100]).numpy()).all())
    # training-side collective (training_loop.py:335-344): one all-reduce over the flat gradient, mean, non-finite values squashed
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    frozen = torch.nn.Parameter(torch.zeros(2))                                   # a parameter without a gradient is skipped
    x = torch.full([5, 4], float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    net[0].bias.grad[0] = float('nan') if rank == 0 else 1.0
    net[0].bias.grad[1] = float('inf')
    flat = D.allreduce_gradients(list(net.parameters()) + [frozen])
    gathered = []
    for src in range(world):
        ys = [t.clone() for t in local]
        for t in ys:
            dist.broadcast(t, src=src)
        gathered.append(ys)
    mean = [sum(gr[i] for gr in gathered) / world for i in range(len(local))]
    ok = ok and flat is not None and flat.numel() == sum(p.numel() for p in net.parameters()) and frozen.grad is None
    ok = ok and torch.allclose(net[0].weight.grad, mean[0]) and torch.allclose(net[1].weight.grad, mean[2]) and torch.allclose(net[1].bias.grad, mean[3])
    ok = ok and float(net[0].bias.grad[0]) == 0.0 and float(net[0].bias.grad[1]) == 1e5 and torch.allclose(net[0].bias.grad[2:], mean[1][2:])
    ok = ok and net[0].weight.grad.data_ptr() == flat.data_ptr()                  # slices of the flat buffer, as in the reference
    q.put((rank, bool(ok), D.rank_seed(3, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_feature_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, 6), (1, True, 7)]                      # seed * world + rank (training_loop.py:73-74)


def test_single_process_helpers(tdgp):
    D = tdgp.distributed
    assert D.shard_items(7, 1, 3) == [1, 4]
    g = D.FeatureGatherer(side_stream=False)
    y = torch.arange(12.).reshape(3, 4)
    assert torch.equal(g.gather(y), y)
    f = D.stand_in_features(torch.randn(2, 3, 64, 64))
    assert f.shape == (2, 2048)
