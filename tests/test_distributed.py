"""N > 1 path on CPU: world-size-2 gloo.  The generator forward never communicates; the only collective is the
FID-style feature all-gather, whose interleaved order must equal the reference's
`torch.stack(ys, dim=1).flatten(0, 1)` (metric_utils.py:145-155)."""
import os
import socket
import sys

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


# ------------------------------------------------------------------------------------------------ RCCL (needs GPUs)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RCCL_ONE_RANK = '''
import importlib, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)            # "nccl" is RCCL on ROCm
D = importlib.import_module("3dgp_amd").distributed
g = D.FeatureGatherer(side_stream=True)
assert g.collective and g.stream is not None
outs = []
for step in range(4):                                            # the bench's pattern: wait for the previous block, launch the next
    y = torch.full([64, 2048], float(step), device="cuda") + torch.arange(64, device="cuda")[:, None]
    if g._pending is not None:
        outs.append(g.wait())
    g.gather_async(y)
    del y                                                        # record_stream keeps the block alive for the side stream
outs.append(g.wait())
torch.cuda.synchronize()
for step, o in enumerate(outs):
    assert o.shape == (64, 2048) and float(o[5, 7]) == step + 5, (step, float(o[5, 7]))
one = torch.ones(1, device="cuda")
dist.all_reduce(one)
assert int(one.item()) == dist.get_world_size() == 1
net = torch.nn.Linear(8, 4).cuda()
net(torch.ones(3, 8, device="cuda")).sum().backward()
flat = D.allreduce_gradients(net.parameters())
assert flat is not None and flat.is_cuda and flat.numel() == 36
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
'''


@pytest.mark.gpu
def test_rccl_single_rank_collectives():
    """The RCCL code path on whatever GPU is there: a one-rank NCCL group still builds a communicator and runs the side-stream
    all_gather_into_tensor (+ record_stream), the all-reduce behind `rccl_ranks_seen`, the flat-gradient all-reduce and a barrier."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-c', _RCCL_ONE_RANK % (REPO, str(_free_port()))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and 'rccl ok' in out.stdout, out.stderr[-3000:]


def _start_one_gpu_ranks(world, mode, extra=(), env_extra=None):
    """Start `world` copies of tools/one_gpu_ranks.py, all mapped to GPU 0, as separate processes (returned with their PIDs: the caller ends them)."""
    import subprocess
    import sys
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HIP_VISIBLE_DEVICES='0',
                   HSA_ENABLE_IPC_MODE_LEGACY='0', **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tools', 'one_gpu_ranks.py'), mode, *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    return procs


@pytest.mark.gpu
def test_four_ranks_share_one_gpu(tmp_path):
    """N ranks with real HIP contexts.  Four processes (tools/one_gpu_ranks.py), each with its own context / allocator / streams / launch
    thread on cuda:0, a gloo process group, three real forwards of four C3 items each (the reference's FID batch_gen, metric_utils.py:289), feature blocks
    gathered through `FeatureGatherer`.  Asserted: per-rank seeds seed * world + rank (training_loop.py:73-74) and distinct RNG streams; every rank holds the
    same gathered block; its interleave (item i from rank i % world, metric_utils.py:154) equals a SINGLE process generating the same items alone afterwards,
    bit for bit -- no cross-rank interference through the allocator, the streams or the device; no device fault; `pin_rank` worked on a real box.  The
    per-rank host enqueue times with all four launch threads active at once go into the parity report."""
    import json
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    world = 4
    base = str(tmp_path / 'ranks')
    procs = _start_one_gpu_ranks(world, 'check', (base,))
    try:
        outs = [p.communicate(timeout=900) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r}: {se[-3000:]}'
    res = [json.load(open(f'{base}.rank{r}.json')) for r in range(world)]
    assert [x['rank'] for x in res] == list(range(world)) and all(x['world'] == world for x in res)
    assert [x['seed'] for x in res] == [5 * world + r for r in range(world)]
    assert len({x['first_random'] for x in res}) == world                     # four different RNG streams
    assert all(x['same_block_on_every_rank'] for x in res)
    assert res[0]['equals_single_process'], res[0]['max_abs_diff_vs_single_process']
    assert all(res[0]['items_in_order'])
    assert all(x['pin']['threads'] is None or x['pin']['threads'] >= 1 for x in res)
    from conftest import report_parity
    enq = [round(float(np.mean(x['enqueue_ms'])), 3) for x in res]
    stp = [round(float(np.mean(x['step_ms'])), 3) for x in res]
    report_parity('4 ranks on one GPU (real HIP contexts, gloo), C3 B = 4: host enqueue ms per forward by rank, all ranks launching at once', rank0=enq[0], rank1=enq[1],
                  rank2=enq[2], rank3=enq[3], step_ms_max=max(stp), cpus_per_rank=res[0]['pin']['cpus'] or 0, mem_mb_per_rank=res[0]['mem_mb'])


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """bench.py --gpus 2 under torchrun, the driver's launch line: runs only where two GPUs are visible (the round-end 8-GPU node)."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip(f'{torch.cuda.device_count()} GPU(s) visible: the two-rank RCCL run needs 2')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2', '--other-batches', '', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks_seen'] == 2 and line['config']['global_batch'] == 4 and line['value'] > 0


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks_over_rccl():
    """Plain `python bench.py --gpus 2` (no launcher, no WORLD_SIZE): bench.py starts its two ranks itself and rank 0 reports what RCCL saw."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip(f'{torch.cuda.device_count()} GPU(s) visible: the two-rank RCCL run needs 2')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    cmd = [sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2', '--other-batches', '', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks_seen'] == 2 and line['config']['global_batch'] == 4 and line['value'] > 0


_SPAWNED_CHILD = '''
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.ones(1)
dist.all_reduce(t)
if sys.argv[1:] == ["--die"] and rank == 1:
    os._exit(3)
dist.barrier()
omp = torch.tensor([float(os.environ.get("OMP_NUM_THREADS", "-1"))])
lo, hi = omp.clone(), omp.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN)
dist.all_reduce(hi, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps(dict(ranks_seen=int(t.item()), argv=sys.argv[1:], omp=[int(lo.item()), int(hi.item())])))
dist.destroy_process_group()          # (without it gloo's threads occasionally abort at interpreter exit: "terminate called without an active exception")
'''

# The driver's multi-GPU command, rehearsed on CPU: the child runs bench.py's OWN argument parser and rank set-up (bench.parse_args,
# bench.setup_rank with backend gloo / no GPU) under the environment bench.spawn_ranks -- or torchrun -- gives it.
_DRIVER_SHAPE_CHILD = '''
import importlib, json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import bench
args = bench.parse_args()
tdgp = importlib.import_module("3dgp_amd")
rank, world, local_rank, dev, host = bench.setup_rank(args, tdgp, backend="gloo", need_gpu=False)
seen = torch.ones(1); dist.all_reduce(seen)
lr = torch.zeros(world); lr[rank] = local_rank; dist.all_reduce(lr)
base = torch.tensor([1.0 if bench.runs_cpu_baseline(args, world, rank) else 0.0]); dist.all_reduce(base)
seeds = torch.zeros(world); seeds[rank] = tdgp.distributed.rank_seed(0, rank, world); dist.all_reduce(seeds)
if rank == 0:
    print(json.dumps(dict(gpus=args.gpus, steps=args.steps, warmup=args.warmup, batch=args.batch, fid_loop_default=not args.no_fid_loop, ranks_seen=int(seen.item()),
                          local_ranks=[int(v) for v in lr.tolist()], cpu_baseline_ranks=int(base.item()), seeds=[int(v) for v in seeds.tolist()],
                          master=[os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])], pinned=host is not None,
                          hsa_ipc=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))))
dist.destroy_process_group()
''' % REPO


def test_bench_spawn_ranks_plumbing(tmp_path):
    """bench.spawn_ranks on CPU: N children with torchrun's environment, rank 0's line passes through, a dead rank fails the launch
    instead of hanging it."""
    import json
    import subprocess
    import sys
    child = tmp_path / 'child.py'
    child.write_text(_SPAWNED_CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.pop('OMP_NUM_THREADS', None)
    cores = len(os.sched_getaffinity(0))
    for n in (2, 8):                       # 8 = the node the driver's scaling run uses: every rank gets its share of the host cores
        drive = 'import sys; sys.path.insert(0, %r); import bench; bench.spawn_ranks(%d, script=%r, argv=sys.argv[1:])' % (REPO, n, str(child))
        out = subprocess.run([sys.executable, '-c', drive, '--x', '1'], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
        assert line == dict(ranks_seen=n, argv=['--x', '1'], omp=[max(1, cores // n)] * 2)
    out = subprocess.run([sys.executable, '-c', drive, '--x', '1'], capture_output=True, text=True, timeout=600, env=dict(env, OMP_NUM_THREADS='3'))
    assert json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])['omp'] == [3, 3]          # a caller's own setting wins
    out = subprocess.run([sys.executable, '-c', drive, '--die'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and 'rank(s) failed' in out.stderr


def test_driver_command_shape_on_cpu(tmp_path):
    """the driver's exact multi-GPU command shapes, rehearsed on CPU through bench.py's own parser and rank set-up:
      (a) `python bench.py --gpus 8 --steps 20 --warmup 5`  -> bench.spawn_ranks: 8 children, free port on 127.0.0.1, LOCAL_RANK i <-> GPU i;
      (b) `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`.
    Checked: argument values, world size seen by a real collective, the LOCAL_RANK -> device mapping, per-rank seeds (seed * world + rank), the
    cpu_baseline leg on NO rank at N > 1, the HSA IPC mode exported to every child."""
    import json
    import socket
    import subprocess
    import sys
    child = tmp_path / 'bench_child.py'
    child.write_text(_DRIVER_SHAPE_CHILD)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    argv = ['--gpus', '8', '--steps', '20', '--warmup', '5']
    drive = 'import sys; sys.path.insert(0, %r); import bench; bench.spawn_ranks(8, script=%r, argv=sys.argv[1:])' % (REPO, str(child))
    out = subprocess.run([sys.executable, '-c', drive] + argv, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert (line['gpus'], line['steps'], line['warmup'], line['batch']) == (8, 20, 5, 16) and line['fid_loop_default'] is True
    assert line['ranks_seen'] == 8 and line['local_ranks'] == list(range(8)) and line['seeds'] == list(range(8))
    assert line['cpu_baseline_ranks'] == 0 and line['master'][0] == '127.0.0.1' and line['pinned'] is True and line['hsa_ipc'] == '0'
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           str(child), '--gpus', '2', '--steps', '20', '--warmup', '5']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['gpus'] == 2 and line['ranks_seen'] == 2 and line['local_ranks'] == [0, 1] and line['cpu_baseline_ranks'] == 0 and line['master'] == ['127.0.0.1', port]
    # N = 1: the parser's defaults finish within minutes and the baseline leg belongs to rank 0
    sys.path.insert(0, REPO)
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup) == (1, 50, 10) and bench.runs_cpu_baseline(a, 1, 0) and not bench.runs_cpu_baseline(a, 8, 0)


def test_host_launch_probe_plumbing():
    """bench.host_launch_probe / with_host_competitors on CPU (torch.cuda.synchronize stubbed): wall and CPU milliseconds per step come back,
    the competitor processes really run while the probe does, and every one of them is gone afterwards (started and killed by PID)."""
    import time
    import psutil
    sys.path.insert(0, REPO)
    import bench
    real = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        before = {p.pid for p in psutil.Process().children(recursive=True)}
        seen = []

        def probe():
            seen.append(len({p.pid for p in psutil.Process().children(recursive=True)} - before))
            return bench.host_launch_probe(lambda: time.sleep(0.002), steps=3)
        wall, cpu = bench.with_host_competitors(3, probe)
        assert seen == [3] and 1.5 <= wall <= 50.0 and 0.0 <= cpu <= wall + 5.0
        deadline = time.time() + 10
        while time.time() < deadline and ({p.pid for p in psutil.Process().children(recursive=True)} - before):
            time.sleep(0.05)
        assert not ({p.pid for p in psutil.Process().children(recursive=True)} - before)
    finally:
        torch.cuda.synchronize = real


def test_rank_cpus_follow_the_gpu_numa_node(tdgp):
    """distributed.rank_cpus: ranks whose GPUs share a NUMA node split that node's cores; without NUMA information an even split of the
    allowed set; never an empty set, never a core outside what the process may use."""
    D = tdgp.distributed
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    got = [D.rank_cpus(r, 8, nodes, allowed=range(128), node_cpus=node_cpus) for r in range(8)]
    assert got[0] == list(range(0, 16)) and got[3] == list(range(48, 64)) and got[4] == list(range(64, 80)) and got[7] == list(range(112, 128))
    assert sorted(c for g in got for c in g) == list(range(128))
    flat = [D.rank_cpus(r, 4, [-1] * 4, allowed=range(10)) for r in range(4)]
    assert flat == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert D.rank_cpus(1, 2, [5, 5], allowed=range(8), node_cpus={}) == [4, 5, 6, 7]           # node without a cpulist -> even split
    assert D.rank_cpus(0, 2, [0, 1], allowed=[3], node_cpus={0: [0, 1], 1: [2, 3]}) == [3]       # nothing of node 0 allowed -> what is allowed
    assert D._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    info = D.pin_rank(0, 1, set_threads=False)               # on this box: must not raise, affinity stays a subset of what it was
    assert info['cpus'] is None or info['cpus'] >= 1


def _timed_steps_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import time
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.synchronize = lambda *a, **k: None             # CPU plumbing test of bench.timed_steps
    per = []

    def step():
        time.sleep(0.02 * (1 + 2 * rank))                     # rank 1 is the straggler
        return torch.ones(1)
    el = bench.timed_steps(step, dist.barrier, 3, 1, world, 'cpu', per_rank=per)
    q.put((rank, el, per))
    dist.destroy_process_group()


def test_bench_reports_every_rank(tdgp):
    """bench.timed_steps hands back each rank's OWN time next to the max-over-ranks figure (a straggler is visible in the line)."""
    import socket
    sys.path.insert(0, REPO)
    import bench
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_timed_steps_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    (_, el0, per0), (_, el1, per1) = res
    assert per0 == per1 and len(per0) == 2 and per0[1] > 1.5 * per0[0] > 0
    assert abs(el0 - el1) < 1e-9 and el0 >= max(per0) - 1e-3
    ms, ratio = bench.straggler_figures(per0, 3)
    assert ratio > 1.5 and ms[1] > ms[0]
