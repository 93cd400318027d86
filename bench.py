#!/usr/bin/env python3
"""Generator-forward throughput on MI355X: img/s at 256^2, 64 ray steps (BASELINE.json metric), one JSON line.

  python bench.py --gpus N --steps K --warmup W          (N > 1 and no WORLD_SIZE in the environment: starts its own N ranks, one per
                                                           GPU, the way scripts/calc_metrics.py:144-149 / src/train.py:100-105 spawn theirs)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one generator forward (mapping -> tri-plane backbone -> volumetric renderer) over one batch of synthetic
inputs per GPU, inputs and random-init weights resident in HBM before the timed region.  N > 1: one process per GPU,
the batch is sharded by image (weak scaling: per-GPU batch fixed), and the FID-style feature block is all-gathered
over RCCL on a side stream (3dgp_amd/distributed.py) -- the only collective on the path.

Protocol: batch 16 per GPU (configs/scripts/inference.yaml:25) is the headline `value`; batch 4
(training/base.yaml:5) is measured in the same process and reported under `other_batches`; defaults 10 warm-up + 50 timed
steps.  Besides the contract fields the line carries
  roofline     -- the dominant kernel (by HIP-event time measured here, on the launch stream, through the library's
                  per-kernel event hooks: start/stop events attached to each dispatch): algorithmic FLOP per launch / average launch
                  duration vs the fp32 MFMA peak; `traffic` (HBM bytes per launch), `hbm_gbs` (= traffic / the duration
                  measured here) and `mfma_busy_pct` from the committed rocprofv3 PMC passes (profiles/pmc_latest.json,
                  labelled with the commit they were taken at; counters cannot be read from inside this process);
  whole_forward-- all algorithmic FLOP of the step / ms_per_step vs the same peak (the metric's "fraction of roofline");
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's CPU/PyTorch path) timed on the host cores
                  (rank 0, N == 1 only) on a bounded sample of the same workload; cpu_baseline_c1 -- the same for
                  BASELINE configs[0] (64^2 / 32 steps: the reference's own CPU-runnable case), always;
  kernels      -- per-kernel ms/step breakdown of the same profiled steps.
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0


PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
SPEC_CLOCK_GHZ = 2.4              # MI355X_MICROARCH.md: max clock, the clock the peaks above are quoted at


# Fraction of the algorithmic (direct-sum) FLOP a kernel actually executes on the matrix pipe: Winograd F(2x2,3x3) multiplies 16 times
# per 2x2 outputs where the direct sum multiplies 36 times.  `tflops` stays the algorithmic rate; `tflops_executed` is what
# the MFMA roofline bounds.
EXECUTED_FRACTION = {'conv_wino_kernel': 16.0 / 36.0, 'conv_wino4_kernel': 36.0 / 144.0, 'conv_wino4f_kernel': 36.0 / 144.0}        # F(2x2): 16 per 4 outputs; F(4x4): 36 per 16 (direct: 9 per output)


def winograd_takes(batch, cin, cout, r):
    """Mirror of wino_ok() in 3dgp_amd/csrc/modconv.hip: which stride-1 3x3 layers the default arithmetic runs as Winograd."""
    return batch is not None and r % 32 == 0 and cin % 8 == 0 and cin >= 64 and (r // 32) * (r // 8) * batch * ((cout + 63) // 64) >= 256


def winograd4_takes(batch, cin, cout, r, plain=True):
    """Mirror of wino4_shape_ok() in 3dgp_amd/csrc/modconv.hip: the stride-1 3x3 layers the default arithmetic runs as F(4x4,3x3);
    `plain` (not a folded x2 layer): launches with too few items may split the input channels 2 or 4 ways (wino4_ksplit_log2)."""
    if batch is None or not (r % 32 == 0 and cin % 4 == 0 and cin >= 64 and cout >= 64):
        return False
    per_sample = (r * r // 512) * ((cin + 3) // 4) * 18432           # bytes of Winograd-domain input per sample
    items = lambda bs: bs * (r * r // 512) * ((cout + 63) // 64)     # noqa: E731
    if cin <= 128:                                                   # few channels: sub-batches whose V stays inside the Infinity Cache
        sub = min(batch, (192 << 20) // per_sample)
        if sub >= 2 and items(sub) >= 256:
            return True
    sub = min(batch, (4095 << 20) // per_sample)                     # sub-batch whose V fits one buffer descriptor
    if sub >= 1 and items(sub) >= 256:
        return True
    nch = cin // 4                                                   # K split: >= 16 chunks per split, V of the whole batch inside the Infinity Cache
    return plain and batch * per_sample <= (192 << 20) and any((items(batch) << l) >= 256 and nch % (1 << l) == 0 and (nch >> l) >= 16 for l in (1, 2))


def winograd4_fused_takes(batch, cin, cout, r):
    """Mirror of the conv3_wino4f_kernel branch of tdgp_modconv2d (modconv.hip; round 6): the F(4x4) layers with few input channels whose input
    transform runs inside the GEMM kernel (plain stride-1 layers, Cin <= 128, Cin % 16 == 0, Cout % 64 == 0, 64-pixel-wide tile groups, >= 256 items)."""
    return (winograd4_takes(batch, cin, cout, r, plain=False) and cin <= 128 and cin % 16 == 0 and cout % 64 == 0 and r % 64 == 0
            and batch * (r * r // 512) * (cout // 64) >= 256)


def algorithmic_flops(cfg, batch=None):
    """Algorithmic FLOP per image of the MFMA kernels: 2 * MAC of every conv launch (stride-1 3x3 layers in
    conv_mfma_kernel -- or, with `batch` given, conv_wino_kernel for the layers the library runs as Winograd at that batch --, the x2
    layers in upconv_mfma_kernel, ToRGB in torgb_mfma_kernel); 5376 per field point.  With reduced-precision
    blocks (BASELINE configs[4]) the 3x3 layers of those blocks run in conv_bf16_kernel / upconv_bf16_kernel and are priced against
    the bf16 MFMA peak.  -> {kernel label: (flop per image, launches per image batch, peak TFLOP/s)}"""
    ch = cfg.channels
    r16 = cfg.fp16_resolution
    acc = dict(conv_mfma_kernel=[0, 0], conv_wino_kernel=[0, 0], conv_wino4_kernel=[0, 0], conv_wino4f_kernel=[0, 0], upconv_wino4_kernel=[0, 0], upconv_mfma_kernel=[0, 0], torgb_mfma_kernel=[0, 0], conv_bf16_kernel=[0, 0],
               upconv_bf16_kernel=[0, 0])
    for i, r in enumerate(cfg.block_resolutions):
        c = ch[r]
        bf = r16 is not None and r >= r16
        if i > 0:
            # fp32 x2 layers: FIR folded into four parity kernels on the F(4x4) path where the library takes the shape (ops/modconv.py); the four
            # 3x3 parity convolutions execute 4 x 36/16 = 9 multiplies per input pixel -- exactly the algorithmic count of the transposed convolution
            folded = (not bf) and ch[r // 2] >= 64 and (r // 2) ** 2 >= 512 and winograd4_takes(batch, ch[r // 2], 4 * c, r // 2, plain=False)
            k = 'upconv_bf16_kernel' if bf else ('upconv_wino4_kernel' if folded else 'upconv_mfma_kernel')
            acc[k][0] += 2 * ch[r // 2] * c * 9 * (r // 2) ** 2   # stride-2 transposed conv: 9 taps per INPUT pixel
            acc[k][1] += 1
        k = 'conv_bf16_kernel' if bf else ('conv_wino4f_kernel' if winograd4_fused_takes(batch, c, c, r) else
                                             ('conv_wino4_kernel' if winograd4_takes(batch, c, c, r) else
                                              ('conv_wino_kernel' if winograd_takes(batch, c, c, r) else 'conv_mfma_kernel')))
        acc[k][0] += 2 * c * c * 9 * r * r                         # conv1
        acc[k][1] += 1
        acc['torgb_mfma_kernel'][0] += 2 * c * cfg.plane_channels * r * r        # ToRGB 1x1
        acc['torgb_mfma_kernel'][1] += 1
    da = cfg.depth_adaptor
    if da is not None:                                             # --depth-adaptor: 5x5 Conv2dLayers + the 1x1 head of the last layer
        dims = [1] + [da.hid_dim] * da.num_hid_layers
        for cin, cout in zip(dims[:-1], dims[1:]):
            acc['conv_mfma_kernel'][0] += 2 * cin * cout * da.kernel_size ** 2 * cfg.img_resolution ** 2
        acc['conv_mfma_kernel'][0] += 2 * dims[-1] * cfg.img_resolution ** 2
        acc['conv_mfma_kernel'][1] += da.num_hid_layers + 1
    pts = 2 * cfg.img_resolution ** 2 * cfg.num_ray_steps          # coarse + fine
    field = pts * (2 * (cfg.feat_dim * cfg.mlp_hid + 4 * cfg.mlp_hid) + 3 * 4 * 2 * cfg.feat_dim)   # MLP 4608 + bilerp 768 @ (32,64)
    out = {k: (v[0], v[1], PEAK_BF16_MFMA_TFLOPS if 'bf16' in k else PEAK_FP32_MFMA_TFLOPS) for k, v in acc.items() if v[1] > 0}
    out['triplane_field_kernel'] = (field, 2, PEAK_FP32_MFMA_TFLOPS)
    return out


def cpu_baseline(tdgp, cfg, n_img=8, budget_s=14.0):
    """Time the CPU oracle on the same workload: whole generator forwards (mapping, backbone, all 256^2 rays), one image at
    a time, until `n_img` images or ~`budget_s` seconds of host time (whichever first; at least one image)."""
    import oracle as O
    # 2-socket hosts: the OpenMP loops of the oracle stop scaling (and then slow down) past a few dozen threads
    cores = int(os.environ.get('TDGP_ORACLE_THREADS', min(os.cpu_count() or 1, 32)))
    O.set_threads(cores)
    sd = tdgp.weights.random_state_dict(cfg, seed=0)
    done, t_total = 0, 0.0
    while done < n_img and (done == 0 or t_total * (done + 1) / done < budget_s):
        inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=done)
        t0 = time.time()
        img, depth = O.generator_forward(sd, cfg.to_dict(), inp['z'], inp['c'], inp['camera'], inp['u_coarse'], inp['u_fine'])
        if cfg.depth_adaptor is not None:
            O.depth_adaptor_forward(sd, cfg.to_dict(), depth, inp['z'])
        t_total += time.time() - t0
        assert np.isfinite(img).all()
        done += 1
    return dict(value=round(done / t_total, 5), unit='img/s', cores=cores, kind='port',
                thread_cap='min(os.cpu_count(), 32): the oracle\'s OpenMP loops stop scaling past a few dozen threads on 2-socket hosts (TDGP_ORACLE_THREADS overrides)',
                reference_calibration='BASELINE.md section 2: the reference itself (imported, PyTorch CPU ops + its own upfirdn2d / bias_act fallbacks) ran this '
                                      'workload at 0.108 img/s on the 8 cores of the build container; this C oracle on the same 8 cores / 8 threads: 0.113 img/s '
                                      '(measured in round 3) -- the "port" reproduces the reference\'s CPU speed on equal cores',
                sample=f'{done} whole images (mapping + tri-plane backbone + all {cfg.img_resolution ** 2} rays x {cfg.num_ray_steps}+{cfg.num_ray_steps} '
                       f'samples) in {t_total:.1f} s; OpenMP C oracle (oracle/tdgp_oracle.c), {cores} threads')


def host_launch_probe(step, steps=3):
    """Host-side cost of ENQUEUEING one eager step (no synchronisation inside the stop-watch): wall and CPU milliseconds per step.  A few steps
    only, so the launch queue never fills and the host is never blocked on the device.  SURVEY.md 8(e): with one Python process per GPU the
    threat to 8-GPU scaling is this figure growing under contention, not xGMI."""
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(steps):
        step()
    t1, c1 = time.perf_counter(), time.process_time()
    torch.cuda.synchronize()
    return (t1 - t0) / steps * 1e3, (c1 - c0) / steps * 1e3


def with_host_competitors(n, fn):
    """Run fn() while `n` dummy processes spin on OTHER cores than this process's (each pinned to one core outside our affinity set when the box
    has room, else unpinned): what the launch thread of one rank sees when 7 sibling ranks keep their own cores busy.  The children are
    started and killed by PID here."""
    import subprocess
    try:
        mine = sorted(os.sched_getaffinity(0))
        every = list(range(os.cpu_count() or 1))
    except AttributeError:
        mine, every = [], []
    others = [c for c in every if c not in mine] or every
    code = ('import os,sys\n'
            'c=int(sys.argv[1])\n'
            'try:\n os.sched_setaffinity(0,{c}) if c>=0 else None\nexcept OSError:\n pass\n'
            'import array\nb=array.array("d",[1.0])*(1<<20)\ni=0\n'
            'while True:\n i=(i+4099)&((1<<20)-1); b[i]+=1.0\n')            # a cache-unfriendly busy loop: ALU + LLC / memory pressure
    procs = [subprocess.Popen([sys.executable, '-c', code, str(others[i % len(others)] if others else -1)]) for i in range(n)]
    try:
        time.sleep(0.3)
        return fn()
    finally:
        for p_ in procs:
            p_.kill()
        for p_ in procs:
            p_.wait()


def with_real_siblings(n, batch, fn, timeout_s=120.0):
    """Run fn() while `n` REAL sibling ranks -- separate Python processes, each with its own HIP context, allocator, streams and launch thread, running
    C3 forwards of `batch` items on THIS GPU (tools/one_gpu_ranks.py load) -- are active: what the launch thread of one rank sees next to live siblings,
    as far as one GPU can show it (the spinning dummies of with_host_competitors model busy cores only).  The siblings share the
    device, so GPU times measured inside fn() are not this process's alone: only host-side figures are taken from it.  Children are started and ended
    by PID; returns (fn's result or None, info)."""
    import select
    import subprocess
    procs, info = [], dict(siblings=n, sibling_batch=batch)
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r + 1), WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0', TDGP_ONE_GPU_CONFIG=os.environ.get('TDGP_ONE_GPU_CONFIG', 'c3'))
            env.pop('MASTER_PORT', None)
            procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tools', 'one_gpu_ranks.py'), 'load', str(batch)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                          text=True, env=env))
        deadline = time.time() + timeout_s
        for p_ in procs:                             # wait for every sibling's `ready`
            ok = False
            while time.time() < deadline and p_.poll() is None:
                if select.select([p_.stdout], [], [], 1.0)[0]:
                    if 'ready' in p_.stdout.readline():
                        ok = True
                        break
            if not ok:
                info['error'] = 'a sibling did not come up'
                return None, info
        time.sleep(0.5)
        t0 = time.time()
        out = fn()
        info['measured_for_s'] = round(time.time() - t0, 2)
        info['siblings_alive'] = sum(p_.poll() is None for p_ in procs)
        return out, info
    except Exception as e:                            # noqa: BLE001  (a probe must never take the bench line down)
        info['error'] = repr(e)[:200]
        return None, info
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
        for p_ in procs:
            try:
                p_.wait(timeout=30)
            except Exception:                         # noqa: BLE001
                pass


def parity_statement(config):
    """What `-m gpu` asserts for this configuration and the figures the last committed GPU test run met (profiles/parity_latest.json =
    gpurun_out/parity_report.json of that run; tests/conftest.py:assert_image_parity defines every figure)."""
    st = dict(stated='<= 1e-4 max-rel RGB vs the reference CPU/PyTorch path (north_star); INT rows bit-exact on identical inputs',
              asserted='tests/test_gpu_parity.py::test_full_size_vs_reference_golden[c1|c2|c3]: range-normalised error <= max(1e-5, 2 x the reference\'s own fp32-vs-float64 '
                       'figure); per-pixel max-rel (SURVEY 9.9) <= max(1e-4, 1.5 x the largest of the reference\'s own three noise figures: re-run, vs exactly rounded, vs its float64 run); '
                       'mean error vs the float64 reference <= 1.25 x the reference\'s own; every INT-row mismatch through the chain explained by a cdf-knot window',
              why_not_flat_1e4='at 512-channel / 512^2 size the reference\'s OWN fp32 image is 1.5e-3 ... 3.2e-3 per pixel (5e-6 ... 7e-6 of the range) from its own float64 run '
                               '(tests/golden/e2e_full_*.npz): a flat 1e-4 per-pixel figure is not a property of the reference path itself')
    path = os.path.join(REPO, 'profiles', 'parity_latest.json')
    # rows of the parity report that belong to THIS configuration (c4 / c5 lines used to carry c3's rows): c1..c4 have a
    # full-size golden from the reference's fp32 run, c5 the reference's own bf16 run; a configuration without rows gets no 'met' entry
    prefix = {'c1': 'c1 full size', 'c2': 'c2 full size', 'c3': 'c3 full size', 'c4': 'c4 full size', 'c5': 'bf16 C5'}.get(config)
    if config == 'c5':
        st['asserted'] = ('tests/test_gpu_parity.py::test_bf16_full_size_vs_reference_golden: tri-planes / image / depth vs the reference\'s own bfloat16 run '
                          '(tests/golden/bf16_full_c5.npz): max <= 1.5e-2, mean <= 2e-3 of the tri-plane scale (SURVEY 9.9: ~2^-8 per reduced-precision layer)')
    if prefix and os.path.exists(path):
        rows = [r for r in json.load(open(path)) if r.get('what', '').startswith(prefix)]
        if rows:
            st['met'] = {r['what']: {k: v for k, v in r.items() if k != 'what'} for r in rows}
            st['met_source'] = 'profiles/parity_latest.json (the -m gpu run committed with this tree)'
    return st


def timed_steps(step, barrier, steps, warmup, world, dev, finish=None, per_rank=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.  `per_rank` (a list) receives
    every rank's OWN time for the K steps -- stop-watch read after its own synchronize, BEFORE the closing barrier -- so that a straggler
    shows in the line instead of hiding behind the maximum."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        img = step()
    if finish is not None:
        finish()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        if per_rank is not None:
            mine = torch.tensor([own], dtype=torch.float64, device=dev)
            every = torch.empty([world], dtype=torch.float64, device=dev)
            torch.distributed.all_gather_into_tensor(every, mine)
            per_rank[:] = [float(v) for v in every.tolist()]
    elif per_rank is not None:
        per_rank[:] = [own]
    assert torch.isfinite(img).all()
    return elapsed


def straggler_figures(per_rank_s, steps):
    """-> (per-rank ms per step, slowest / fastest)."""
    ms = [round(v / steps * 1e3, 3) for v in per_rank_s]
    return ms, (round(max(ms) / min(ms), 4) if ms and min(ms) > 0 else None)


def spawn_ranks(n, script=None, argv=None):
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT as torchrun would set them), pass rank 0's JSON line through, fail if any rank fails.  The
    parent never touches a GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    # Host threads: N Python ranks each defaulting to every host core (input synthesis, ATen micro-kernels, the OpenMP pools) oversubscribe the
    # box N-fold; torchrun pins OMP_NUM_THREADS to 1, here each rank gets its share of the cores (and pins itself to the cores of its GPU's
    # NUMA node once it knows its device: distributed.pin_rank).  A caller's own OMP_NUM_THREADS wins.
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    omp = os.environ.get('OMP_NUM_THREADS') or str(max(1, cores // n))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS=omp, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    try:
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                rcs[r] = p.poll()
            if any(rc not in (None, 0) for rc in rcs):      # one rank died: the others would sit in a collective for ever
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
    if bad:
        sys.exit(f'bench.py: rank(s) failed: {bad}')


def parse_args(argv=None):
    """The command line of bench.py (the driver's: `--gpus N --steps K --warmup W`)."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU per step of the headline value (16 = configs/scripts/inference.yaml:25)')
    ap.add_argument('--other-batches', default='4', help='comma list of further per-GPU batch sizes measured in the same run (4 = training/base.yaml:5); "" = none')
    ap.add_argument('--config', default='c3', choices=['c1', 'c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cmax', type=int, default=0, help='override the backbone width (cbase = 64 x cmax): `--config c5 --cmax 1024` is the bf16 configuration as BASELINE.md section 3 sizes it')
    ap.add_argument('--no-host-probe', action='store_true', help='skip the host-side launch-cost probe (host_launch_ms / launch_bound_margin)')
    ap.add_argument('--no-real-siblings', action='store_true', help='host probe without the 3 real sibling processes on the same GPU (host_launch.batch_4.*_real_siblings)')
    ap.add_argument('--depth-adaptor', action='store_true', help='also run the DepthAdaptor inside G.forward (SURVEY 8f rank 1; off = the 8a hot path)')
    ap.add_argument('--profile-steps', type=int, default=3)
    ap.add_argument('--chunk', type=int, default=-1, help='samples per pass through the high-resolution blocks + renderer (Infinity-Cache-sized working set); '
                                                           '0 = whole batch through every kernel; -1 = the package default')
    ap.add_argument('--chunk-from', type=int, default=0, help='first block resolution that runs chunked (0 = the package default)')
    ap.add_argument('--graph', action='store_true', help='headline = replays of the forward captured as one HIP graph (3dgp_amd/graphs.py) instead of eager launches '
                                                          '(one host call per kernel).  The other mode is timed and reported next to it either way; measured r03: the two '
                                                          'agree to 0.3 % at B = 16 and B = 4 -- the forward is not launch-bound')
    ap.add_argument('--fid-lanes', type=int, default=1, help='--fid-loop: run the 16 sub-batch forwards of a 64-image block as this many concurrent graph replays (one '
                    'captured graph and one stream per lane); 1 = one after the other, the order the reference issues them in')
    ap.add_argument('--no-fid-loop', action='store_true', help='N > 1 times the FID-loop shape by default (it is the multi-GPU workload: one all-gather per 64 images per rank); this switches it off')
    ap.add_argument('--batch-gen', default='4,16', help='--fid-loop: comma list of generator sub-batch sizes (MetricOptions.batch_gen, metric_utils.py:26,289: a caller option, '
                                                         'default min(batch_size, 4)); the first one is `fid_loop`, the others are listed under `fid_loop.other_batch_gen`')
    ap.add_argument('--fid-loop', action='store_true', help="also time the reference's FID generation loop shape (metric_utils.py:288-319): 64 images per rank as 16 "
                                                             'sub-batches of 4 with device-side draws, one feature block (and, N > 1, one RCCL all-gather) per 64')
    ap.add_argument('--arith', default='f32', choices=['f32', 'direct', 'split'],
                    help="arithmetic of the large 3x3 convolutions: f32 = fp32 MFMA (default, the reported metric); split = opt-in 3 x bf16 split operands, "
                         "6 piece products, fp32 accumulation (fp32-grade results; reported with dtype 'bf16x3->f32' and never mixed with the default line)")
    args = ap.parse_args(argv)
    args.no_graph = not args.graph
    return args


def setup_rank(args, tdgp, backend='nccl', need_gpu=True):
    """One process per GPU: rank / world / local rank from the launcher's environment (torchrun's or spawn_ranks'), process group on `backend`
    (nccl == RCCL on ROCm), device = cuda:LOCAL_RANK, host threads pinned to the cores of that GPU's NUMA node.  `backend='gloo',
    need_gpu=False` is the CPU rehearsal of exactly this code (tests/test_distributed.py runs the driver's command shape through it)."""
    D = tdgp.distributed
    rank, world, local_rank = D.init_from_env(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if need_gpu:
        assert torch.cuda.is_available(), 'bench.py needs a GPU'
        assert local_rank < torch.cuda.device_count(), f'LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible (HIP_VISIBLE_DEVICES={os.environ.get("HIP_VISIBLE_DEVICES")})'
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    else:
        dev = torch.device('cpu')
    host = D.pin_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world))) if world > 1 else None      # cores of this GPU's NUMA node
    return rank, world, local_rank, dev, host


def runs_cpu_baseline(args, world, rank):
    """The CPU-baseline leg runs on rank 0 of a ONE-GPU job only (the contract: N = 1; at N > 1 eight copies of a 32-thread OpenMP run
    would fight the launch threads for the host)."""
    return world == 1 and rank == 0 and not args.no_cpu_baseline


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        assert torch.cuda.is_available() and torch.cuda.device_count() >= args.gpus, \
            f'--gpus {args.gpus} but {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible'
        return spawn_ranks(args.gpus)

    tdgp = importlib.import_module('3dgp_amd')
    D = tdgp.distributed
    rank, world, local_rank, dev, host = setup_rank(args, tdgp)
    if not args.no_fid_loop:
        args.fid_loop = True                # N > 1: it IS the multi-GPU workload; N == 1: the FID forward with and without the adaptors
    tdgp._lib.load()                       # the HIP library must be there: no fallback
    if args.arith == 'split':
        tdgp._lib.set_conv_arith(1)
    elif args.arith == 'direct':            # fp32 MFMA with direct sums in every 3x3 layer (no Winograd): the A/B of the default
        tdgp._lib.set_conv_arith(2)

    cfg = getattr(tdgp.config, f'config_{args.config}')()
    if args.cmax > 0:
        cfg.cmax, cfg.cbase = args.cmax, 64 * args.cmax
    if args.depth_adaptor:
        cfg.depth_adaptor = tdgp.config.DepthAdaptorConfig()
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=0))        # random-init weights, identical on every rank
    G = G.to(dev)
    if args.chunk >= 0:
        G.synthesis.chunk = args.chunk or None
    if args.chunk_from > 0:
        G.synthesis.chunk_from = args.chunk_from
    T = lambda a: torch.as_tensor(a).to(dev)   # noqa: E731
    gather = D.FeatureGatherer() if world > 1 else None
    # what RCCL actually saw: an all-reduce of ones over the process group, AFTER a real collective (not WORLD_SIZE from the env)
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == torch.distributed.get_world_size() == world

    def inputs(batch):
        inp = tdgp.weights.synthetic_inputs(cfg, batch=batch, seed=D.rank_seed(0, rank, world))
        return dict(z=T(inp['z']), c=T(inp['c']), cam={k: T(v) for k, v in inp['camera'].items()}, u_coarse=T(inp['u_coarse']), u_fine=T(inp['u_fine']))

    graphs = {}

    def graphed(batch):
        """The captured forward for this batch size, its static buffers holding the (resident) synthetic inputs."""
        if batch not in graphs:
            graphs[batch] = tdgp.graphs.GraphedGenerator(G, batch, noise_mode='const', explicit_draws=True)
        return graphs[batch]

    def make_step(x, use_graph=None):
        use_graph = (not args.no_graph) if use_graph is None else use_graph
        if use_graph:
            gg = graphed(x['z'].shape[0])
            gg.load(x['z'], x['c'], x['cam'], x['u_coarse'], x['u_fine'])          # inputs resident in the graph's buffers before the timed region
        def step():
            img = gg.replay() if use_graph else G(x['z'], x['c'], x['cam'], noise_mode='const', u_coarse=x['u_coarse'], u_fine=x['u_fine'])
            if gather is not None:
                if gather._pending is not None:
                    gather.wait()
                gather.gather_async(D.stand_in_features(img))
            return img
        return step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def finish():
        if gather is not None and gather._pending is not None:
            gather.wait()

    x = inputs(args.batch)
    # ---- host-side launch cost: is the eager forward launch-bound, alone and beside 7 busy sibling processes? ----
    host_probe = None
    if not args.no_host_probe:
        eager = make_step(x, use_graph=False)
        for _ in range(3):
            eager()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eager()
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) / 3 * 1e3
        wall_ms, cpu_ms = host_launch_probe(eager)
        host_probe = dict(host_launch_ms=round(wall_ms, 3), host_cpu_ms=round(cpu_ms, 3), gpu_step_ms=round(gpu_ms, 3), launch_bound_margin=round(gpu_ms / max(wall_ms, 1e-6), 2),
                          batch_per_gpu=args.batch, note='enqueue time of one eager forward (every kernel a host call, no sync inside the stop-watch) vs its GPU time; '
                          'margin = gpu_step_ms / host_launch_ms: > 1 means the host stays ahead of the device')
        if world == 1:
            w8, c8 = with_host_competitors(7, lambda: host_launch_probe(eager))
            host_probe.update(host_launch_ms_7_competitors=round(w8, 3), host_cpu_ms_7_competitors=round(c8, 3), launch_bound_margin_7_competitors=round(gpu_ms / max(w8, 1e-6), 2))
            xb4 = inputs(4)
            e4 = make_step(xb4, use_graph=False)
            for _ in range(3):
                e4()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                e4()
            torch.cuda.synchronize()
            g4 = (time.perf_counter() - t0) / 3 * 1e3
            w4, _c4 = host_launch_probe(e4)
            w48, _ = with_host_competitors(7, lambda: host_launch_probe(e4))
            host_probe['batch_4'] = dict(host_launch_ms=round(w4, 3), gpu_step_ms=round(g4, 3), launch_bound_margin=round(g4 / max(w4, 1e-6), 2),
                                         host_launch_ms_7_competitors=round(w48, 3), launch_bound_margin_7_competitors=round(g4 / max(w48, 1e-6), 2))
            del xb4, e4
        else:
            # every rank must take the same launch mode: the smallest margin over the ranks decides
            m = torch.tensor([host_probe['launch_bound_margin']], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MIN)
            host_probe['launch_bound_margin_min_over_ranks'] = round(float(m.item()), 2)
            if float(m.item()) < 2.0 and not args.graph:
                args.graph, args.no_graph = True, False            # launch-bound within a factor 2: replay the captured forward instead
                host_probe['decision'] = 'margin < 2 on some rank: headline switched to HIP graph replay'
            else:
                host_probe['decision'] = 'eager kept' if not args.graph else 'graph requested'
    per_rank_s = []
    elapsed = timed_steps(make_step(x), barrier, args.steps, args.warmup, world, dev, finish, per_rank=per_rank_s)
    # the other launch mode next to the headline (eager when the headline replays a graph, and the other way round)
    alt_elapsed = timed_steps(make_step(x, use_graph=args.no_graph), barrier, args.steps, args.warmup, world, dev, finish)

    # ---- per-kernel timing of the same step (HIP events on the launch stream, inside the library) -------------------
    # While profiling is on the library launches through hipExtLaunchKernelGGL with a start and a stop event attached to each
    # dispatch: the pair reads the kernel's own begin/end timestamps (what rocprofv3's kernel trace reports), with no marker
    # packets around it -- per-kernel averages are directly comparable with profiles/*_kernel_stats.md.
    # (per-kernel durations are taken on ONE stream: in the timed steps the ToRGB layers run on a second stream beside the next block's x2
    #  layer, and two kernels sharing the chip would each be charged the other's time)
    dec = G.synthesis.tri_plane_decoder
    overlap_was, dec.overlap_torgb = dec.overlap_torgb, False
    for _ in range(2):                      # back to steady state after the other launch mode's capture / replays
        G(x['z'], x['c'], x['cam'], noise_mode='const', u_coarse=x['u_coarse'], u_fine=x['u_fine'])
    tdgp._lib.profile_enable(True)
    torch.cuda.synchronize()
    t_prof = time.perf_counter()
    for _ in range(args.profile_steps):
        G(x['z'], x['c'], x['cam'], noise_mode='const', u_coarse=x['u_coarse'], u_fine=x['u_fine'])
    torch.cuda.synchronize()
    # wall time of the PROFILED steps: a dispatch that carries its own start / stop signals is a little slower than a plain one, so
    # the per-kernel sum is to be held against this, not against the un-instrumented `ms_per_step`
    profiled_step_ms = (time.perf_counter() - t_prof) / max(args.profile_steps, 1) * 1e3
    prof = tdgp._lib.profile_report()
    tdgp._lib.profile_enable(False)
    dec.overlap_torgb = overlap_was
    nprof = max(args.profile_steps, 1)
    kernels = {}
    for k, v in prof.items():
        tot = v['total_ms']
        kernels[k] = dict(ms_per_step=round(tot / nprof, 4), launches_per_step=v['launches'] // nprof, avg_ms=round(tot / max(v['launches'], 1), 5))
    kernels = dict(sorted(kernels.items(), key=lambda kv: -kv[1]['ms_per_step']))
    dominant = next(iter(kernels), None)
    flops = algorithmic_flops(cfg, args.batch if args.arith == 'f32' else None)
    # HBM bytes per launch and matrix-pipe busy from the committed PMC passes (tools/profile_round.sh -> profiles/pmc_latest.json);
    # only used when they were taken on this very workload.
    pmc, pmc_src = {}, None
    ppath = os.path.join(REPO, 'profiles', 'pmc_latest.json')
    if os.path.exists(ppath):
        pj = json.load(open(ppath))
        if pj.get('config') == args.config and pj.get('batch_per_gpu') == args.batch and not args.depth_adaptor and args.arith == 'f32':
            pmc, pmc_src = pj.get('kernels', {}), dict(file='profiles/pmc_latest.json', commit=pj.get('commit'), passes=pj.get('source'))
    roofline = None
    if dominant in flops:
        fl_img, launches_img, peak = flops[dominant]
        k = kernels[dominant]
        fl_per_launch = fl_img * args.batch / k['launches_per_step']
        achieved = fl_per_launch / (k['avg_ms'] * 1e-3) / 1e12 * EXECUTED_FRACTION.get(dominant, 1.0)      # what the matrix pipe executes
        pk = pmc.get(dominant, {})
        traffic = pk.get('hbm_bytes_per_launch')
        roofline = dict(kernel=dominant, bound='mfma', achieved=round(achieved, 3), peak=peak, unit='TFLOP/s',
                        frac=round(achieved / peak, 4), traffic=traffic,
                        hbm_gbs=None if traffic is None else round(traffic / (k['avg_ms'] * 1e-3) / 1e9, 1), mfma_busy_pct=pk.get('mfma_busy_pct'),
                        l2_hit_pct=pk.get('l2_hit_pct'), traffic_fetch=pk.get('fetch_bytes_per_launch'), traffic_write=pk.get('write_bytes_per_launch'),
                        traffic_note=None if traffic is None else 'memory-side bytes of the L2 per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE counts '
                                     '128-B requests at 64 B: profiles/r03_pmc_calibration.md); Infinity-Cache hits included',
                        pmc_source=pmc_src, flop_per_launch=fl_per_launch, avg_launch_ms=k['avg_ms'], launches_per_step=k['launches_per_step'])
        # The shader clock the kernel actually sustains (GRBM_GUI_ACTIVE / 8 XCDs / duration, from the same committed counter pass): the chip runs to its
        # power budget, not to the 2.4 GHz the 157.3 TFLOP/s peak is quoted at.  `frac` stays against the spec peak (the metric's figure);
        # `frac_at_clock` = the same rate against the peak AT the sustained clock -- the share of the issue slots the kernel's structure leaves unused.
        if pk.get('sclk_ghz'):
            roofline.update(sclk_ghz=pk['sclk_ghz'], spec_clock_ghz=SPEC_CLOCK_GHZ, peak_at_clock=round(peak * pk['sclk_ghz'] / SPEC_CLOCK_GHZ, 1),
                            frac_at_clock=round(achieved / (peak * pk['sclk_ghz'] / SPEC_CLOCK_GHZ), 4),
                            sclk_note='sclk_ghz is from the committed counter pass (profiles/pmc_latest.json: profiled dispatches clock ~2-3 % lower than un-profiled ones), not measured in this process')
    for k, v in kernels.items():               # the same three figures for every kernel the PMC passes cover
        pk = pmc.get(k)
        if pk and v['avg_ms'] > 0:
            v.update(hbm_gbs=round(pk['hbm_bytes_per_launch'] / (v['avg_ms'] * 1e-3) / 1e9, 1), mfma_busy_pct=pk.get('mfma_busy_pct'))
            if pk.get('sclk_ghz'):
                v['sclk_ghz'] = pk['sclk_ghz']
            if 'l2_hit_pct' in pk:
                v['l2_hit_pct'] = pk['l2_hit_pct']
        if k in flops and v['avg_ms'] > 0:
            v['tflops'] = round(flops[k][0] * args.batch / v['launches_per_step'] / (v['avg_ms'] * 1e-3) / 1e12, 2)
            if k in EXECUTED_FRACTION:
                v['tflops_executed'] = round(v['tflops'] * EXECUTED_FRACTION[k], 2)
            # against the kernel's OWN matrix peak (fp32 157.3 / bf16 2500 TFLOP/s), on the FLOP it executes
            v['peak_tflops'] = flops[k][2]
            v['frac_of_own_peak'] = round(v.get('tflops_executed', v['tflops']) / flops[k][2], 4)
    total_flop_img = sum(f for f, _, _ in flops.values())
    # what the matrix pipe EXECUTES: the Winograd layers multiply 16/36 of their algorithmic (direct-sum) FLOP.  `frac` (algorithmic) is
    # the metric's figure; `frac_executed` is the one an MFMA roofline bounds -- both are printed, neither alone.
    exec_flop_img = sum(f * EXECUTED_FRACTION.get(k, 1.0) for k, (f, _, _) in flops.items())
    ms_step = elapsed / args.steps * 1e3
    busy_w = [(v['ms_per_step'], v['mfma_busy_pct']) for v in kernels.values() if v.get('mfma_busy_pct') is not None]
    t_all = sum(v['ms_per_step'] for v in kernels.values())
    whole = dict(flop_per_image=total_flop_img, achieved=round(total_flop_img * args.batch / (ms_step * 1e-3) / 1e12, 2), peak=PEAK_FP32_MFMA_TFLOPS,
                 unit='TFLOP/s', frac=round(total_flop_img * args.batch / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                 flop_executed_per_image=round(exec_flop_img), achieved_executed=round(exec_flop_img * args.batch / (ms_step * 1e-3) / 1e12, 2),
                 frac_executed=round(exec_flop_img * args.batch / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                 # matrix-pipe busy over the whole step: per-kernel SQ_VALU_MFMA_BUSY (committed PMC pass) weighted by the kernel times measured
                 # here; kernels without a counter row (ATen / rocBLAS micro-kernels) count as 0 % busy
                 mfma_busy_pct_time_weighted=round(sum(t * b for t, b in busy_w) / t_all, 1) if busy_w and t_all > 0 else None,
                 kernel_ms_sum=round(sum(v['ms_per_step'] for v in kernels.values()), 3), profiled_step_ms=round(profiled_step_ms, 3))

    if cfg.fp16_resolution:
        # BASELINE configs[4]: bf16 and fp32 FLOP divided by the fp32 peak is not a roofline -- the per-kernel
        # `frac_of_own_peak` figures are the meaningful ones; the aggregate keeps the achieved rate only
        for k_ in ('frac', 'frac_executed', 'peak'):
            whole[k_] = None
        whole['note'] = 'mixed bf16 / fp32 matrix work: no single peak; see kernels[*].frac_of_own_peak'
    others = {}
    for b in [int(t) for t in args.other_batches.split(',') if t.strip()]:
        if b == args.batch:
            continue
        xb = inputs(b)
        eb = timed_steps(make_step(xb), barrier, args.steps, args.warmup, world, dev, finish)
        eb_alt = timed_steps(make_step(xb, use_graph=args.no_graph), barrier, args.steps, args.warmup, world, dev, finish)
        others[str(b)] = dict(value=round(b * world * args.steps / eb, 3), ms_per_step=round(eb / args.steps * 1e3, 3), batch_per_gpu=b, steps=args.steps,
                              frac_of_fp32_mfma_ceiling=None if cfg.fp16_resolution else round(total_flop_img * b / (eb / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                              launch='eager' if args.no_graph else 'hip graph replay',
                              **{('graph_value' if args.no_graph else 'eager_value'): round(b * world * args.steps / eb_alt, 3)})
        del xb

    fid_loop = None
    if args.fid_loop:
        # metric_utils.py:288-319 (compute_feature_stats_for_generator): batch_size 64 per rank, generated as 64 // batch_gen forwards of
        # batch_gen images with noise_mode='random' and the renderer's own draws, concatenated, passed through the detector (stand-in: a fixed
        # pooling to [64, 2048], the Inception pickle is a URL download) and appended -- one all-gather of the [64, 2048] block per 64 images.
        # batch_gen is MetricOptions.batch_gen (a caller option; the reference's default is min(batch_size, 4)): both 4 and 16 are timed.
        per = 64
        lanes = max(1, args.fid_lanes) if not args.no_graph else 1

        def time_fid(gen, Gx=None, cam_adaptor=False):
            Gx = G if Gx is None else Gx
            gen = tdgp.metrics.resolve_batch_gen(per, gen)
            ggs = [tdgp.graphs.GraphedGenerator(Gx, gen, noise_mode='random', explicit_draws=False) for _ in range(lanes)] if not args.no_graph else []
            gg = ggs[0] if ggs else None
            lane_streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)] if lanes > 1 else []
            zs = [inputs(gen) for _ in range(2)]

            def fid_step():
                imgs = []
                if lanes > 1:
                    # the block's forwards are independent: `lanes` of them are in flight at a time, each a replay of its lane's own graph on its
                    # lane's stream -- the low-resolution layers of a 4-image forward fill a fraction of the chip, the lanes fill the rest
                    main = torch.cuda.current_stream(dev)
                    for st in lane_streams:
                        st.wait_stream(main)
                    for i in range(per // gen):
                        xi = zs[i & 1]
                        with torch.cuda.stream(lane_streams[i % lanes]):
                            imgs.append(ggs[i % lanes](xi['z'], xi['c'], xi['cam']).clone())
                    for st in lane_streams:
                        main.wait_stream(st)
                else:
                    for i in range(per // gen):
                        xi = zs[i & 1]
                        cam_i = xi['cam']
                        if cam_adaptor:                                   # metric_utils.py:305-307: the camera adaptor runs on the sampled cameras first
                            cam_i = Gx.synthesis.camera_adaptor(cam_i, xi['z'], xi['c'])
                        if gg is not None:
                            imgs.append(gg(xi['z'], xi['c'], cam_i).clone())
                        else:
                            imgs.append(Gx(xi['z'], xi['c'], cam_i, noise_mode='random'))
                feats = D.stand_in_features(torch.cat(imgs))
                if gather is not None:
                    if gather._pending is not None:
                        gather.wait()
                    gather.gather_async(feats)
                return imgs[-1]
            nfid = max(args.steps // 8, 3)
            pr = []
            ef = timed_steps(fid_step, barrier, nfid, 1, world, dev, finish, per_rank=pr)
            ms, strag = straggler_figures(pr, nfid)
            return dict(value=round(per * world * nfid / ef, 3), unit='img/s', images_per_rank_per_step=per, batch_gen=gen, sub_batch=gen, steps=nfid,
                        ms_per_64=round(ef / nfid * 1e3, 3), noise_mode='random', draws='device (inside the graph)' if gg is not None else 'device',
                        launch='hip graph replay' if gg is not None else 'eager', lanes=lanes, per_rank_ms_per_64=ms, straggler_ratio=strag,
                        reference='metric_utils.py:288-319; batch_gen = MetricOptions.batch_gen (:26,289), reference default min(batch_size, 4) = 4')
        gens = [int(t) for t in args.batch_gen.split(',') if t.strip()] or [4]
        fid_loop = time_fid(gens[0])
        fid_loop['other_batch_gen'] = {str(g_): time_fid(g_) for g_ in gens[1:]}
        if cfg.depth_adaptor is None and lanes == 1:
            # The FID forward AS THE REFERENCE RUNS IT (metric_utils.py:305-310): depth and camera adaptors are enabled in every 3dgp config
            # (configs/training/base.yaml:7,9), so G.forward evaluates the DepthAdaptor (27 GFLOP per 256^2 image) and multiplies it by 0.0
            # (networks_epigraf.py:253).  `elided` = this package's default (the adaptor is skipped when nothing consumes it; same bits, see
            # SynthesisNetwork.forward), `literal` = strict_nan_propagation (the reference's exact op sequence).
            import copy
            cfg_ad = copy.deepcopy(cfg)
            cfg_ad.depth_adaptor, cfg_ad.camera_adaptor = tdgp.config.DepthAdaptorConfig(), tdgp.config.CameraAdaptorConfig()
            G_ad = tdgp.generator.Generator(cfg_ad)
            G_ad.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg_ad, seed=0))
            G_ad = G_ad.to(dev)
            ad = {}
            for name, strict in (('elided', False), ('literal', True)):
                G_ad.synthesis.strict_nan_propagation = strict
                ad[name] = {str(g_): {k_: v_ for k_, v_ in time_fid(g_, G_ad, cam_adaptor=True).items() if k_ in ('value', 'ms_per_64', 'batch_gen', 'launch')} for g_ in gens}
            base = {str(gens[0]): fid_loop['value'], **{k_: v_['value'] for k_, v_ in fid_loop['other_batch_gen'].items()}}
            ad['vs_adaptor_off_pct'] = {k_: dict(elided=round(100.0 * (ad['elided'][k_]['value'] / base[k_] - 1.0), 2), literal=round(100.0 * (ad['literal'][k_]['value'] / base[k_] - 1.0), 2)) for k_ in base}
            fid_loop['adaptors'] = ad
            del G_ad

    if host_probe is not None and world == 1 and not args.no_real_siblings:
        # LAST, after every timed measurement of this process (round 6: it used to run before the headline's timed region, and once a fresh box gave 27.6 ms
        # per step there against 22.1 in the same process's own probes and graph replays -- the killed siblings' HIP contexts are torn down by the driver
        # asynchronously; nothing timed may follow them).  Next to 3 REAL sibling ranks on this very GPU (own HIP contexts, allocators, launch threads; C3
        # forwards at the reference's FID batch of 4): only the HOST figure is meaningful -- the device is shared.
        xb4 = inputs(4)
        e4 = make_step(xb4, use_graph=False)
        for _ in range(3):
            e4()
        torch.cuda.synchronize()
        g4 = host_probe['batch_4']['gpu_step_ms']
        r4, sib = with_real_siblings(3, 4, lambda: host_launch_probe(e4, steps=6))
        if r4 is not None:
            host_probe['batch_4'].update(host_launch_ms_3_real_siblings=round(r4[0], 3), host_cpu_ms_3_real_siblings=round(r4[1], 3),
                                         launch_bound_margin_3_real_siblings=round(g4 / max(r4[0], 1e-6), 2))
        host_probe['real_siblings'] = dict(sib, note='3 sibling processes (tools/one_gpu_ranks.py load) with their own HIP contexts running C3 forwards of 4 items on the same '
                                           'GPU while this process enqueues its batch-4 forwards; margin = this process\'s un-shared GPU step time / that enqueue time; '
                                           'measured after every timed region of this line')
        del xb4, e4

    if rank == 0:
        total_imgs = args.batch * world * args.steps
        names = dict(c1='BASELINE configs[0]: SDFood-like 64x64', c2='BASELINE configs[1]: Dogs 128x128', c3='BASELINE configs[2]: ImageNet 256x256',
                     c4='BASELINE configs[3]: ImageNet 256x256', c5='BASELINE configs[4]: ImageNet 256x256')
        out = {
            'metric': 'generator-forward img/s @256^2, 64 steps' if args.config in ('c3', 'c4') else (
                'generator-forward img/s @256^2, 96 steps, bf16 blocks' if args.config == 'c5' else f'generator-forward img/s ({args.config})'),
            'value': round(total_imgs / elapsed, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('bf16 (backbone blocks >= %d^2: bf16 activations + weights, fp32 accumulate), f32 elsewhere' % cfg.fp16_resolution) if cfg.fp16_resolution
                     else ('f32' if args.arith in ('f32', 'direct') else 'bf16x3->f32 (3x3 and x2 layers), f32 elsewhere'), 'data': 'synthetic',
            'config': {'workload': f'{names[args.config]}, {cfg.num_ray_steps}(+{cfg.num_ray_steps}) ray steps, cmax {cfg.cmax}, '
                                   f'c_dim {cfg.c_dim}, tri-plane {cfg.tri_plane_res}^2 x {cfg.plane_channels}, full HIP path',
                       'batch_per_gpu': args.batch, 'global_batch': args.batch * world, 'img_resolution': cfg.img_resolution,
                       'num_ray_steps': cfg.num_ray_steps, 'depth_adaptor': bool(args.depth_adaptor), 'parallelism': f'dp{world} (batch-sharded, weights replicated)',
                       'schedule': dict(chunk=G.synthesis.chunk, chunk_from=G.synthesis.chunk_from, torgb_on_second_stream=bool(overlap_was))},
            'launch': 'eager' if args.no_graph else 'hip graph replay (3dgp_amd/graphs.py: every kernel of the forward, one submission per step)',
            ('graph_value' if args.no_graph else 'eager_value'): round(total_imgs / alt_elapsed, 3), 'fid_loop': fid_loop,
            'rccl_ranks_seen': ranks_seen, 'per_rank_ms': straggler_figures(per_rank_s, args.steps)[0], 'straggler_ratio': straggler_figures(per_rank_s, args.steps)[1],
            'host': None if host is None else dict(rank0=host, omp_num_threads=os.environ.get('OMP_NUM_THREADS')), 'host_launch': host_probe,
            'parity_tolerance': parity_statement(args.config), 'roofline': roofline, 'whole_forward': whole, 'other_batches': others, 'kernels': kernels,
        }
        if runs_cpu_baseline(args, world, rank):
            out['cpu_baseline'] = cpu_baseline(tdgp, cfg)
            out['cpu_baseline_c1'] = out['cpu_baseline'] if args.config == 'c1' else cpu_baseline(tdgp, tdgp.config.config_c1(), n_img=16, budget_s=8.0)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
