#!/usr/bin/env python3
"""Fold the three PMC tables written by tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE;
separate rocprofv3 --pmc passes) into per-launch figures per kernel family (all template instantiations of a kernel pooled, the
way bench.py's event timing pools them): HBM bytes per launch and matrix-pipe busy %.

  python tools/pmc_fold.py gpurun_out/r02 gpurun_out/r02/pmc.json --bench gpurun_out/r02/bench.json
  (review, then copy to profiles/pmc_latest.json with the commit the passes were taken at)

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB.  Calibrated in round 3 against known byte
counts (tools/pmc_calibrate.hip, profiles/r03_pmc_calibration.md: 2 GiB buffers read / written exactly once with 4-, 8- and 16-B
loads per lane, whole-line and half-line gathers, stores of both widths): **FETCH_SIZE reports exactly half the bytes for EVERY read
shape** -- the L2 issues 128-B read requests (TCC_EA0_RDREQ = bytes / 128, TCC_BUBBLE = TCC_EA0_RDREQ_32B = 0) and rocprofv3's gfx950
expression tallies them at 64 B -- and WRITE_SIZE is exact (64-B write requests).  So fetch bytes = 2 x FETCH_SIZE x 1024 for every
kernel (FETCH_FACTOR below; round 2's "4-B/lane loads match uncorrected" was wrong: those kernels over-fetch 2x), cross-checked per
kernel against TCC_EA0_RDREQ_sum x 128 when that pass is present.  These are bytes at the L2's memory side: Infinity-Cache hits are
included, so this is an upper bound of the HBM bytes.  L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).  MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD instances x 256 CUs x 4 SIMDs), the gfx94x MfmaUtil formula (ROCm 7.2 ships no
gfx950 derived-counter section).
"""
import argparse
import collections
import json
import os
import re
import subprocess


FETCH_FACTOR = 2.0      # calibrated: profiles/r03_pmc_calibration.md

# device function name -> the label the library's event timing (and bench.py) pools it under
ALIAS = {'conv3_mfma_kernel': 'conv_mfma_kernel', 'conv3_wino_kernel': 'conv_wino_kernel', 'conv3_wino4_kernel': 'conv_wino4_kernel', 'conv3_wino4f_kernel': 'conv_wino4f_kernel', 'conv3_bf16_kernel': 'conv_bf16_kernel', 'triplane_walk_kernel': 'triplane_field_kernel', 'triplane_walk2_kernel': 'triplane_field_kernel', 'conv3s_mfma_kernel': 'conv_mfma_kernel', 'upconv3s_mfma_kernel': 'upconv_mfma_kernel'}


def table(path):
    """rocpd_pmc.py table -> {kernel: (calls, avg_us, {counter: avg value})}."""
    rows, cols = {}, None
    if not os.path.exists(path):
        return rows
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip('|').split('|')]
        if cells and cells[0] == 'kernel':
            cols = cells[3:]
        elif cols and len(cells) == 3 + len(cols) and cells[0].startswith('`'):
            rows[cells[0].strip('`')] = (int(cells[1]), float(cells[2]), {c: float(v) for c, v in zip(cols, cells[3:])})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('indir')
    ap.add_argument('out')
    ap.add_argument('--bench', default=None, help='bench.json of the same run: config / batch are read from it')
    a = ap.parse_args()
    cfg, batch = 'c3', None
    if a.bench and os.path.exists(a.bench):
        try:
            bj = json.loads(open(a.bench).read().strip().splitlines()[-1])
            batch = bj['config']['batch_per_gpu']
            cfg = {'configs[0]': 'c1', 'configs[1]': 'c2', 'configs[2]': 'c3', 'configs[3]': 'c4', 'configs[4]': 'c5'}.get(
                re.search(r'configs\[\d\]', bj['config']['workload']).group(0), 'c3')
        except Exception as e:        # noqa: BLE001
            print('bench.json not parsed:', e)
    def family(name):
        # the two instances of the F(4x4) kernel are different layers (x2 folded vs plain 3x3): keep them apart
        if name.startswith('conv3_wino4_kernel<true'):
            return 'upconv_wino4_kernel'
        base = name.split('<')[0].strip()
        return ALIAS.get(base, base)

    fam = collections.defaultdict(lambda: dict(launches=0, fetch=0.0, write=0.0, mfma=0.0, active=0.0, mlaunches=0, mdur_us=0.0, hit=0.0, miss=0.0, req=0.0, rdreq=0.0, rlaunches=0))
    for key, fn, counter in (('fetch', 'pmc_FETCH_SIZE.md', 'FETCH_SIZE'), ('write', 'pmc_WRITE_SIZE.md', 'WRITE_SIZE')):
        for name, (calls, _, cs) in table(os.path.join(a.indir, fn)).items():
            f = fam[family(name)]
            f[key] += calls * cs[counter] * 1024.0 * (FETCH_FACTOR if key == 'fetch' else 1.0)
            if key == 'fetch':
                f['launches'] += calls
    for name, (calls, avg_us, cs) in table(os.path.join(a.indir, 'pmc_SQ_VALU_MFMA_BUSY_CYCLES.md')).items():
        f = fam[family(name)]
        f['mfma'] += calls * cs['SQ_VALU_MFMA_BUSY_CYCLES']
        f['active'] += calls * cs['GRBM_GUI_ACTIVE']
        f['mlaunches'] += calls
        f['mdur_us'] += calls * avg_us          # duration of the same dispatches IN THE SAME PASS (the counter pass runs a little slower than an un-profiled one)
    for name, (calls, _, cs) in table(os.path.join(a.indir, 'pmc_TCC_HIT_sum.md')).items():
        f = fam[family(name)]
        f['hit'] += calls * cs['TCC_HIT_sum']
        f['miss'] += calls * cs['TCC_MISS_sum']
        f['req'] += calls * cs.get('TCC_REQ_sum', 0.0)
    for name, (calls, _, cs) in table(os.path.join(a.indir, 'pmc_TCC_EA0_RDREQ_sum.md')).items():
        f = fam[family(name)]
        f['rdreq'] += calls * cs['TCC_EA0_RDREQ_sum']
        f['rlaunches'] += calls
    try:
        commit = subprocess.check_output(['git', '-C', os.path.dirname(os.path.abspath(__file__)), 'rev-parse', '--short', 'HEAD'], text=True).strip()
    except Exception:                 # noqa: BLE001  (the GPU box has no .git: filled in when the file is committed)
        commit = None
    out = dict(config=cfg, batch_per_gpu=batch, commit=commit,
               source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE / TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum / TCC_EA0_RDREQ_sum (separate passes), tools/profile_round.sh',
               fetch_factor=FETCH_FACTOR,
               fetch_note='bytes = FETCH_SIZE x 1024 x 2: on gfx950 FETCH_SIZE tallies the 128-B read requests of the L2 at 64 B for every access width (calibrated on known byte counts, profiles/r03_pmc_calibration.md); WRITE_SIZE is exact; memory-side bytes include Infinity-Cache hits', kernels={})
    for name, f in fam.items():
        n = max(f['launches'], 1)
        k = dict(hbm_bytes_per_launch=round((f['fetch'] + f['write']) / n), fetch_bytes_per_launch=round(f['fetch'] / n),
                 write_bytes_per_launch=round(f['write'] / n), launches_sampled=f['launches'])
        if f['hit'] + f['miss'] > 0:
            k['l2_hit_pct'] = round(100.0 * f['hit'] / (f['hit'] + f['miss']), 1)
            k['l2_requests_per_launch'] = round(f['req'] / n)
        if f['rlaunches'] > 0:
            k['fetch_bytes_per_launch_rdreq128'] = round(f['rdreq'] * 128.0 / f['rlaunches'])      # cross-check of the corrected FETCH_SIZE
        if f['active'] > 0:
            k['mfma_busy_pct'] = round(100.0 * f['mfma'] / (f['active'] / 8.0 * 1024.0), 1)
            if f['mdur_us'] > 0:
                # sustained shader clock of the kernel's dispatches: GRBM_GUI_ACTIVE is summed over the 8 XCD instances, each counts the cycles its
                # graphics pipe was busy = the dispatch's duration in shader clocks (MI355X_MICROARCH.md: DVFS give-back)
                ghz, avg = f['active'] / 8.0 / (f['mdur_us'] * 1e3), f['mdur_us'] / max(f['mlaunches'], 1)
                k['pmc_pass_avg_us'] = round(avg, 1)
                # (GRBM_GUI_ACTIVE also counts the dispatch's ramp-up / drain outside the kernel's own begin / end stamps: a few microseconds, i.e. the
                #  quotient is only a clock for launches that last hundreds of microseconds -- short kernels read above the 2.4 GHz the part can reach)
                if avg >= 250.0 and ghz <= 2.45:
                    k['sclk_ghz'] = round(ghz, 3)
        out['kernels'][name] = k
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in list(out['kernels'].items())[:8]}, indent=1))


if __name__ == '__main__':
    main()
