#!/usr/bin/env python3
"""Fold the three PMC tables written by tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE;
separate rocprofv3 --pmc passes) into per-launch figures per kernel family (all template instantiations of a kernel pooled, the
way bench.py's event timing pools them): HBM bytes per launch and matrix-pipe busy %.

  python tools/pmc_fold.py gpurun_out/r02 gpurun_out/r02/pmc.json --bench gpurun_out/r02/bench.json
  (review, then copy to profiles/pmc_latest.json with the commit the passes were taken at)

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB.  WRITE_SIZE matched the known output size
of the 64->64 @512^2 layer exactly.  FETCH_SIZE is documented to under-count 16-B/lane streaming reads by 2x on gfx950; the conv
kernels stage activations with 4-B/lane loads, for which the counter matched the expected bytes (x + halo re-read), so no
correction factor is applied to them -- `fetch_note` records that.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD
instances x 256 CUs x 4 SIMDs), the gfx94x MfmaUtil formula (ROCm 7.2 ships no gfx950 derived-counter section).
"""
import argparse
import collections
import json
import os
import re
import subprocess


# device function name -> the label the library's event timing (and bench.py) pools it under
ALIAS = {'conv3_mfma_kernel': 'conv_mfma_kernel', 'conv3_wino_kernel': 'conv_wino_kernel', 'conv3_bf16_kernel': 'conv_bf16_kernel', 'triplane_walk_kernel': 'triplane_field_kernel', 'conv3s_mfma_kernel': 'conv_mfma_kernel', 'upconv3s_mfma_kernel': 'upconv_mfma_kernel'}


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
    fam = collections.defaultdict(lambda: dict(launches=0, fetch=0.0, write=0.0, mfma=0.0, active=0.0, mlaunches=0))
    for key, fn, counter in (('fetch', 'pmc_FETCH_SIZE.md', 'FETCH_SIZE'), ('write', 'pmc_WRITE_SIZE.md', 'WRITE_SIZE')):
        for name, (calls, _, cs) in table(os.path.join(a.indir, fn)).items():
            base = name.split('<')[0].strip()
            f = fam[ALIAS.get(base, base)]
            f[key] += calls * cs[counter] * 1024.0
            if key == 'fetch':
                f['launches'] += calls
    for name, (calls, _, cs) in table(os.path.join(a.indir, 'pmc_SQ_VALU_MFMA_BUSY_CYCLES.md')).items():
        base = name.split('<')[0].strip()
        f = fam[ALIAS.get(base, base)]
        f['mfma'] += calls * cs['SQ_VALU_MFMA_BUSY_CYCLES']
        f['active'] += calls * cs['GRBM_GUI_ACTIVE']
        f['mlaunches'] += calls
    try:
        commit = subprocess.check_output(['git', '-C', os.path.dirname(os.path.abspath(__file__)), 'rev-parse', '--short', 'HEAD'], text=True).strip()
    except Exception:                 # noqa: BLE001  (the GPU box has no .git: filled in when the file is committed)
        commit = None
    out = dict(config=cfg, batch_per_gpu=batch, commit=commit,
               source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes), tools/profile_round.sh',
               fetch_note='KiB units; no 2x gfx950 correction applied (4-B/lane staging loads matched expected bytes uncorrected)', kernels={})
    for name, f in fam.items():
        n = max(f['launches'], 1)
        k = dict(hbm_bytes_per_launch=round((f['fetch'] + f['write']) / n), fetch_bytes_per_launch=round(f['fetch'] / n),
                 write_bytes_per_launch=round(f['write'] / n), launches_sampled=f['launches'])
        if f['active'] > 0:
            k['mfma_busy_pct'] = round(100.0 * f['mfma'] / (f['active'] / 8.0 * 1024.0), 1)
        out['kernels'][name] = k
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in list(out['kernels'].items())[:8]}, indent=1))


if __name__ == '__main__':
    main()
