#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE tables written by tools/profile_bench.sh (two separate --pmc passes) into per-launch HBM
traffic per kernel family (all template instantiations of a kernel pooled, like bench.py's event timing pools them).

  python tools/pmc_traffic.py gpurun_out/r01b profiles/hbm_traffic.json --batch 8 --config c3

Units (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB.  WRITE_SIZE matched the known output size of
the 64->64 @512^2 layer exactly (524288 KiB for 8*64*512*512*4 B).  FETCH_SIZE is documented to under-count 16-B/lane
streaming reads by 2x on gfx950; the conv kernels stage activations with 4-B/lane loads, for which the counter matched the
expected bytes (x + 1.5x halo re-read), so no correction factor is applied -- `fetch_note` records that.
"""
import argparse
import collections
import json
import os
import re


# device function name -> the label the library's event timing (and bench.py) pools it under
ALIAS = {'conv3_mfma_kernel': 'conv_mfma_kernel'}


def table(path):
    rows = {}
    for line in open(path):
        m = re.match(r'\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.e+]+) \|', line)
        if m:
            rows[m.group(1)] = (int(m.group(2)), float(m.group(4)))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('indir')
    ap.add_argument('out')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--config', default='c3')
    a = ap.parse_args()
    fam = collections.defaultdict(lambda: dict(launches=0, fetch_bytes=0.0, write_bytes=0.0))
    for key, fn in (('fetch_bytes', 'pmc_FETCH_SIZE.md'), ('write_bytes', 'pmc_WRITE_SIZE.md')):
        for name, (calls, kib) in table(os.path.join(a.indir, fn)).items():
            base = name.split('<')[0].strip()
            f = fam[ALIAS.get(base, base)]
            f[key] += calls * kib * 1024.0
            if key == 'fetch_bytes':
                f['launches'] += calls
    out = dict(config=a.config, batch_per_gpu=a.batch, source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_bench.sh',
               fetch_note='KiB units; no 2x gfx950 correction applied (4-B/lane staging loads matched expected bytes uncorrected)', kernels={})
    for name, f in fam.items():
        n = max(f['launches'], 1)
        out['kernels'][name] = dict(hbm_bytes_per_launch=round((f['fetch_bytes'] + f['write_bytes']) / n), fetch_bytes_per_launch=round(f['fetch_bytes'] / n),
                                    write_bytes_per_launch=round(f['write_bytes'] / n), launches_sampled=f['launches'])
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main()
