#!/usr/bin/env python3
"""Per-kernel PMC summary from a rocprofv3 (rocpd sqlite) --pmc run: sums each counter over its per-XCD/SE samples per
dispatch, then averages per kernel name.  python tools/rocpd_pmc.py <db> [out.md]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', n)[:70]


def main(path, out=None):
    db = sqlite3.connect(path)
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # (name, dispatch) -> counter -> sum
    dur = {}
    for name, disp, cname, val, d in db.execute('select name, dispatch_id, counter_name, counter_value, duration from pmc_events'):
        per[(name, disp)][cname] += val
        dur[(name, disp)] = d
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (name, disp), cs in per.items():
        for c, v in cs.items():
            agg[name][c].append(v)
        agg[name]['duration_us'].append(dur[(name, disp)] / 1e3)
    counters = sorted({c for v in agg.values() for c in v if c != 'duration_us'})
    lines = ['| kernel | calls | avg us | ' + ' | '.join(counters) + ' |', '|---|---|---|' + '---|' * len(counters)]
    order = sorted(agg, key=lambda n: -sum(agg[n]['duration_us']))
    for n in order[:16]:
        a = agg[n]
        k = len(a['duration_us'])
        lines.append(f'| `{short(n)}` | {k} | {sum(a["duration_us"]) / k:.1f} | ' + ' | '.join(f'{sum(a[c]) / max(len(a[c]), 1):.4g}' for c in counters) + ' |')
    text = '\n'.join(lines)
    print(text)
    if out:
        open(out, 'w').write(text + '\n')


if __name__ == '__main__':
    main(*sys.argv[1:])
