#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<tag>_* (tracked) and stamp the
commit they were taken at (the GPU box has no .git).  profiles/pmc_latest.json = the headline workload's PMC fold, which bench.py
reports as roofline.traffic / hbm_gbs / mfma_busy_pct when config and batch match.
    python tools/collect_profiles.py r02 <commit> [--latest]"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, commit, latest=False):
    src, dst = os.path.join(REPO, 'gpurun_out', tag), os.path.join(REPO, 'profiles')
    line = open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, f'{tag}_bench.json'), 'w').write(line + '\n')
    for f, t in (('kernel_stats.md', 'kernel_stats.md'), ('kernel_stats.csv', 'kernel_stats.csv'), ('pmc_FETCH_SIZE.md', 'pmc_fetch_size.md'),
                 ('pmc_WRITE_SIZE.md', 'pmc_write_size.md'), ('pmc_SQ_VALU_MFMA_BUSY_CYCLES.md', 'pmc_mfma_busy.md')):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f'{tag}_{t}'))
    p = json.load(open(os.path.join(src, 'pmc.json')))
    p['commit'] = commit
    json.dump(p, open(os.path.join(dst, f'{tag}_pmc.json'), 'w'), indent=1)
    if latest:
        json.dump(p, open(os.path.join(dst, 'pmc_latest.json'), 'w'), indent=1)
    print('collected', tag, 'at', commit)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], '--latest' in sys.argv)
