#!/usr/bin/env python3
"""Per-(kernel, grid) durations of a rocprofv3 rocpd trace: the layers of one kernel differ by grid, and the per-name average hides
which of them is slow.   python tools/rocpd_by_grid.py trace.db [name-substring]"""
import sqlite3
import sys


def main(db_path, pat=''):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    grid = [c for c in cols if c.startswith('grid')]
    q = f"select name, {', '.join(grid)}, count(*), avg(duration), min(duration) from kernels where name like ? group by name, {', '.join(grid)} order by name, avg(duration) desc"
    print('| kernel | grid | calls | avg us | min us |\n|---|---|---|---|---|')
    for r in db.execute(q, (f'%{pat}%',)):
        g = 'x'.join(str(v) for v in r[1:1 + len(grid)])
        print(f'| `{r[0][:60]}` | {g} | {r[-3]} | {r[-2] / 1e3:.1f} | {r[-1] / 1e3:.1f} |')


if __name__ == '__main__':
    main(*sys.argv[1:3])
