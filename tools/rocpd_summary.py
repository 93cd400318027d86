#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the per-kernel table `--stats` prints:
name, calls, total / average / min / max duration (ns), percentage.  Writes markdown + csv next to each other.

  python tools/rocpd_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats
"""
import csv
import sqlite3
import sys


def main(db_path, out_prefix):
    db = sqlite3.connect(db_path)
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), '
                      'max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_prefix + '.csv', 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage', 'VGPR', 'AGPR', 'LDS', 'Scratch'])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 3), r[6], r[7], r[8], r[9]])
    with open(out_prefix + '.md', 'w') as f:
        f.write('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds | scratch |\n|---|---|---|---|---|---|---|---|---|---|---|\n')
        for r in rows:
            name = r[0] if len(r[0]) < 110 else r[0][:107] + '...'
            f.write(f'| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | {100.0 * r[2] / total:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} |\n')
    print(f'{len(rows)} kernels, {total / 1e6:.2f} ms total -> {out_prefix}.md/.csv')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
