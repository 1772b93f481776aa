#!/usr/bin/env python
"""Summarise a tools/gpu_profile.sh output directory (rocprofv3 csv) into a small text table:
per kernel: calls, avg us, share; PMC counters averaged per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace('void efe::', '').replace('efe::', '')
    return n.split('(')[0][:60]


def main(d):
    out = []
    ks = glob.glob(os.path.join(d, 'kt', '**', '*kernel_stats.csv'), recursive=True)
    if ks:
        out.append('== rocprofv3 --kernel-trace --stats (bench.py --steps 2 --warmup 1) ==')
        out.append(f'{"kernel":62s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"pct":>6s}')
        for r in csv.DictReader(open(ks[0])):
            out.append(f'{short(r["Name"]):62s} {r["Calls"]:>6s} {float(r["TotalDurationNs"]) / 1e3:12.1f} '
                       f'{float(r["AverageNs"]) / 1e3:10.1f} {float(r["Percentage"]):6.2f}')
    for sub in ('pmc_sq', 'pmc_mem', 'pmc_fetch', 'pmc_write'):
        cs = glob.glob(os.path.join(d, sub, '**', '*counter_collection.csv'), recursive=True)
        if not cs:
            continue
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(cs[0])):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k][r['Counter_Name']] += 1
        names = sorted({c for k in acc for c in acc[k]})
        out.append(f'\n== PMC pass {sub}: mean per dispatch ==')
        out.append(f'{"kernel":62s} ' + ' '.join(f'{n[-22:]:>22s}' for n in names))
        for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
            if not k.startswith('k_'):
                continue
            out.append(f'{k:62s} ' + ' '.join(f'{acc[k][n] / max(cnt[k][n], 1):22.4g}' for n in names))
    print('\n'.join(out))


if __name__ == '__main__':
    main(sys.argv[1])
