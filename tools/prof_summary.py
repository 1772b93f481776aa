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
        out.append('== rocprofv3 --kernel-trace --stats (the default bench command: bench.py --steps 5 --warmup 2; the first call of a kernel is cold) ==')
        out.append(f'{"kernel":62s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"pct":>6s}')
        for r in csv.DictReader(open(ks[0])):
            out.append(f'{short(r["Name"]):62s} {r["Calls"]:>6s} {float(r["TotalDurationNs"]) / 1e3:12.1f} '
                       f'{float(r["AverageNs"]) / 1e3:10.1f} {float(r["MinNs"]) / 1e3:10.1f} {float(r["Percentage"]):6.2f}')
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
    # per-kernel roofline table: MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)),
    # effective shader clock, HBM GB/s from FETCH_SIZE (x2 on gfx950) + WRITE_SIZE over the un-profiled kernel time
    try:
        kt = {}
        for r in csv.DictReader(open(ks[0])):
            kt[short(r['Name'])] = (int(r['Calls']), float(r['TotalDurationNs']))
        sq = defaultdict(lambda: defaultdict(float))
        cs = glob.glob(os.path.join(d, 'pmc_sq', '**', '*counter_collection.csv'), recursive=True)
        for r in csv.DictReader(open(cs[0])):
            k = short(r['Kernel_Name'])
            sq[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                sq[k]['_dur'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        mem = defaultdict(lambda: defaultdict(float))
        for sub, key in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
            cs = glob.glob(os.path.join(d, sub, '**', '*counter_collection.csv'), recursive=True)
            for r in csv.DictReader(open(cs[0])):
                if r['Counter_Name'] == key:
                    mem[short(r['Kernel_Name'])][key] += float(r['Counter_Value'])
        out.append('\n== per-kernel roofline view (whole profiled run) ==')
        out.append(f'{"kernel":28s} {"time_ms":>9s} {"MFMA_busy":>10s} {"clk_GHz":>8s} {"HBM_read_GB":>12s} {"HBM_write_GB":>13s} {"HBM_GB/s":>9s}')
        for k in sorted(kt, key=lambda k: -kt[k][1]):
            if not k.startswith('k_') or k not in sq:
                continue
            t_ms = kt[k][1] / 1e6
            g = sq[k]['GRBM_GUI_ACTIVE'] / 8.0
            busy = sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * g) if g else 0.0
            clk = g / sq[k]['_dur'] if sq[k]['_dur'] else 0.0
            rd = 2 * mem[k]['FETCH_SIZE'] * 1024 / 1e9
            wr = mem[k]['WRITE_SIZE'] * 1024 / 1e9
            out.append(f'{k:28s} {t_ms:9.3f} {busy:10.3f} {clk:8.3f} {rd:12.3f} {wr:13.3f} {(rd + wr) / (t_ms * 1e-3):9.0f}')
    except Exception as ex:
        out.append(f'(no roofline table: {ex})')
    # HBM traffic of the dominant kernel per decoder image: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts
    # 128-B requests at 64 B (MI355X_MICROARCH.md "HBM"), so it is doubled.  One k_dec_b workgroup = one image.
    try:
        import json
        tr = {}
        for sub, key in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
            cs = glob.glob(os.path.join(d, sub, '**', '*counter_collection.csv'), recursive=True)
            for r in csv.DictReader(open(cs[0])):
                k = short(r['Kernel_Name'])
                if r['Counter_Name'] != key or not k.startswith('k_dec'):
                    continue
                k = 'k_dec_b' if k.startswith('k_dec_b') else k.split('<')[0]      # k_dec_b4 / k_dec_b<SR,RW>: one workgroup = one image
                e = tr.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'images_f': 0, 'images_w': 0})
                e[key] += float(r['Counter_Value'])
                e['images_f' if key == 'FETCH_SIZE' else 'images_w'] += int(r['Grid_Size']) // int(r['Workgroup_Size'])
        res = {}
        for k, e in tr.items():
            # k_dec_a is persistent (grid 512): images per dispatch are not visible from the grid; report k_dec_b only
            if k != 'k_dec_b':
                continue
            res[k] = {'hbm_read_bytes_per_image': 2 * e['FETCH_SIZE'] * 1024 / max(e['images_f'], 1),
                      'hbm_write_bytes_per_image': e['WRITE_SIZE'] * 1024 / max(e['images_w'], 1),
                      'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH doubled per MI355X_MICROARCH.md'}
        out.append('\n== HBM traffic (JSON) ==')
        out.append(json.dumps(res))
    except Exception as ex:      # summary stays usable without the PMC passes
        out.append(f'(no traffic summary: {ex})')
    print('\n'.join(out))


if __name__ == '__main__':
    main(sys.argv[1])
