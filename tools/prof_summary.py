#!/usr/bin/env python
"""Summarise a tools/gpu_profile.sh output directory (rocprofv3 csv) into a small text table:
per kernel: calls, avg us, share; the STEADY-STATE duration (the MEDIAN of each kernel's calls in the kernel trace: the first two launches
after every idle gap -- start of the run, the barrier between warm-up and timed steps -- run inside the clock ramp) with the roofline
fraction recomputed from it (algorithmic FLOPs of DESIGN section 5 / median), next to MFMA-busy and the shader clock of the PMC pass; PMC counters averaged per dispatch.

  python tools/prof_summary.py <dir> [rows]        rows = decoder images per launch of the persistent kernels (default: the largest
                                                   one-workgroup-per-image grid in the trace, i.e. 19 200 for the headline)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace('void efe::', '').replace('efe::', '')
    return n.split('(')[0][:60]


PEAK_TF = 157.3          # fp32 MFMA, MI355X_MICROARCH.md
# algorithmic MACs per decoder image / row: dSprites kernels, and the generic path at BASELINE configs[4] (3 x 84 x 84)
MACS = {'k_dec_b4<1>': 20054016, 'k_dec_b4<4>': 20054016, 'k_dec_a': 18874368, 'k_dec_a_s': 18874368, 'k_fc4<2>': 4194304, 'k_fc4<1>': 4194304}
MACS_G84 = {'k_dec_bg<3>': 21 * 21 * 4 * 9 * 64 * 32 + 84 * 84 * 9 * 32 * 3, 'k_convt_p<1, 4>': 21 * 21 * 9 * 64 * 64, 'k_convt_p<2, 4>': 21 * 21 * 9 * 64 * 64,
            'k_convt_12<4>': 2 * 21 * 21 * 9 * 64 * 64,
            'k_fc4<2>': 256 * 64 * 21 * 21}


# the opt-in split-operand kernels (csrc/bf16x3.hip): fp32-equivalent work against the dense 16-bit MFMA peak / products per MAC
PEAK_16 = 2516.6
SPLIT = {'SchB3': 6, 'SchH2': 3}
MACS_SPLIT = {'k_fc4_b3': 4194304, 'k_dec_a_b3': 18874368, 'k_dec_b_b3': 20054016}


def split_peak(k):
    """(macs per image, peak TFLOP/s) of a split-operand kernel instance, or None"""
    base = k.split('<')[0]
    if base not in MACS_SPLIT:
        return None
    for sc, nprod in SPLIT.items():
        if sc in k:
            return MACS_SPLIT[base], PEAK_16 / nprod
    return None


def steady_table(d, rows_arg):
    """-> lines; per kernel the MEDIAN duration of its calls in the kernel trace and the roofline fraction it implies"""
    tr = glob.glob(os.path.join(d, 'kt', '**', '*kernel_trace.csv'), recursive=True)
    if not tr:
        return ['(no kernel trace csv: steady-state table skipped)'], {}
    calls = defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        k = short(r['Kernel_Name'])
        wg = max(int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 256)) or 256), 1)
        calls[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) // wg))
    generic = any(k.startswith('k_dec_bg') for k in calls)
    macs = dict(MACS, **MACS_G84) if generic else MACS
    per_img = [k for k in calls if k.startswith('k_dec_b4<1>') or k.startswith('k_dec_bg')]
    rows = rows_arg or max((max(c[2] for c in calls[k]) for k in per_img), default=0) or 19200       # (a split-mode profile has no one-workgroup-per-image kernel: the headline's 19 200)
    out = ['\n== steady state (kernel trace, median over each kernel\'s calls) and the roofline fraction it implies ==',
           f'(persistent kernels: {rows} images per launch; peak {PEAK_TF} TFLOP/s fp32 MFMA; algorithmic MACs per image from DESIGN section 5)',
           f'{"kernel":28s} {"calls":>6s} {"all_avg_us":>11s} {"median_us":>14s} {"min_us":>14s} {"images":>8s} {"TFLOP/s":>9s} {"frac":>7s}']
    fr = {}
    for k in sorted(calls, key=lambda k: -sum(c[1] for c in calls[k])):
        if not k.startswith('k_'):
            continue
        c = sorted(calls[k])
        big = max(x[2] for x in c)
        c = [x for x in c if x[2] == big] if k in per_img else c          # the full-size launches only
        durs = sorted(x[1] for x in c)
        avg_all = sum(durs) / len(durs) / 1e3
        avg = (durs[(len(durs) - 1) // 2] + durs[len(durs) // 2]) / 2e3
        mn = durs[0] / 1e3
        n_img = big if k in per_img else rows
        sp = split_peak(k)
        if sp and n_img:
            tf = 2.0 * sp[0] * n_img / (avg * 1e-6) / 1e12
            fr[k] = tf / sp[1]
            out.append(f'{k:28s} {len(c):6d} {avg_all:11.1f} {avg:14.1f} {mn:14.1f} {n_img:8d} {tf:9.1f} {tf / sp[1]:7.3f}   (fp32-equivalent; of {sp[1]:.1f} = 16-bit peak / products)')
        elif k in macs and n_img:
            tf = 2.0 * macs[k] * n_img / (avg * 1e-6) / 1e12
            fr[k] = tf / PEAK_TF
            out.append(f'{k:28s} {len(c):6d} {avg_all:11.1f} {avg:14.1f} {mn:14.1f} {n_img:8d} {tf:9.1f} {tf / PEAK_TF:7.3f}')
        else:
            out.append(f'{k:28s} {len(c):6d} {avg_all:11.1f} {avg:14.1f} {mn:14.1f}')
    return out, fr


def main(d, rows_arg=0):
    out = []
    ks = glob.glob(os.path.join(d, 'kt', '**', '*kernel_stats.csv'), recursive=True)
    if ks:
        out.append('== rocprofv3 --kernel-trace --stats (tools/gpu_profile.sh: bench.py --steps 8 --warmup 8; avg_us includes the clock ramp of the first calls -- see the steady-state table) ==')
        out.append(f'{"kernel":62s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"pct":>6s}')
        for r in csv.DictReader(open(ks[0])):
            out.append(f'{short(r["Name"]):62s} {r["Calls"]:>6s} {float(r["TotalDurationNs"]) / 1e3:12.1f} '
                       f'{float(r["AverageNs"]) / 1e3:10.1f} {float(r["MinNs"]) / 1e3:10.1f} {float(r["Percentage"]):6.2f}')
    st_lines, frac_steady = steady_table(d, rows_arg)
    out += st_lines
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
        out.append('(MFMA_busy and clk_GHz are measured INSIDE the PMC pass -- GRBM_GUI_ACTIVE over the dispatch durations of that pass; frac_steady is the')
        out.append(' kernel-trace pass above, un-profiled clock: bench.py samples that one as roofline.sclk_mhz_under_load)')
        out.append(f'{"kernel":28s} {"time_ms":>9s} {"MFMA_busy":>10s} {"clk_GHz":>8s} {"frac_steady":>12s} {"HBM_read_GB":>12s} {"HBM_write_GB":>13s} {"HBM_GB/s":>9s}')
        for k in sorted(kt, key=lambda k: -kt[k][1]):
            if not k.startswith('k_') or k not in sq:
                continue
            t_ms = kt[k][1] / 1e6
            g = sq[k]['GRBM_GUI_ACTIVE'] / 8.0
            busy = sq[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * g) if g else 0.0
            clk = g / sq[k]['_dur'] if sq[k]['_dur'] else 0.0
            rd = 2 * mem[k]['FETCH_SIZE'] * 1024 / 1e9
            wr = mem[k]['WRITE_SIZE'] * 1024 / 1e9
            fs = f'{frac_steady[k]:12.3f}' if k in frac_steady else f'{"":12s}'
            out.append(f'{k:28s} {t_ms:9.3f} {busy:10.3f} {clk:8.3f} {fs} {rd:12.3f} {wr:13.3f} {(rd + wr) / (t_ms * 1e-3):9.0f}')
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
                if r['Counter_Name'] != key or not k.startswith('k_dec') or k.startswith('k_dec_b_b3'):        # (k_dec_b_b3 is persistent: its grid is not its image count)
                    continue
                k = 'k_dec_b' if k.startswith('k_dec_b') else k.split('<')[0]      # k_dec_b4 / k_dec_b<SR,RW>: one workgroup = one image
                e = tr.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'images_f': 0, 'images_w': 0})
                e[key] += float(r['Counter_Value'])
                e['images_f' if key == 'FETCH_SIZE' else 'images_w'] += int(r['Grid_Size']) // int(r['Workgroup_Size'])
        res = {}
        # the persistent split-operand kernels (csrc/bf16x3.hip): bytes per LAUNCH (their grid is one workgroup per CU, not their image count)
        pl = {}
        for sub, key in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
            cs = glob.glob(os.path.join(d, sub, '**', '*counter_collection.csv'), recursive=True)
            for r in csv.DictReader(open(cs[0])):
                k = short(r['Kernel_Name'])
                if r['Counter_Name'] != key or split_peak(k) is None:
                    continue
                e = pl.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'n_f': 0, 'n_w': 0})
                e[key] += float(r['Counter_Value'])
                e['n_f' if key == 'FETCH_SIZE' else 'n_w'] += 1
        for k, e in pl.items():
            res[k] = {'hbm_read_bytes_per_launch': 2 * e['FETCH_SIZE'] * 1024 / max(e['n_f'], 1), 'hbm_write_bytes_per_launch': e['WRITE_SIZE'] * 1024 / max(e['n_w'], 1),
                      'note': 'persistent kernel: per dispatch of the profiled run (19 200 images); FETCH doubled per MI355X_MICROARCH.md'}
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
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
