#!/usr/bin/env python3
"""Compact view of one bench.py JSON line (stdin or a file argument): headline, roofline, per-class kernel times, every extras leg."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(txt.strip().splitlines()[-1])
r = d['roofline']
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  roofline frac {r['frac']:.3f}  sclk {r.get('sclk_mhz_under_load')} MHz  devices {d.get('devices')}")
print({k: (v['ms'], v.get('tflops')) for k, v in d.get('kernels_one_step', {}).items()})
cb = d.get('cpu_baseline')
if cb:
    print(f"cpu_baseline {cb['value']:.2f} {cb['unit']} on {cb['cores']} cores ({cb['kind']}): x{d['value'] / cb['value']:.0f}")
for k, v in d.get('extras', {}).items():
    if not isinstance(v, dict):
        continue
    if 'value' in v:
        fr = (v.get('roofline') or {}).get('frac')
        print(f"extras.{k}: {v['value']:.1f} {v.get('unit', '')}  {v.get('ms_per_step', 0):.3f} ms/step" + (f"  frac {fr:.3f}" if fr else ''))
        if isinstance(v.get('mcts_cfg3'), dict) and 'value' in v['mcts_cfg3']:
            print(f"    .mcts_cfg3: {v['mcts_cfg3']['value']:.1f} decisions/s  x{v['mcts_cfg3']['speedup_vs_fp32_adjacent']:.3f} the adjacent fp32 measurement")
    elif k == 'single_episode':
        print('extras.single_episode:', {kk: round(vv, 3) for kk, vv in v.items() if isinstance(vv, float)})
