#!/usr/bin/env python3
"""Compact view of one bench.py JSON line (stdin): value, ms/step, roofline fraction, per-class ms / TFLOP/s."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  roofline frac {d['roofline']['frac']:.3f}")
print({k: (v['ms'], v.get('tflops')) for k, v in d.get('kernels_one_step', {}).items()})
