"""dev tool: GPU busy / idle analysis of a rocprofv3 --kernel-trace csv: union of kernel intervals over all streams inside the window
of the last N dispatches of a marker kernel.   usage: trace_gaps.py <kernel_trace.csv> [window_start_fraction]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
w0 = t0 + int((t1 - t0) * frac)
ev = [e for e in ev if e[0] >= w0]
span = max(e[1] for e in ev) - ev[0][0]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in ev:
    if cur_e is None: cur_s, cur_e = s, e; continue
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split('(')[0].replace('void efe::', '')[:50]
    tot[k][0] += e - s; tot[k][1] += 1
print(f'window {span / 1e6:.2f} ms, {len(ev)} dispatches, GPU busy (union) {busy / 1e6:.2f} ms = {busy / span:.3f}, idle gaps {len(gaps)}: total {(span - busy) / 1e6:.2f} ms, '
      f'mean {((span - busy) / max(len(gaps), 1)) / 1e3:.1f} us')
print('sum of kernel durations %.2f ms (overlap factor %.3f)' % (sum(v[0] for v in tot.values()) / 1e6, sum(v[0] for v in tot.values()) / busy))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f'  {k:52s} {v[1]:6d} x {v[0] / v[1] / 1e3:9.1f} us = {v[0] / 1e6:8.2f} ms')
big = sorted(gaps, reverse=True)[:8]
print('largest gaps (us, next kernel):', [(round(g / 1e3, 1), n.split('(')[0][-30:]) for g, n in big])
