#!/usr/bin/env python3
"""dev tool: kernel timeline of ONE one-episode planner iteration from a rocprofv3 kernel trace of tools/single_lat.py:

  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/single_kt -- python $GRAFT_REPO_ROOT/tools/single_lat.py
  python tools/single_trace.py gpurun_out/single_kt

prints, for a steady-state iteration (between two k_mcts_step launches late in the run), every kernel with its stream, start offset,
duration and the gap to the previous kernel of the same stream."""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void efe::', '').replace('efe::', '').split('(')[0]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
steps = [i for i, r in enumerate(rows) if r[2].startswith('k_mcts_step')]
i0, i1 = steps[-20], steps[-19]
t0 = rows[i0][0]
last = {}
print(f'iteration = {1e-3 * (rows[i1][0] - t0):.1f} us between two k_mcts_step launches; {i1 - i0} kernels')
for s, e, n, q in rows[i0:i1 + 1]:
    gap = 1e-3 * (s - last[q]) if q in last else 0.0
    print(f'{1e-3 * (s - t0):8.1f} us  +{1e-3 * (e - s):6.1f} us  gap {gap:6.1f}  stream {q:>4s}  {n[:60]}')
    last[q] = e
