"""dev tool: timeline of the kernels of ONE calculate_G call from a rocprofv3 kernel trace.
   run:      rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/calcg_timeline.py run
   analyse:  python tools/calcg_timeline.py show OUT/*/*_kernel_trace.csv"""
import sys, os, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == 'run':
    import torch, daimc_amd
    m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=1)
    E = 64
    s = torch.randn(E * 4, 10, device='cuda') * 0.3
    pi = torch.eye(4, device='cuda').repeat(E, 1)
    for _ in range(12):
        m.calculate_G(s, pi, samples=10)
    torch.cuda.synchronize()
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
    # one call = from a k_pack_x to the next k_pack_x; take the 9th
    idx = [i for i, r in enumerate(rows) if 'k_pack_x' in r['Kernel_Name']]
    a, b = idx[8], idx[9]
    t0 = int(rows[a]['Start_Timestamp'])
    prev_end = None
    tot_gap = 0
    for r in rows[a:b + 1]:
        s_, e_ = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        gap = (s_ - prev_end) if prev_end is not None else 0
        tot_gap += max(gap, 0)
        print(f"{s_ / 1e3:9.1f} us  +{(e_ - s_) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {r['Kernel_Name'].split('(')[0][-40:]}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
        prev_end = e_
    print('call period %.1f us, idle inside %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, tot_gap / 1e3))
