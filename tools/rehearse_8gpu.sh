#!/bin/bash
# dev tool (GPU box): rehearsal of the driver's 8-GPU line on a 1-GPU lease -- `python bench.py --gpus 8 --share-device` (8 ranks launched by
# bench.py itself, every rank on cuda:0, gloo), full line with the collective extras (mcts_cfg4_sharded, Animal-AI); wall time for the
# driver's 1800 s budget.  Then the RCCL path at one rank (--force-dist).
mkdir -p gpurun_out/rehearse
t0=$(date +%s)
timeout 1700 python bench.py --gpus 8 --share-device --min-seconds 2 > gpurun_out/rehearse/gpus8_share.json 2> gpurun_out/rehearse/gpus8_share.err
echo "gpus8 rc=$? wall=$(( $(date +%s) - t0 ))s"
t0=$(date +%s)
timeout 600 python bench.py --force-dist --min-seconds 2 > gpurun_out/rehearse/force_dist.json 2> gpurun_out/rehearse/force_dist.err
echo "force-dist rc=$? wall=$(( $(date +%s) - t0 ))s"
python - <<PY
import json
for f in ('gpus8_share', 'force_dist'):
    try:
        d = json.loads(open('gpurun_out/rehearse/%s.json' % f).read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ('value', 'n_gpus', 'rccl_ranks', 'ranks', 'backend', 'per_rank_ms_per_step', 'all_gather_ms')},
              'roofline' in d, 'cpu_baseline' in d, sorted(d.get('extras', {})))
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -5 gpurun_out/rehearse/gpus8_share.err
