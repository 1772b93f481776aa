#!/bin/bash
# dev tool: sample sclk / power with rocm-smi while the default rollout bench runs (is the step clock- or power-limited?)
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu --no-extras --no-prof --steps 400 --warmup 20 --single-region > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*sclk clock level: [0-9]+: \(([0-9]+Mhz)\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/power \1 W/' | tr '\n' ' '; echo
  sleep 0.2
done | awk '$4 > 600' | tail -25
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/clock_probe_bench.json')); print('ms_per_step', d['ms_per_step'], 'rollouts/s', d['value'])"
