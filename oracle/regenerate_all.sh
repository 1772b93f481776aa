#!/bin/bash
# ORACLE tooling -- build container only (needs /root/reference): regenerates EVERY fixture under tests/golden/ from the reference.
# Each generator is deterministic (committed seeds; torch.manual_seed for the stats fixtures), so the files come out bit-identical.
# About 16 minutes on 8 cores.      usage:  bash oracle/regenerate_all.sh
set -e
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
python -m oracle.make_golden            # networks, calculate_G*, rollouts, simulate, planners (injected Philox noise)
python -m oracle.make_golden_deep       # planners at benchmark depth, upstream-intent reward
python -m oracle.make_golden_env        # environment (Game) capture
python -m oracle.make_golden_invalid    # bare-except fallback of mcts_step_simulate
python -m oracle.make_golden_thr        # the planner's early stops at benchmark depth (thresholds 0.5 / 0.4)
python -m oracle.make_golden_defaults   # the reference planner at MCTS_Params() defaults (300 repeats) and with simulation_repeats = 2
python -m oracle.make_golden_stats      # SAMPLES of the unpatched reference under torch's own generator (statistical pin)
python -m oracle.make_golden_stats_planner   # ... and 512 whole planner decisions under that generator
python tests/golden/make_c_blob.py      # calcG_m4s1_g115.npz -> flat blob for tests/c_abi_smoke.c (needs no reference)
