"""dev tool: ten one-episode planner decisions (lock-step planner at E = 1, S = 10, depth 5) for a rocprofv3 kernel trace:
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <out> -- python $GRAFT_REPO_ROOT/tools/single_trace.py ; python tools/trace_gaps.py <csv> 0.5"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import daimc_amd
m = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device='cuda:0', seed=7)
frame = torch.rand(1, 1, 64, 64, device='cuda:0')
q = daimc_amd.MCTS_Params(); q.repeats, q.simulation_depth, q.threshold, q.use_means, q.samples = 50, 5, 2.0, False, 10
for _ in range(10):
    daimc_amd.active_inference_mcts_batch(m, frame, q, o_shape=(1, 64, 64))
torch.cuda.synchronize()
