"""dev tool: lock-step MCTS over 64 episodes as G concurrent groups (own context, own stream, own host thread each)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, daimc_amd
E, G = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda', 0)
base = daimc_amd.ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, device=dev, seed=1)
models = [base] + [base.replica() for _ in range(G - 1)]
streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
p = daimc_amd.MCTS_Params(); p.repeats, p.simulation_depth, p.use_means, p.threshold, p.samples = 50, 5, False, 2.0, 10
frames = torch.rand(E, 1, 64, 64, device=dev)
per = E // G
def work(g, out):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[g]):
        m = models[g]
        m._stage = 0
        out[g] = daimc_amd.active_inference_mcts_batch(m, frames[g * per:(g + 1) * per], p, o_shape=(1, 64, 64), episode_offset=g * per)
        streams[g].synchronize()
def run():
    out = [None] * G
    th = [threading.Thread(target=work, args=(g, out)) for g in range(G)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return out
run()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); out = run(); ts.append(time.perf_counter() - t0)
print(f'G={G}: {min(ts) * 1e3:.1f} ms per decision batch -> {E / min(ts):.1f} decisions/s', [round(t * 1e3, 1) for t in ts])
if G > 1:       # same results as one lock-step batch
    base._stage = 0
    ref, dist = daimc_amd.active_inference_mcts_batch(base, frames, p, o_shape=(1, 64, 64))
    got = [o for g in range(G) for o in out[g][0]]
    print('paths equal:', [o[0] for o in got] == [o[0] for o in ref], ' G equal:', [o[4] for o in got] == [o[4] for o in ref])
