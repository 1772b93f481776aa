"""ORACLE tooling (build container only: needs /root/reference).  Cross-check for SURVEY 8d / BASELINE.md section 4: the CPU
baseline that bench.py times on the GPU box is the restatement oracle/efe_oracle.py (the reference's Python cannot travel);
this script times the restatement and the shimmed reference itself on the same rows, same thread count, torch RNG on both
sides, so that the proxy is known to be fair (target: within +-10 %).

Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.time_vs_reference [rows depth samples threads]"""
import os
import sys
import time
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import synth
from oracle.efe_oracle import OracleModel, TorchNoise


def main():
    rows, depth, samples, threads = (int(x) for x in (sys.argv[1:5] + ['32', '2', '4', '8'][len(sys.argv) - 1:]))
    torch.set_num_threads(threads)
    torch.set_grad_enabled(False)
    weights = synth.make_weights(1234, 1.15)
    sys.path.insert(0, '/root/reference')
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    from src.torchmodel import ActiveInferenceModel
    ref = ActiveInferenceModel(10, 4, 0.0, 1.0, 1.0, colour_channels=1, resolution=64)
    ref.model_down.qs_net[9] = nn.Linear(576, 256)
    ref.precision = torch.float32
    for part, mod in (('top', ref.model_top), ('mid', ref.model_mid), ('down', ref.model_down)):
        mod.load_state_dict({k[len(part) + 1:]: torch.from_numpy(v.copy()) for k, v in weights.items() if k.startswith(part + '.')})
    orc = OracleModel(weights, TorchNoise())
    o = torch.from_numpy(np.repeat(synth.make_frames(5, (rows + 3) // 4), 4, axis=0)[:rows])
    pi = torch.eye(4).repeat((rows + 3) // 4, 1)[:rows]

    def t_ref():
        t = time.perf_counter(); ref.calculate_G_repeated(o, pi, steps=depth, calc_mean=False, samples=samples); return time.perf_counter() - t

    def t_orc():
        t = time.perf_counter(); orc.calculate_G_repeated(o, pi, depth, False, samples, 0); return time.perf_counter() - t
    t_ref(); t_orc()
    tr = sorted(t_ref() for _ in range(3))[1]
    to = sorted(t_orc() for _ in range(3))[1]
    print(f'{rows} rows x depth {depth} x {samples} samples, {threads} threads, torch {torch.__version__}: '
          f'reference {tr:.3f} s ({rows / tr:.2f} rollouts/s), restatement {to:.3f} s ({rows / to:.2f} rollouts/s), '
          f'restatement / reference time = {to / tr:.3f}')


if __name__ == '__main__':
    main()
