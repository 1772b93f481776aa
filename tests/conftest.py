import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    d = np.load(os.path.join(GOLD, name + '.npz'))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(scope='session')
def weights_cache():
    from oracle import synth
    cache = {}

    def get(wseed, gain):
        key = (int(wseed), float(gain))
        if key not in cache:
            cache[key] = synth.make_weights(int(wseed), float(gain))
        return cache[key]
    return get


def eps_calcG(seed, M, S, stage, row_offset=0):
    """injected normals in the layout efe_calculate_g expects: [T1_0..T1_{S-1}, T2_*, D2B_*] x [M,10]"""
    from oracle import philox as PX
    out = []
    for pas in (PX.PASS_T1, PX.PASS_T2, PX.PASS_D2B):
        for i in range(S):
            out.append(PX.normals(seed, M, 10, pas, i, stage, row_offset))
    return np.stack(out, 0)


def eps_rollout(seed, M, D, S, stage0, row_offset=0):
    from oracle import philox as PX
    parts = [PX.normals(seed, M, 10, PX.PASS_ROOT, 0, stage0, row_offset).reshape(-1)]
    for t in range(D):
        parts.append(eps_calcG(seed, M, S, stage0 + t, row_offset).reshape(-1))
    return np.concatenate(parts)


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota (oversubscribing torch's intra-op
    pool on a quota-limited container stalls it)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)
