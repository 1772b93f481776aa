"""dev tool (CPU, this container or the GPU box): POWER of tests/test_noise_statistics.py -- the oracle run with a deliberately broken
noise keying (two draws the reference makes independently share their Philox counters) against the unpatched-reference samples.

  python tools/noise_power.py <replicas> <kind> [<kind> ...]      kinds: none | d2b=d2a | t2=t1 | sample0 | fc4only_d2b=d2a

Round-5 results (profiles/README.md): at 4096 replicas `d2b=d2a` (loop-2 decoders share their masks) and `t2=t1` (loop-2 transition
masks = loop-1's) are detected (variance of G 5 standard errors low; corr(term2_1, term2_2) 0.02 -> 0.11), `sample0` (every MC sample
the same masks) already at 384; sharing ONLY the 16 384-feature mask between the two loop-2 decoders moves no statistic of G measurably
(corr 0.01): that class of collision is excluded by test_no_two_logical_draws_share_a_philox_counter, not by statistics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import efe_oracle as EO, philox as PX, synth
import test_noise_statistics as T
from conftest import load_golden


class Broken(EO.PhiloxNoise):
    def __init__(self, seed, kind):
        super().__init__(seed)
        self.kind = kind

    def mask(self, tag, rows, nf, pas, sample, stage, row_offset=None, fc4_perm=False):
        k = self.kind
        if k == 'd2b=d2a' and pas == PX.PASS_D2B: pas = PX.PASS_D2A
        if k == 't2=t1' and pas == PX.PASS_T2: pas = PX.PASS_T1
        if k == 'sample0': sample = 0
        if k == 'fc4only_d2b=d2a' and pas == PX.PASS_D2B and nf == 16384: pas = PX.PASS_D2A
        return super().mask(tag, rows, nf, pas, sample, stage, row_offset, fc4_perm)


def main():
    ref = load_golden('stats_calcG')
    w = synth.make_weights(int(ref['wseed']), float(ref['gain']))
    n = int(sys.argv[1])
    for kind in sys.argv[2:]:
        got = T._oracle_calcG_samples(EO.OracleModel(w, Broken(11, kind)), ref['s0'], int(ref['samples']), n, 3)
        for a in range(4):
            r0 = np.corrcoef(ref['t2_1'][:, a], ref['t2_2'][:, a])[0, 1]; r1 = np.corrcoef(got['t2_1'][:, a], got['t2_2'][:, a])[0, 1]
            print(f'{kind} row {a}: corr(t2_1, t2_2) reference {r0:.4f} broken {r1:.4f}; var G reference {ref["G"][:, a].var():.1f} broken {got["G"][:, a].var():.1f}')
        try:
            T.compare_calcG(ref, got, kind, [])
            print(kind, 'NOT detected')
        except AssertionError as e:
            print(kind, 'detected:', e)


if __name__ == '__main__':
    main()
