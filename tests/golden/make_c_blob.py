"""tests/golden/calcG_m4s1_g115.bin <- tests/golden/calcG_m4s1_g115.npz: the reference-captured calculate_G fixture (oracle/make_golden.py)
re-packed as a flat little-endian blob for the plain-C consumer of the ABI (tests/c_abi_smoke.c):

  int32 M, S, stage, 0;  uint64 noise seed;  float32 s0[M][10], pi0[M][4], eps[3S][M][10] (the injected normals: oracle/philox.py);
  float32 G[M], term0[M], term1[M], term2[M], term2_1[M], term2_2[M]        <- the REFERENCE's outputs

Usage: python tests/golden/make_c_blob.py      (needs nothing but the committed .npz; no reference)"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import philox as PX


def main(name='calcG_m4s1_g115'):
    g = np.load(os.path.join(HERE, name + '.npz'))
    M, S, stage, seed = len(g['s0']), int(g['samples']), int(g['stage']), int(g['nseed'])
    eps = np.stack([PX.normals(seed, M, 10, pas, i, stage) for pas in (PX.PASS_T1, PX.PASS_T2, PX.PASS_D2B) for i in range(S)], 0)
    with open(os.path.join(HERE, name + '.bin'), 'wb') as f:
        f.write(struct.pack('<4iQ', M, S, stage, 0, seed))
        for a in (g['s0'], g['pi0'], eps, g['G'], g['t0'], g['t1'], g['t2'], g['t2_1'], g['t2_2']):
            f.write(np.ascontiguousarray(a, dtype='<f4').tobytes())


if __name__ == '__main__':
    main()
