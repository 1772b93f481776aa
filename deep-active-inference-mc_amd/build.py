"""Builds the HIP engine in-tree: hipcc --offload-arch=gfx950 -> libefe_mi355x.so next to this file.
gfx950 (MI355X / CDNA4) is the only target; hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['csrc/kernels.hip', 'csrc/decoder.hip', 'csrc/encoder.hip', 'csrc/mcts.hip', 'csrc/engine.hip']
HEADERS = ['csrc/kernels.h', 'csrc/philox.h', 'csrc/mfma_pipe.h', '../include/efe_engine.h']
LIB = os.path.join(HERE, 'libefe_mi355x.so')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-pass-failed',
           *[os.path.join(HERE, s) for s in SOURCES], '-o', LIB]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError('hipcc failed building libefe_mi355x.so')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
