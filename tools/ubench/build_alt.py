#!/usr/bin/env python
"""dev tool: build an EXPERIMENT copy of the engine library from patched sources, outside the product tree.

  python tools/ubench/build_alt.py <name> <patch.py>

<patch.py> defines PATCH = {'generic_dec.hip': [(old_text, new_text), ...], ...}; the sources under deep-active-inference-mc_amd/csrc are
copied to a scratch directory, patched, and compiled to tools/ubench/alt/<name>/libefe_mi355x.so (git-ignored, travels with gpurun).
Run a bench against it with  EFE_LIB_PATH=tools/ubench/alt/<name>/libefe_mi355x.so python bench.py ...  -- timing experiments that
remove a piece of a kernel give WRONG RESULTS on purpose; nothing built here is part of the product."""
import os, shutil, subprocess, sys, tempfile, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'deep-active-inference-mc_amd')
name, patch = sys.argv[1], runpy.run_path(sys.argv[2])['PATCH']
tmp = tempfile.mkdtemp(prefix='efe_alt_')
shutil.copytree(os.path.join(PKG, 'csrc'), os.path.join(tmp, 'p', 'csrc'))
shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(tmp, 'include'))
for f, reps in patch.items():
    p = os.path.join(tmp, 'p', 'csrc', f)
    s = open(p).read()
    for old, new in reps:
        assert old in s, (f, old[:60])
        s = s.replace(old, new)
    open(p, 'w').write(s)
sys.path.insert(0, PKG)
import build as B
out = os.path.join(ROOT, 'tools', 'ubench', 'alt', name)
os.makedirs(out, exist_ok=True)
srcs = [os.path.join(tmp, 'p', s_) for s_ in B.SOURCES]
open(os.path.join(tmp, 'id.cpp'), 'w').write('extern "C" const char* efe_build_id(void) { return "alt-%s"; }\n' % name)
subprocess.check_call(['g++', '-O1', '-fPIC', '-c', os.path.join(tmp, 'id.cpp'), '-o', os.path.join(tmp, 'id.o')])
objs = []
procs = []
for s_ in srcs:
    o = s_ + '.o'; objs.append(o)
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', *[f for f in B.FLAGS if f != '-shared' and not f.startswith('-Wl,')], '-c', s_, '-o', o]))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(['/opt/rocm/bin/hipcc', *B.FLAGS, os.path.join(tmp, 'id.o'), *objs, '-o', os.path.join(out, 'libefe_mi355x.so')])
shutil.rmtree(tmp)
print(os.path.join(out, 'libefe_mi355x.so'))
