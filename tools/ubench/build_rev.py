#!/usr/bin/env python
"""dev tool: build the engine library as it was at a git revision, outside the product tree (same-box A/B against the working tree).

  python tools/ubench/build_rev.py <name> <git-rev>      ->  tools/ubench/alt/<name>/libefe_mi355x.so

Run a bench against it with  EFE_LIB_PATH=tools/ubench/alt/<name>/libefe_mi355x.so python bench.py ...  (the build id is "alt-<name>",
so `_lib.load()` does not compare it with the sources of the working tree)."""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = 'deep-active-inference-mc_amd'
name, rev = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp(prefix='efe_rev_')
ar = subprocess.Popen(['git', '-C', ROOT, 'archive', rev, PKG + '/csrc', PKG + '/build.py', 'include'], stdout=subprocess.PIPE)
subprocess.check_call(['tar', '-x', '-C', tmp], stdin=ar.stdout)
assert ar.wait() == 0
sys.path.insert(0, os.path.join(tmp, PKG))
import build as B
out = os.path.join(ROOT, 'tools', 'ubench', 'alt', name)
os.makedirs(out, exist_ok=True)
open(os.path.join(tmp, 'id.cpp'), 'w').write('extern "C" const char* efe_build_id(void) { return "alt-%s"; }\n' % name)
subprocess.check_call(['g++', '-O1', '-fPIC', '-c', os.path.join(tmp, 'id.cpp'), '-o', os.path.join(tmp, 'id.o')])
objs, procs = [], []
for s_ in B.SOURCES:
    src = os.path.join(tmp, PKG, s_)
    objs.append(src + '.o')
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', *[f for f in B.FLAGS if f != '-shared' and not f.startswith('-Wl,')], '-c', src, '-o', src + '.o']))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(['/opt/rocm/bin/hipcc', *B.FLAGS, os.path.join(tmp, 'id.o'), *objs, '-o', os.path.join(out, 'libefe_mi355x.so')])
shutil.rmtree(tmp)
print(os.path.join(out, 'libefe_mi355x.so'))
