"""Builds the HIP engine in-tree: hipcc --offload-arch=gfx950 -> libefe_mi355x.so next to this file (and, with torch present,
the torch.ops registration library libefe_torch_ops.so).  gfx950 (MI355X / CDNA4) is the only target; hipcc cross-compiles
without a GPU.

Every library is stamped with a digest of the sources it was compiled from (`efe_build_id()`); `_lib.load()` compares the
stamp with the sources next to it, so a stale binary shipped beside newer sources fails loudly instead of passing tests."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['csrc/kernels.hip', 'csrc/decoder.hip', 'csrc/encoder.hip', 'csrc/mcts.hip', 'csrc/fused.hip', 'csrc/generic.hip', 'csrc/generic_dec.hip', 'csrc/generic_enc.hip', 'csrc/bf16x3.hip', 'csrc/engine.hip']
HEADERS = ['csrc/kernels.h', 'csrc/philox.h', 'csrc/mfma_pipe.h', '../include/efe_engine.h']
LIB = os.path.join(HERE, 'libefe_mi355x.so')
OPS_SOURCES = ['csrc/torch_ops.cpp']
OPS_LIB = os.path.join(HERE, 'libefe_torch_ops.so')
# the SONAME lets the dynamic linker recognise an engine build loaded from another path (EFE_LIB_PATH) as THE engine library when
# libefe_torch_ops.so asks for it: one copy in the process, never two
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-pass-failed', '-Wl,-soname,libefe_mi355x.so']


def source_digest(files=None):
    """sha256 over the engine sources and headers (names + contents), first 16 hex digits"""
    h = hashlib.sha256()
    for f in (files or SOURCES + HEADERS):
        path = os.path.join(HERE, f)
        if not os.path.exists(path):
            continue
        h.update(f.encode())
        with open(path, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def _stamp(lib, marker):
    """build id compiled into an existing library, read from the file's bytes (`<marker>=<16 hex digits>` string constant) --
    no dlopen: a library loaded here would stay mapped, and a rebuilt one at the same path would not be re-read"""
    if not os.path.exists(lib):
        return None
    try:
        data = open(lib, 'rb').read()
        key = (marker + '=').encode()
        i = data.find(key)
        return data[i + len(key):i + len(key) + 16].decode() if i >= 0 else None
    except Exception:
        return None


def needs_build():
    return _stamp(LIB, 'EFE_BUILD_ID') != source_digest()


def _compile_objects(hipcc, srcs, verbose):
    """one hipcc -c per source, in parallel, cached under build/ by a digest of (source, headers, flags): a kernel edit recompiles
    one file (10-30 s) instead of the whole library"""
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(HERE, 'build')
    os.makedirs(obj_dir, exist_ok=True)
    hdr = source_digest(HEADERS)

    def one(src):
        h = hashlib.sha256(open(src, 'rb').read() + hdr.encode() + ' '.join(FLAGS).encode()).hexdigest()[:16]
        obj = os.path.join(obj_dir, os.path.basename(src) + '.' + h + '.o')
        if not os.path.exists(obj):
            for old in os.listdir(obj_dir):
                if old.startswith(os.path.basename(src) + '.'):
                    os.unlink(os.path.join(obj_dir, old))
            cmd = [hipcc, *[f for f in FLAGS if f != '-shared' and not f.startswith('-Wl,')], '-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f'hipcc failed compiling {src}')
        return obj
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        return list(ex.map(one, srcs))


def build(force=False, verbose=False):
    if force or needs_build():
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
        if force:
            import shutil
            shutil.rmtree(os.path.join(HERE, 'build'), ignore_errors=True)     # --force: every object from source
        objs = _compile_objects(hipcc, srcs, verbose)
        stamp = os.path.join(HERE, 'build', 'build_id.cpp')
        with open(stamp, 'w') as fh:
            fh.write('extern "C" const char* efe_build_id(void) {\n'
                     f'    static const char stamp[] = "EFE_BUILD_ID={source_digest()}";      // the marker lets build.py read the stamp from the file without dlopen\n'
                     '    return stamp + 13;\n}\n')
        stamp_o = stamp[:-4] + '.o'
        r = subprocess.run(['g++', '-O1', '-fPIC', '-c', stamp, '-o', stamp_o], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError('g++ failed compiling the build-id stamp')
        cmd = [hipcc, *FLAGS, stamp_o, *objs, '-o', LIB]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError('hipcc failed linking libefe_mi355x.so')
    build_torch_ops(force=force, verbose=verbose)
    return LIB


def ops_digest():
    return source_digest(OPS_SOURCES + ['../include/efe_engine.h'])


def build_torch_ops(force=False, verbose=False):
    """torch.ops.efe.* registration (csrc/torch_ops.cpp): plain C++ against the torch headers, linked to libefe_mi355x.so"""
    src = os.path.join(HERE, OPS_SOURCES[0])
    if not os.path.exists(src):
        return None
    if not force and _stamp(OPS_LIB, 'EFE_OPS_BUILD_ID') == ops_digest():
        return OPS_LIB
    import torch
    from torch.utils import cpp_extension as ce
    inc = [f'-I{p}' for p in ce.include_paths()] + ['-I/opt/rocm/include', f'-I{os.path.join(HERE, "..", "include")}']
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    abi = int(getattr(torch._C, '_GLIBCXX_USE_CXX11_ABI', 1))
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', f'-D_GLIBCXX_USE_CXX11_ABI={abi}',
           f'-DEFE_OPS_BUILD_ID="{ops_digest()}"', *inc, src, '-o', OPS_LIB,
           f'-L{HERE}', '-lefe_mi355x', f'-L{tlib}', '-ltorch', '-ltorch_cpu', '-lc10', '-ltorch_hip', '-lc10_hip',
           '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,$ORIGIN', f'-Wl,-rpath,{tlib}', '-Wl,-rpath,/opt/rocm/lib']
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError('g++ failed building libefe_torch_ops.so')
    return OPS_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
