#!/usr/bin/env python
"""Per-kernel register / spill / LDS figures of the BUILT engine library, read from the AMDGPU code-object metadata inside it
(no recompilation): the gfx950 ELF images are cut out of the .hip_fatbin section and their NT_AMDGPU_METADATA note is printed by
llvm-readelf.   python tools/isa_report.py [path/to/libefe_mi355x.so]"""
import os, re, struct, subprocess, sys, tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, 'deep-active-inference-mc_amd', 'libefe_mi355x.so')
FIELDS = ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size',
          '.group_segment_fixed_size', '.max_flat_workgroup_size')


def code_objects(lib):
    """the EM_AMDGPU ELF images embedded in the library"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', lib, os.path.join(tmp, 'discard')])
        data = open(fat, 'rb').read()
    out = []
    for m in re.finditer(b'\x7fELF', data):
        o = m.start()
        hdr = data[o:o + 64]
        if len(hdr) < 64 or hdr[4] != 2 or struct.unpack_from('<H', hdr, 0x12)[0] != 224:       # ELF64, EM_AMDGPU
            continue
        e_shoff = struct.unpack_from('<Q', hdr, 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from('<HH', hdr, 0x3A)
        out.append(data[o:o + e_shoff + e_shentsize * e_shnum])
    return out


def demangle(names):
    try:
        p = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
        return p.stdout.splitlines() if p.returncode == 0 and p.stdout else names
    except OSError:
        return names


def kernels(lib=DEFAULT_LIB):
    """{demangled kernel name (arguments stripped): {field: int}}"""
    res = {}
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix='.elf') as f:
            f.write(img); f.flush()
            txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], capture_output=True, text=True).stdout
        for blk in re.split(r'\n  - \.agpr_count:', txt)[1:]:
            blk = '.agpr_count:' + blk
            name = re.search(r'\n\s+\.name:\s+(\S+)', blk)
            if not name:
                continue
            vals = {}
            for k in FIELDS:
                mm = re.search(r'(?:^|\n)\s*' + re.escape(k) + r':\s+(\d+)', blk)
                if mm:
                    vals[k] = int(mm.group(1))
            res[name.group(1)] = vals
    names = list(res)
    pretty = [re.sub(r'\(.*$', '', d).replace('efe::', '').replace('void ', '') for d in demangle(names)]
    return {p: res[n] for p, n in zip(pretty, names)}


if __name__ == '__main__':
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB)
    print(f'{"kernel":44s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"vspill":>6s} {"sspill":>6s} {"scratch":>7s} {"lds":>7s} {"wg":>5s}')
    for k in sorted(ks):
        v = ks[k]
        print(f'{k[:44]:44s} ' + ' '.join(f'{v.get(f, -1):>{w}d}' for f, w in zip(FIELDS, (5, 5, 5, 6, 6, 7, 7, 5))))
