#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in the built objects (subgnn_amd/lib/*.o), read from the
code objects' metadata notes -- no GPU needed.

    python tools/kernel_resources.py [--spills] [file.o ...]

Each .o carries its gfx950 code object in the .hip_fatbin section as a clang offload bundle: the section is
dumped (llvm-objcopy), unbundled (clang-offload-bundler) and its notes read (llvm-readelf --notes).
tests/test_kernel_resources.py uses ``kernels()`` to fail the CPU suite when a hot kernel spills.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(HERE, '..', 'subgnn_amd', 'lib')
FIELDS = ('.name', '.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.sgpr_spill_count',
          '.private_segment_fixed_size', '.group_segment_fixed_size', '.max_flat_workgroup_size')


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True)
        return out.stdout.split('\n')[:len(names)]
    except Exception:
        return list(names)


def kernels(obj):
    """-> list of dicts (one per kernel of ``obj``'s gfx950 code object): name, demangled, vgpr_count, ..."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat'), os.path.join(d, 'co')
        r = subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj], capture_output=True, text=True)
        if r.returncode != 0:
            if 'not found' in r.stderr:              # host-only object (lib.o): no kernels
                return []
            raise RuntimeError(r.stderr)
        subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in notes.split('\n'):
        m = re.match(r'\s*(- )?(\.[a-z_]+):\s+(.*)$', line)
        if not m:
            continue
        dash, key, val = m.groups()
        if key in FIELDS:
            # a kernel's map starts at the list dash in front of its first key; '.name' also occurs in the args list,
            # whose entries never carry '.vgpr_count' -- keep only maps that do
            if cur is None:
                cur = {}
            if key in cur and key == '.name' and '.vgpr_count' not in cur:
                cur = {}
            cur[key] = val.strip()
            if all(f in cur for f in ('.name', '.vgpr_count', '.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size')) \
                    and key == '.vgpr_spill_count':
                out.append(cur)
                cur = None
    res = []
    names = demangle([k['.name'] for k in out])
    for k, dn in zip(out, names):
        r = {f[1:]: (int(v) if v.lstrip('-').isdigit() else v) for f, v in k.items()}
        r['demangled'] = dn
        res.append(r)
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    only_spills = '--spills' in sys.argv
    objs = args or sorted(os.path.join(LIBDIR, f) for f in os.listdir(LIBDIR) if f.endswith('.o'))
    for o in objs:
        for k in kernels(o):
            spilled = k['vgpr_spill_count'] or k['private_segment_fixed_size']
            if only_spills and not spilled:
                continue
            print('%-22s vgpr %3d  sgpr %3s  vspill %3d  sspill %3d  scratch %4d B  lds %6s  %s' % (
                os.path.basename(o), k['vgpr_count'], k.get('sgpr_count', '?'), k['vgpr_spill_count'], k['sgpr_spill_count'],
                k['private_segment_fixed_size'], k.get('group_segment_fixed_size', '?'), k['demangled'][:110]))


if __name__ == '__main__':
    main()
