"""Build libsubgnn_hip.so for gfx950 with hipcc (in-tree, so that it travels with the repo
snapshot to the GPU box).  ``python -m subgnn_amd.build`` or ``__graft_entry__.build()``."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libsubgnn_hip.so')
SOURCES = ['lib.hip', 'degree_sequence.hip', 'graph_sets.hip', 'samplers.hip', 'similarity.hip', 'dtw.hip', 'embed.hip', 'mpn.hip', 'attention.hip', 'lstm.hip', 'probe.hip', 'scatter.hip', 'update.hip', 'optim.hip', 'readout.hip', 'loss.hip', 'head.hip']
ARCH = 'gfx950'
# dtw.hip: no NaN can arise in the DP (costs are finite or +inf, only min and + are applied); telling the compiler so removes
# the canonicalising v_max x, x it otherwise puts in front of every v_min_f64 (3 of 16 instructions per cell)
# -disable-promote-alloca-to-vector: the row registers are arrays indexed statically once the row loops are unrolled, but the
# AMDGPU alloca-to-vector promotion runs before that and turns arrays of <= 16 doubles into one vector value -- every update
# of a row then copied the whole array (32 v_mov per row pair on the coarse levels)
EXTRA_FLAGS = {'dtw.hip': ['-fno-honor-nans', '-mllvm', '-disable-promote-alloca-to-vector']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'common.h'),
                                                      os.path.join(HERE, '..', 'include', 'subgnn_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        src = os.path.join(CSRC, s)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(
                os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, 'common.h')),
                os.path.getmtime(os.path.join(HERE, '..', 'include', 'subgnn_hip.h'))):
            cmd = [_hipcc(), '--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics',
                   '-Wall', '-Wno-unused-function'] + EXTRA_FLAGS.get(s, []) + os.environ.get('SGNN_HIPCC_FLAGS', '').split() + ['-c', src, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
