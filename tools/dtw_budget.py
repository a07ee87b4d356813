#!/usr/bin/env python3
"""Per-level budget of the DTW register kernel on the benchmark's external side (50k component rows x 210 patches): what a
wavefront (= one patch x 64 component rows) executes on every fastdtw level -- cells evaluated by the wavefront (the union of
its lanes' windows) against cells of the lanes' own windows, columns swept, (column, row pair) visits, back-trace steps.
Needs the counting build:

    SGNN_HIPCC_FLAGS=-DDTW_PROBE_COUNT python -c "import os; from subgnn_amd import build; os.utime(os.path.join(build.CSRC, 'dtw.hip')); build.build()"
    python tools/dtw_budget.py [external|internal] > profiles/r05_dtw_budget.json
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import _lib, ops, synthetic, tape

side = sys.argv[1] if len(sys.argv) > 1 else 'external'
n, m, S, NX = 1_000_000, 10, 50_000, 20
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, NX, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
patches = ops.triangular_walks(g, 0, 210, 50, 0.65, 0, tape.stream_id(tape.STREAM_STRUCT_PATCH))
a_sets = ops.Ragged.from_padded(patches)
ai, ae = ops.degree_sequence(g, a_sets)
ci, ce = ops.degree_sequence(g, sets)
x, y = (ce, ae) if side == 'external' else (ci, ai)
lib = _lib.load()
fn = getattr(lib, 'sgnn_dtw_probe_counts', None)
if fn is None:
    raise SystemExit('libsubgnn_hip.so was not built with -DDTW_PROBE_COUNT')
fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
prep = {}
ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50, x_prep=prep)       # (keeps the grouping / order: the counted call is the steady one)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
fn(None, 1)
out = ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50, x_prep=prep)
torch.cuda.synchronize()
fn(buf, 0)
c = np.array(list(buf), dtype=np.float64).reshape(8, 8)
levels = {}
for lev in range(8):
    w = c[lev, 0]
    if w == 0:
        continue
    pairs = c[lev, 6]
    levels['level %d' % lev] = {
        'wavefront_levels': int(w), 'pairs_per_wavefront': round(pairs / w, 2),
        'cells_evaluated_per_wavefront(union)': round(c[lev, 1] / w, 1),
        'cells_needed_per_pair(own window)': round(c[lev, 2] / max(pairs, 1), 1),
        'evaluated_over_needed': round((c[lev, 1] / w) / max(c[lev, 2] / max(pairs, 1), 1e-9), 3),
        'columns_per_wavefront': round(c[lev, 3] / w, 2), 'row_pair_visits_per_wavefront': round(c[lev, 4] / w, 2),
        'backtrace_steps_per_pair': round(c[lev, 5] / max(pairs, 1), 2)}
print(json.dumps({'side': side, 'checksum': float(out.double().sum()), 'levels': levels}, indent=1))
