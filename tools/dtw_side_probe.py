"""Times ONE side of the benchmark's DTW call (external by default) without the wrapper's grouping, for rocprofv3 / counter
passes and compile-time variant sweeps (tools/tune_dtw.sh):  python tools/dtw_side_probe.py [external|internal] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic, tape

side = sys.argv[1] if len(sys.argv) > 1 else 'external'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
prep = {}
f = lambda: ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50, x_prep=prep)
out = f(); f()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    out = f()
torch.cuda.synchronize()
print('%s %.3f ms per call (kept preparation), checksum %.6f' % (side, (time.perf_counter() - t) / reps * 1e3, float(out.double().sum())))
if os.environ.get('DTW_PROBE_DUPS'):
    # how often a column of the finest level repeats its predecessor's value (the y series is wave-uniform: a repeated value
    # means the column's costs are the previous column's for every row and lane)
    yp = ops.Ragged(a_sets.ptr, y, max_len=50).to_padded(width=50, fill=-1, dtype=torch.int32).cpu().numpy()
    same = (yp[:, 1:] == yp[:, :-1]) & (yp[:, 1:] >= 0)
    print('y series: %d x %d, mean length %.1f, columns equal to their predecessor: %.3f' % (yp.shape[0], yp.shape[1], (yp >= 0).sum(1).mean(), same.sum() / max(1, (yp[:, 1:] >= 0).sum())))
    h = (yp[:, 0::2][:, :25].astype(np.float64) + yp[:, 1::2][:, :25]) / 2
    print('level-1 columns equal to their predecessor: %.3f' % ((h[:, 1:] == h[:, :-1]).mean()))
    xp = ops.Ragged(sets.ptr, x, max_len=NX).to_padded(width=NX, fill=-1, dtype=torch.int32).cpu().numpy()
    print('x rows: entries equal to their predecessor: %.3f; pairs (2k, 2k+1) equal: %.3f' % ((xp[:, 1:] == xp[:, :-1]).mean(), (xp[:, 0::2] == xp[:, 1::2]).mean()))
    print('distinct y values per series: %.1f' % np.mean([len(set(r[r >= 0])) for r in yp]))
