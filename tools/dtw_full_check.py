import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from subgnn_amd import ops, synthetic, tape as T, _lib
n, m, S, K = 1_000_000, 10, 50_000, 20
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
subs = synthetic.bfs_subgraphs(rowptr, col, S, K, 1)
sets = ops.Ragged.from_lists(subs, dev)
walks = ops.triangular_walks(g, 0, 210, 50, 0.65, 0, T.stream_id(T.STREAM_STRUCT_PATCH))
a_sets = ops.Ragged.from_padded(walks)
ai, ae = ops.degree_sequence(g, a_sets)
ci, ce = ops.degree_sequence(g, sets)
for nm, cx, ax in (('ext', ce, ae), ('int', ci, ai)):
    fast = ops.dtw_similarity(sets.ptr, cx, K, a_sets.ptr, ax, 50)
    gen = ops.dtw_similarity(sets.ptr, cx, K, a_sets.ptr, ax, 50, kernel=1)
    print(nm, 'equal', bool(torch.equal(fast, gen)), 'n diff', int((fast != gen).sum()))
