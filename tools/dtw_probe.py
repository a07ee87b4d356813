"""Stand-alone driver of the DTW similarity kernel on benchmark-shaped inputs (for rocprofv3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic, tape

n, m, S = 1_000_000, 10, 50_000
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
NX = int(sys.argv[2]) if len(sys.argv) > 2 else 20          # nodes per subgraph = rows of the DP
subs = synthetic.bfs_subgraphs(rowptr, col, S, NX, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
patches = ops.triangular_walks(g, 0, 210, 50, 0.65, 0, tape.stream_id(tape.STREAM_STRUCT_PATCH))
a_sets = ops.Ragged.from_padded(patches)
ai, ae = ops.degree_sequence(g, a_sets)
ci, ce = ops.degree_sequence(g, sets)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for nm, x, y in (('internal', ci, ai), ('external', ce, ae)):
    ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50)
    torch.cuda.synchronize()
    print(nm, (time.perf_counter() - t) / reps * 1e3, 'ms', float(out.double().sum()))

# ---- where the wrapper's time goes (external side) -------------------------------------------
def _t(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

rows = ops.Ragged(sets.ptr, ce, max_len=NX).to_padded(width=NX, fill=-1, dtype=torch.int32)
print('to_padded', _t(lambda: ops.Ragged(sets.ptr, ce, max_len=NX).to_padded(width=NX, fill=-1, dtype=torch.int32)))
print('_unique_rows', _t(lambda: ops._unique_rows(rows)))
print('call without dedupe', _t(lambda: ops.dtw_similarity(sets.ptr, ce, NX, a_sets.ptr, ae, 50, dedupe=False)))
print('call without dedupe/order', _t(lambda: ops.dtw_similarity(sets.ptr, ce, NX, a_sets.ptr, ae, 50, dedupe=False, order_rows=False)))
