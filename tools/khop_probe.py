"""Timing split of the fused border BFS + anchor draw on the benchmark inputs."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic, _lib

n, m, S = 1_000_000, 10, 50_000
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
lib = _lib.load()

def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

def count_only(lds):
    ws, wsb = ops._khop_ws(lib, g, sets.n, lds)
    counts = torch.zeros(sets.n, dtype=torch.int64, device=dev)
    _lib.check(lib.sgnn_khop_border(ops._ptr(g.rowptr), ops._ptr(g.col), g.nnz, g.max_id, ops._ptr(sets.ptr), ops._ptr(sets.nodes),
               sets.n, 1, 0, ops._ptr(counts), None, None, None, ops._ptr(ws), wsb, 1 if lds else 0, ops._stream()), 'khop')
    return counts

print('BFS count-only, LDS bitmap', timeit(lambda: count_only(True)))
print('BFS count-only, global bitmap', timeit(lambda: count_only(False)))
for A in (4, 16, 43):
    print('BFS + draw (bitmap rank query), %d slots' % A, timeit(lambda: ops.khop_border_sample(g, sets, 1, A, 0, 77)))
order = ops.heaviest_first(g, sets)
print('BFS + draw, 43 slots, heaviest-first dispatch', timeit(lambda: ops.khop_border_sample(g, sets, 1, 43, 0, 77, order=order)))
c = count_only(True)
print('border entries total', int(c.sum()), 'mean', float(c.float().mean()), 'max', int(c.max()))
