"""Stand-alone driver of the structure-channel CSR gather on the benchmark inputs (for rocprofv3
--pmc passes): builds the 1M-node graph + 50k subgraph sets and launches sgnn_degree_sequence."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic

n, m, S = 1_000_000, 10, 50_000
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ONLY_ORDERED = len(sys.argv) > 2 and sys.argv[2] == 'ordered'      # PMC passes: only the benchmark's form of the launch
for srt in (() if ONLY_ORDERED else (True,)):
    ops.degree_sequence(g, sets, sort=srt)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ops.degree_sequence(g, sets, sort=srt)
    torch.cuda.synchronize()
    print('sorted' if srt else 'unsorted', (time.perf_counter() - t) / reps * 1e3, 'ms')

order = ops.heaviest_first(g, sets)
ops.degree_sequence(g, sets, order=order)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    oi2, oe2 = ops.degree_sequence(g, sets, order=order)
torch.cuda.synchronize()
print('heaviest first', (time.perf_counter() - t) / reps * 1e3, 'ms')
if not ONLY_ORDERED:
    oi, oe = ops.degree_sequence(g, sets)
    print('same results', bool(torch.equal(oi, oi2) and torch.equal(oe, oe2)))
