"""Timing of the multi-source BFS (position channel) on the benchmark graph for several push/pull
switch points (the pull_alpha argument), checking that every setting returns the same hop table."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic

n, m = 1_000_000, 10
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)


def timeit(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3


for ns in (57, 183, 1000):
    src = torch.from_numpy(np.random.default_rng(5).integers(1, n + 1, ns).astype(np.int32)).to(dev)
    ref = None
    for alpha in (0, 16, 64, 256, 1024, 4096, 1 << 30):
        ms = timeit(lambda: ops.bfs_hops(g, src, max_hops=32, node_major=True, pull_alpha=alpha))
        d = ops.bfs_hops(g, src, max_hops=32, node_major=True, pull_alpha=alpha)
        if ref is None:
            ref = d
        print('sources %5d  alpha %10d  %8.3f ms  same=%s  max hop %d' % (ns, alpha, ms, bool(torch.equal(d, ref)),
                                                                         int(d[d != 255].max())))

# the fused form the position channel uses (min over the members of 50k component sets)
rng = np.random.default_rng(7)
sets = ops.Ragged.from_padded(torch.from_numpy(rng.integers(1, n + 1, (50_000, 20)).astype(np.int64)).to(dev))
src = torch.from_numpy(np.random.default_rng(5).integers(1, n + 1, 183).astype(np.int32)).to(dev)
ref = None
for alpha in (0, 2, 4, 8, 16, 32, 64, 128, 256, 1024):
    ms = timeit(lambda: ops.bfs_min_hops_to_sets(g, src, sets, max_hops=32, pull_alpha=alpha), reps=10)
    w = ops.bfs_min_hops_to_sets(g, src, sets, max_hops=32, pull_alpha=alpha)
    if ref is None:
        ref = w
    print('min-hops-to-sets 183 sources  alpha %6d  %8.3f ms  same=%s' % (alpha, ms, bool(torch.equal(w, ref))))
