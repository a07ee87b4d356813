"""Per-level statistics of the position channel's multi-source BFS on the benchmark graph (183 sources): what a level's
expansion has to touch in either direction.  From the hop table of ops.bfs_hops."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from subgnn_amd import ops, synthetic
n, m = 1_000_000, 10
rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, m, seed=42), n)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 183
src = torch.from_numpy(np.random.default_rng(5).integers(1, n + 1, ns).astype(np.int32)).to(dev)
d = ops.bfs_hops(g, src, max_hops=32, node_major=True)[1:]          # (n, ns) uint8
deg = torch.from_numpy(np.diff(rowptr)[1:]).to(dev)
nnz = int(deg.sum())
print('nodes', n, 'edges(directed)', nnz, 'sources', ns)
maxl = int(d[d != 255].max())
for L in range(1, maxl + 1):
    new = (d == L)                                   # pairs discovered at level L
    fr_nodes = new.any(1)                            # nodes on the frontier after level L
    before = (d >= L)                                # pairs missing before level L (incl. unreachable)
    inc_nodes = before.any(1)
    prev_new = (d == L - 1)
    pf_nodes = prev_new.any(1)
    print('level %d: new pairs %10d  frontier-after nodes %8d | push side: prev frontier nodes %8d, their edges %10d | '
          'pull side: incomplete nodes %8d, their edges %10d, missing pairs %11d' %
          (L, int(new.sum()), int(fr_nodes.sum()), int(pf_nodes.sum()), int(deg[pf_nodes].sum()),
           int(inc_nodes.sum()), int(deg[inc_nodes].sum()), int(before.sum())))
