"""Debug build only (SGNN_HIPCC_FLAGS=-DK1_DEBUG_TIMING): where the one-hop border + draw kernel's
time goes, from clock64() deltas accumulated by thread 0 of every workgroup."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from subgnn_amd import ops, synthetic, _lib
n, m, S = 1_000_000, 10, 50_000
rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, m, seed=42), n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
lib = _lib.load()
ops.khop_border_sample(g, sets, 1, 43, 0, 77); torch.cuda.synchronize()
lib.sgnn_debug_k1_timing(None, 1)
ops.khop_border_sample(g, sets, 1, 43, 0, 77); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lib.sgnn_debug_k1_timing(out, 0)
names = ['dispatch', 'expansion: barrier wait + prefetch issue', 'members un-set', 'rank table', 'draw', 'wipe', 'expansion: tile ready', 'expansion: own chunks']
tot = sum(out[:8])
for nm, v in zip(names, out):
    print('%-16s %6.2f %%   %8.1f clock64 ticks per set' % (nm, 100.0 * v / tot, v / S))
