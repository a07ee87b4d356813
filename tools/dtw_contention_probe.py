"""How long do chains of small kernels take on one stream while the DTW launch runs on another?
(the training half of a pipelined pass next to the preparation of the next one)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic, tape

n, m, S = 1_000_000, 10, 50_000
rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, m, seed=42), n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
patches = ops.triangular_walks(g, 0, 210, 50, 0.65, 0, tape.stream_id(tape.STREAM_STRUCT_PATCH))
a_sets = ops.Ragged.from_padded(patches)
ai, ae = ops.degree_sequence(g, a_sets)
ci, ce = ops.degree_sequence(g, sets)
def dtw():
    return ops.dtw_similarity(sets.ptr, ce, 20, a_sets.ptr, ae, 50, dedupe=False)
tiny = torch.randn(64, device=dev)
mid = torch.randn(50000, 64, device=dev)
big = torch.randn(50000, 448, device=dev)
w = torch.randn(64, 64, device=dev)
chains = {'100 x tiny (64 floats)': lambda: [tiny.add_(1.0) for _ in range(100)],
          '100 x mid (50000 x 64 element-wise)': lambda: [mid.mul_(1.0001) for _ in range(100)],
          '100 x big (50000 x 448 element-wise)': lambda: [big.mul_(1.0001) for _ in range(100)],
          '100 x GEMM 50000x64x64': lambda: [mid @ w for _ in range(100)]}
side, main = torch.cuda.Stream(), torch.cuda.current_stream()
dtw(); torch.cuda.synchronize()
for name, f in chains.items():
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dtw()
    time.sleep(0.002)                       # the DTW launch is running by now (7 ms)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    print('%-40s alone %.3f ms, beside the DTW launch %.3f ms' % (name, alone, e0.elapsed_time(e1)))
