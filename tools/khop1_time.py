#!/usr/bin/env python3
"""ms per launch of the one-hop border + neighbourhood-border draw (sgnn_khop_border_sample, k = 1) on the benchmark's inputs,
HIP events, with a checksum of anchors / hop levels / border sizes (compile-time variants: tools/tune_khop.sh).
    python tools/khop1_time.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic, tape

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, m, S = 1_000_000, 10, 50_000
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
st = tape.stream_id(tape.STREAM_N_BOR, 'train', 0)
a, w, c = ops.khop_border_sample(g, sets, 1, 43, 0, st)
width = c.max().view(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    a, w, c = ops.khop_border_sample(g, sets, 1, 43, 0, st, width=width)
e1.record()
torch.cuda.synchronize()
print('khop1 %.4f ms per launch (incl. the finish launch), checksum anchors %d sims %.1f counts %d' % (
    e0.elapsed_time(e1) / reps, int(a.sum()), float(w.double().sum()), int(c.sum())))
