"""DTW launch time per predecessor rule on the benchmark's external side.  usage: python tools/dtw_tie_probe.py"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from subgnn_amd import ops, synthetic

sys.argv = sys.argv[:1]
args = bench.parse()
rowptr, col, subs, _, _ = bench.build_inputs(args, 0, 1)
dev = torch.device('cuda', 0)
g = ops.DeviceGraph(rowptr, col, np.arange(1, args.nodes + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
patches = synthetic.bfs_subgraphs(rowptr, col, 210, 50, seed=6)
a_sets = ops.Ragged.from_lists(patches, dev)
ci, ce = ops.degree_sequence(g, sets)
ai, ae = ops.degree_sequence(g, a_sets)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for tie in (0, 1, 2):
    for side, x, y in (('ext', ce, ae), ('int', ci, ai)):
        prep = {}
        ops.dtw_similarity(sets.ptr, x, 20, a_sets.ptr, y, 50, tie, x_prep=prep)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            r = ops.dtw_similarity(sets.ptr, x, 20, a_sets.ptr, y, 50, tie, x_prep=prep)
        e1.record()
        torch.cuda.synchronize()
        out['tie%d_%s_ms' % (tie, side)] = round(e0.elapsed_time(e1) / 5, 3)
        out['tie%d_%s_sum' % (tie, side)] = float(r.double().sum())
print(json.dumps(out))
