#!/usr/bin/env python3
"""The position channel's multi-source BFS alone on the benchmark graph (BA n=1M m=10, 183 sources, 50k component sets):
ms per search by HIP events for the forms the pass runs (levels and push levels capped from the first search's status) and the
uncapped one.  Under ``rocprofv3 --kernel-trace --stats`` it gives the per-level kernel times.

    python tools/bfs_probe.py [--reps 10] [--sets 50000]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--sets', type=int, default=50000)
    ap.add_argument('--sources', type=int, default=183)
    args = ap.parse_args()
    from subgnn_amd import ops, synthetic
    dev = torch.device('cuda:0')
    n = 1_000_000
    edges = synthetic.barabasi_albert_edges(n, 10, seed=42)
    rowptr, col = synthetic.sorted_csr(edges, n)
    g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
    subs = synthetic.bfs_subgraphs(rowptr, col, args.sets, 20, seed=1000)
    sets = ops.Ragged.from_lists(subs, dev)
    src = torch.from_numpy(np.random.default_rng(2).integers(1, n + 1, args.sources).astype(np.int32)).to(dev)
    ref, st = ops.bfs_min_hops_to_sets(g, src, sets, max_hops=32, want_status=True)
    last, more, first_pull, _ = st.tolist()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(**kw):
        out = ops.bfs_min_hops_to_sets(g, src, sets, **kw)
        assert torch.equal(out, ref), kw
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            ops.bfs_min_hops_to_sets(g, src, sets, **kw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps
    res = {'levels': last, 'first_pull_level': first_pull,
           'ms_uncapped(32 levels, all may push)': timed(max_hops=32),
           'ms_levels_capped': timed(max_hops=last + 3),
           'ms_levels_and_push_capped(the pass)': timed(max_hops=last + 3, push_levels=first_pull),
           'ms_push_levels_1': timed(max_hops=last + 3, push_levels=1),
           'ms_always_push': timed(max_hops=last + 3, pull_alpha=0)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
