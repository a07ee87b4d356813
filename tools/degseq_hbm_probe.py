#!/usr/bin/env python3
"""The structure-channel CSR gather (sgnn_degree_sequence) on a graph whose CSR does NOT fit the
256 MiB Infinity Cache: BA n = 8M, m = 16 (col ~1 GB, rowptr 64 MB), 50k sets of 20 nodes.

Two families of sets over the same graph:
  bfs     DENSITY-style BFS subgraphs (what the benchmark uses): they collect hubs, so most of the
          algorithmic bytes are hub lists that many sets re-read -- cache hits even on this graph;
  random  20 uniformly drawn nodes per set: no list is shared between sets to speak of, every byte has
          to come from HBM -- the regime where algorithmic bytes ~ memory-side bytes.

Prints one JSON line: time per launch (HIP events, back to back) of the streaming form (every list read
in full: the launch SURVEY.md 8(d)'s byte count describes) and of the shipped form (lists of >= 512
entries binary-searched), algorithmic bytes of both, rates against 8 TB/s; plus the calibration copies
(4 and 16 bytes per lane over a 1 GiB buffer) whose known byte counts scale the rocprofv3 counters:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_x -- python3 tools/degseq_hbm_probe.py --family random --reps 3
    python tools/pmc_summary.py gpurun_out/pmc_x            # per-kernel averages of the counter
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DS_SEARCH = 512          # csrc/degree_sequence.hip: lists of >= DS_SEARCH entries are searched, not streamed


def algorithmic_bytes(deg_of_member, set_size, search):
    """SURVEY.md 8(d) per set: sum_v (16 + 4 deg v) + 12 |S|.  Search form: a list of >= DS_SEARCH entries
    costs each of the |S| members floor(log2 deg) + 1 probes and one verifying read, 4 bytes each, instead
    of 4 deg."""
    d = deg_of_member.astype(np.int64)
    per_list = 4 * d
    if search == 'bits':
        # round 6: the list has a membership bitmap -- each of the |S| members reads ONE 4-byte word of it; every member also
        # reads its hub_index entry (4 bytes)
        per_list = np.where(d >= DS_SEARCH, 4 * set_size, per_list) + 4
    elif search:
        steps = np.floor(np.log2(np.maximum(d, 1))).astype(np.int64) + 2
        per_list = np.where(d >= DS_SEARCH, np.minimum(4 * d, 4 * steps * set_size), per_list)
    return int((16 + per_list).sum() + 12 * len(d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=8_000_000)
    ap.add_argument('--m', type=int, default=16)
    ap.add_argument('--sets', type=int, default=50_000)
    ap.add_argument('--set-nodes', type=int, default=20)
    ap.add_argument('--family', choices=['bfs', 'random'], default='bfs')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--calib-mib', type=int, default=1024)
    ap.add_argument('--benchmark-graph', action='store_true',
                    help="bench.py's own inputs instead: numpy BA n=1M m=10 seed 42 (CSR 88 MB: cache resident), rank 0's 50k BFS sets")
    args = ap.parse_args()
    from subgnn_amd import ops, synthetic
    dev = torch.device('cuda:0')
    t0 = time.time()
    if args.benchmark_graph:
        args.nodes, args.m, args.family = 1_000_000, 10, 'bfs'
        rp, cl = synthetic.sorted_csr(synthetic.barabasi_albert_edges(args.nodes, args.m, seed=42), args.nodes)
        g = ops.DeviceGraph(rp, cl, np.arange(1, args.nodes + 1, dtype=np.int32), dev)
        rowptr, col = g.rowptr, g.col
    else:
        rowptr, col = synthetic.barabasi_albert_csr_device(args.nodes, args.m, 42, dev)
        g = ops.DeviceGraph.from_device_csr(rowptr, col)
    torch.cuda.synchronize()
    t_graph = time.time() - t0
    K, S = args.set_nodes, args.sets
    if args.family == 'bfs':
        subs = synthetic.bfs_subgraphs(rowptr.cpu().numpy(), col.cpu().numpy(), S, K, seed=1000)
        subs = [s for s in subs if len(s) == K]
    else:
        rng = np.random.default_rng(1000)
        subs = [np.sort(rng.choice(args.nodes, K, replace=False) + 1).tolist() for _ in range(S)]
    sets = ops.Ragged.from_lists(subs, dev)
    order = ops.heaviest_first(g, sets)
    deg = (rowptr[1:] - rowptr[:-1])[sets.nodes[:sets.total].long()].cpu().numpy()
    out = {'graph': 'BA n=%d m=%d (%s generator, seed 42)' % (args.nodes, args.m, 'numpy: bench.py\'s graph' if args.benchmark_graph else 'torch'), 'nnz': g.nnz,
           'csr_bytes': int(g.nnz * 4 + rowptr.numel() * 8), 'infinity_cache_bytes': 256 << 20,
           'family': args.family, 'sets': len(subs), 'set_nodes': K, 'graph_build_s': round(t_graph, 1),
           'mean_member_degree': float(deg.mean()), 'max_member_degree': int(deg.max()),
           'distinct_members': int(torch.unique(sets.nodes[:sets.total]).numel())}
    # bytes of the DISTINCT lists the launch touches: what has to come from HBM at least once when nothing is resident
    um = torch.unique(sets.nodes[:sets.total]).long()
    out['distinct_list_bytes'] = int(((rowptr[um + 1] - rowptr[um]) * 4 + 16).sum().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    same = None
    for form, search in (('streaming', False), ('shipped_search', True)):
        res = ops.degree_sequence(g, sets, order=order, search_long_lists=search)
        same = res if same is None else bool(torch.equal(same[0], res[0]) and torch.equal(same[1], res[1]))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            ops.degree_sequence(g, sets, order=order, search_long_lists=search)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        ab = algorithmic_bytes(deg, K, ('bits' if g.hub_tables() is not None else True) if search else False)
        out[form] = {'ms_per_launch': ms, 'algorithmic_bytes_per_launch': ab, 'achieved_GBs': ab / ms / 1e6,
                     'frac_of_8TBs': ab / ms / 1e6 / 8000.0}
    out['forms_agree'] = same
    # calibration copies: known byte counts in the gather's access width (4 B/lane) and the wide one (16 B/lane)
    nb = args.calib_mib << 20
    a = torch.empty(nb // 4, dtype=torch.int32, device=dev).random_()
    b = torch.empty_like(a)
    out['calibration'] = {'bytes_read': nb, 'bytes_written': nb}
    for w in (4, 16):
        ops.probe_stream_copy(a, b, w)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            ops.probe_stream_copy(a, b, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out['calibration']['copy_%dB_per_lane' % w] = {'ms': ms, 'GBs_read_plus_write': 2 * nb / ms / 1e6}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
