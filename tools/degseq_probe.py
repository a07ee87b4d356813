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

if len(sys.argv) > 2 and sys.argv[2] == 'parts':
    # where the launch's time goes: without the in-register sort; with only the table build (no edges)
    def tm(**kw):
        ops.degree_sequence(g, sets, order=order, **kw); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): ops.degree_sequence(g, sets, order=order, **kw)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    print('ordered, sorted output  ', tm(sort=True))
    print('ordered, unsorted output', tm(sort=False))

if len(sys.argv) > 2 and sys.argv[2] == 'heavy':
    # critical path: the heaviest sets alone (a set is one wavefront; its lists are streamed 8 x 256 B at a time)
    deg = torch.from_numpy(np.diff(rowptr)).to(dev)
    tot = deg[sets.nodes[:S * 20].long()].view(S, 20).sum(1)
    o = torch.argsort(tot, descending=True)
    print('sum of member degrees: max', int(tot.max()), 'mean', float(tot.float().mean()), 'top8', tot[o[:8]].tolist())
    lists = sets.to_lists()
    for K in (1, 4, 64, 1024):
        sub = ops.Ragged.from_lists([lists[int(i)] for i in o[:K].tolist()], dev)
        ops.degree_sequence(g, sub); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): ops.degree_sequence(g, sub)
        torch.cuda.synchronize(); print('heaviest %4d sets alone: %.1f us' % (K, (time.perf_counter() - t) / 20 * 1e6))

if len(sys.argv) > 2 and sys.argv[2] == 'orders':
    deg = torch.from_numpy(np.diff(rowptr)).to(dev)
    tot = deg[sets.nodes[:S * 20].long()].view(S, 20).sum(1)
    desc = torch.argsort(tot, descending=True)
    def tm(o):
        o = o.to(torch.int32).contiguous()
        ops.degree_sequence(g, sets, order=o); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): ops.degree_sequence(g, sets, order=o)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    print('descending', tm(desc))
    print('random', tm(torch.randperm(S, device=dev)))
    for k in (2, 4, 8):
        # heavy sets dealt out every k-th position of the first part of the order, light ones (from the light end) between them
        nh = S // k
        o = torch.empty(S, dtype=torch.int64, device=dev)
        heavy, light = desc[:nh], desc[nh:].flip(0)
        pos = torch.arange(S, device=dev)
        is_h = (pos % k == 0) & (pos // k < nh)
        o[is_h] = heavy[: int(is_h.sum())]
        o[~is_h] = light[: int((~is_h).sum())]
        print('every %d-th position heavy, lightest between' % k, tm(o))
        o2 = torch.empty(S, dtype=torch.int64, device=dev)
        o2[is_h] = heavy[: int(is_h.sum())]
        o2[~is_h] = desc[nh:][: int((~is_h).sum())]
        print('every %d-th position heavy, next-heaviest between' % k, tm(o2))

if len(sys.argv) > 2 and sys.argv[2] == 'search':
    def tm(**kw):
        ops.degree_sequence(g, sets, order=order, **kw); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): ops.degree_sequence(g, sets, order=order, **kw)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
    a = ops.degree_sequence(g, sets, order=order, search_long_lists=False)
    b = ops.degree_sequence(g, sets, order=order, search_long_lists=True)
    print('same results', bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])))
    print('streamed', tm(search_long_lists=False), 'ms;  long lists searched', tm(search_long_lists=True), 'ms')
