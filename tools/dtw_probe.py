"""Stand-alone driver of the DTW similarity kernel on benchmark-shaped inputs (for rocprofv3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import ops, synthetic, tape

n, m, S = 1_000_000, 10, 50_000
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
NX = int(sys.argv[2]) if len(sys.argv) > 2 else 20          # nodes per subgraph = rows of the DP
subs = synthetic.bfs_subgraphs(rowptr, col, S, NX, seed=1000)
dev = torch.device('cuda:0')
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
sets = ops.Ragged.from_lists(subs, dev)
patches = ops.triangular_walks(g, 0, 210, 50, 0.65, 0, tape.stream_id(tape.STREAM_STRUCT_PATCH))
a_sets = ops.Ragged.from_padded(patches)
ai, ae = ops.degree_sequence(g, a_sets)
ci, ce = ops.degree_sequence(g, sets)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for nm, x, y in (('internal', ci, ai), ('external', ce, ae)):
    ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = ops.dtw_similarity(sets.ptr, x, NX, a_sets.ptr, y, 50)
    torch.cuda.synchronize()
    print(nm, (time.perf_counter() - t) / reps * 1e3, 'ms', float(out.double().sum()))

# ---- where the wrapper's time goes (external side) -------------------------------------------
def _t(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

rows = ops.Ragged(sets.ptr, ce, max_len=NX).to_padded(width=NX, fill=-1, dtype=torch.int32)
print('to_padded', _t(lambda: ops.Ragged(sets.ptr, ce, max_len=NX).to_padded(width=NX, fill=-1, dtype=torch.int32)))
print('_unique_rows', _t(lambda: ops._unique_rows(rows)))
print('call without dedupe', _t(lambda: ops.dtw_similarity(sets.ptr, ce, NX, a_sets.ptr, ae, 50, dedupe=False)))
print('call without dedupe/order', _t(lambda: ops.dtw_similarity(sets.ptr, ce, NX, a_sets.ptr, ae, 50, dedupe=False, order_rows=False)))

# ---- processing orders of the x rows (external side): how much does grouping similar series buy? ----
if len(sys.argv) > 3 and sys.argv[3] == 'orders':
    R = rows.long()
    lens = (R >= 0).sum(1)
    Rz = R.clamp(min=0)
    def lex(cols):
        o = torch.arange(R.shape[0], device=dev)
        for c in reversed(cols):                     # least significant first, stable
            o = o[torch.sort(Rz[o, c], stable=True).indices]
        return o
    half = NX // 2
    cands = {
        'default (len, median, sum)': None,
        'random': torch.randperm(R.shape[0], device=dev),
        'identity': torch.arange(R.shape[0], device=dev),
        'lex first->last': lex(list(range(NX))),
        'lex last->first': lex(list(range(NX - 1, -1, -1))),
        'len, then lex last->first': None,
        'sum': torch.argsort(Rz.sum(1)),
        'len, max, sum': torch.argsort((lens << 50) | (Rz.max(1).values.clamp(max=(1 << 24) - 1) << 26) | Rz.sum(1).clamp(max=(1 << 26) - 1)),
        'quartile sums': torch.argsort((Rz[:, 3 * NX // 4:].sum(1).clamp(max=(1 << 20) - 1) << 40) | (Rz[:, NX // 2:3 * NX // 4].sum(1).clamp(max=(1 << 20) - 1) << 20) | Rz[:, :NX // 2].sum(1).clamp(max=(1 << 20) - 1)),
    }
    c16 = lambda c: Rz[:, c].clamp(max=0xFFFF)
    cands['first 4 packed'] = torch.argsort((c16(0) << 48) | (c16(1) << 32) | (c16(2) << 16) | c16(3))
    cands['x0, x1, median, sum'] = torch.argsort((Rz[:, 0].clamp(max=0xFFF) << 52) | (Rz[:, 1].clamp(max=0xFFF) << 40) | (Rz[:, NX // 2].clamp(max=0xFFFF) << 24) | Rz.sum(1).clamp(max=(1 << 24) - 1))
    cands['x0, x4, x8, x12 packed'] = torch.argsort((c16(0) << 48) | (c16(NX // 5) << 32) | (c16(2 * NX // 5) << 16) | c16(3 * NX // 5))
    cands['median, x0, sum'] = torch.argsort((Rz[:, NX // 2].clamp(max=0xFFFF) << 40) | (Rz[:, 0].clamp(max=0xFFF) << 28) | Rz.sum(1).clamp(max=(1 << 28) - 1))
    def packed(pos, bits):
        k = torch.zeros(R.shape[0], dtype=torch.int64, device=dev)
        for c in pos:
            k = (k << bits) | Rz[:, min(c, NX - 1)].clamp(max=(1 << bits) - 1)
        return torch.argsort(k)
    cands['B q(0,1/4,1/2,3/4) x16'] = packed([0, NX // 4, NX // 2, 3 * NX // 4], 16)
    cands['C q(1/5..4/5) x16'] = packed([NX // 5, 2 * NX // 5, 3 * NX // 5, 4 * NX // 5], 16)
    cands['D q(0..4/5) x12'] = packed([0, NX // 5, 2 * NX // 5, 3 * NX // 5, 4 * NX // 5], 12)
    cands['E q(1/2,1/4,3/4,0) x16'] = packed([NX // 2, NX // 4, 3 * NX // 4, 0], 16)
    cands['F q(0,1/3,2/3,last) x16'] = packed([0, NX // 3, 2 * NX // 3, NX - 1], 16)
    cands['G 6 quantiles x10'] = packed([0, NX // 6, 2 * NX // 6, 3 * NX // 6, 4 * NX // 6, 5 * NX // 6], 10)
    cands['H q(0,1/5,2/5) x16 + sum'] = torch.argsort((c16(0) << 48) | (c16(NX // 5) << 36) | (Rz[:, 2 * NX // 5].clamp(max=0xFFF) << 24) | Rz.sum(1).clamp(max=(1 << 24) - 1))
    def morton(fields, bits):
        k = torch.zeros(R.shape[0], dtype=torch.int64, device=dev)
        for b in range(bits - 1, -1, -1):
            for f in fields:
                k = (k << 1) | ((f >> b) & 1)
        return torch.argsort(k)
    qpos = [0, NX // 3, 2 * NX // 3, NX - 1]
    lin = [Rz[:, c].clamp(max=4095) for c in qpos]
    lg = [(8 * torch.log2(1.0 + Rz[:, c].double())).long().clamp(max=127) for c in qpos]
    cands['morton 4 quantiles x12 (linear)'] = morton(lin, 12)
    cands['morton 4 quantiles x7 (log)'] = morton(lg, 7)
    k = torch.zeros(R.shape[0], dtype=torch.int64, device=dev)
    for f in lg:
        k = (k << 7) | f
    cands['packed 4 quantiles x7 (log)'] = torch.argsort(k)
    lg6 = [(8 * torch.log2(1.0 + Rz[:, c].double())).long().clamp(max=127) for c in [0, NX // 5, 2 * NX // 5, 3 * NX // 5, 4 * NX // 5, NX - 1]]
    cands['morton 6 quantiles x7 (log)'] = morton(lg6, 7)
    o = lex(list(range(NX - 1, -1, -1)))
    cands['len, then lex last->first'] = o[torch.sort(lens[o], stable=True).indices]
    for name, o in cands.items():
        f = lambda: ops.dtw_similarity(sets.ptr, ce, NX, a_sets.ptr, ae, 50, dedupe=False, order=o)
        print('%-32s %8.3f ms' % (name, _t(f, 3)))
