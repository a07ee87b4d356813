"""Can the DTW launch (bound by fp64 vector issue, every vector register of a CU at three wavefronts per SIMD) share the chip with the
memory- / latency-bound stages of a pass?  Each stage alone, then the DTW launch on one stream with a stage on another, started
together: `both` against `alone + alone` and `max(alone, alone)`.  Run with SGNN_DTW_LDS_PAD=24000 for two DTW workgroups per CU.
usage: [SGNN_DTW_LDS_PAD=24000] python tools/dtw_overlap_probe.py"""
import os
import sys
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
prep = {}
src = torch.from_numpy(np.random.default_rng(0).integers(1, n + 1, 183).astype(np.int32)).to(dev)
E = torch.randn(n + 1, 64, device=dev)
gE, mE, vE = torch.randn_like(E), torch.zeros_like(E), torch.zeros_like(E)
stages = {
    'dtw': lambda: ops.dtw_similarity(sets.ptr, ce, 20, a_sets.ptr, ae, 50, x_prep=prep),
    'bfs': lambda: ops.bfs_min_hops_to_sets(g, src, sets, max_hops=10),
    'khop1': lambda: ops.khop_border_sample(g, sets, 1, 43, 0, 77),
    'degseq': lambda: ops.degree_sequence(g, sets),
    'adam(table)': lambda: ops.adam_step(E, gE, mE, vE, 1e-3, (0.9, 0.999), 1e-8, 1),
    'walks': lambda: ops.triangular_walks(g, 0, 1050, 10, 0.65, 0, 5),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fs):
    """fs: list of (stream, fn) started together; ms until all are done"""
    for _, f in fs:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for st, f in fs:
            st.wait_event(e0)
            with torch.cuda.stream(st):
                f()
        for st, _ in fs:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


alone = {k: timed([(s1, f)]) for k, f in stages.items()}
print('pad', os.environ.get('SGNN_DTW_LDS_PAD', '0'), 'alone (ms):', {k: round(v, 3) for k, v in alone.items()})
for k in stages:
    if k == 'dtw':
        continue
    both = timed([(s1, stages['dtw']), (s2, stages[k])])
    print('dtw || %-12s both %.3f   sum %.3f   max %.3f   hidden %.0f %%' % (
        k, both, alone['dtw'] + alone[k], max(alone['dtw'], alone[k]), 100 * (alone['dtw'] + alone[k] - both) / alone[k]))
three = timed([(s1, stages['dtw']), (s2, lambda: (stages['bfs'](), stages['walks']())), (torch.cuda.Stream(), lambda: (stages['adam(table)'](), stages['degseq']()))])
print('dtw || (bfs, walks) || (adam, degseq): both %.3f   sum %.3f' % (three, alone['dtw'] + alone['bfs'] + alone['walks'] + alone['adam(table)'] + alone['degseq']))
