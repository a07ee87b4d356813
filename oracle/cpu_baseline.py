"""cpu_baseline leg of bench.py: the oracle (plain C + numpy + torch-CPU restatement of the
same sparse algorithm the GPU path runs) timed on the host cores.  TEST INFRASTRUCTURE.

The reference's own Python cannot run at this scale (dense N x N structures, SURVEY.md section 7
hard part 5) and does not travel to the GPU box, so kind = "port".  A full CPU pass over the
50k-subgraph shard would take hours; the baseline is therefore measured on a BOUNDED sample and
extrapolated linearly, stage by stage:

  per-subgraph stages (components, 1-hop border, N/P anchor draws, degree sequences, DTW against
      all structure patches, position similarities, forward + backward + Adam)  -> timed on
      ``n_sample`` subgraphs, scaled by S / n_sample;
  shared stages (structure patches + their walks: timed on a few patches and scaled; one BFS per P-border anchor
      over the whole graph: run for ALL sources in C, the sources shared among the host's cores) -> counted once
      per pass.
The C stages (DTW, BFS) use OpenMP over the host's cores, the torch stages torch's thread pool, and (round 4) the
Python / numpy integer stages -- structure patches + walks, components, borders, anchor draws -- are dealt to a pool of
worker PROCESSES, one per core (``start_pool``: forked by bench.py BEFORE the GPU is initialised, so the workers hold no
HIP state; without a pool the stages run in this process, single-threaded).  "cores" reports the thread / process count.

``calibration``: profiles/r06_cpu_calibration.json (round 4: r04_...) holds the imported REFERENCE and this port timed on the same inputs on
the build container's 8 cores (tools/bench_density_n.py --mode reference / port: BASELINE.json configs[0]'s dataset, with
the neighbourhood channel only and with all three channels): ratio = reference time / port time, so
value / ratio estimates "the reference's own CPU path" at a size where the reference itself cannot run (BASELINE.md
section 3, steps i-iii).
"""
import json
import os
import time

import numpy as np
import torch

from . import cbind, float_half as FH, integer_half as IH, tape as T
from .graph import CSRGraph

_POOL = None
_W = {}            # what the workers see (set before the fork)


def _graph_identity(rowptr, col):
    """(n, nnz, checksum over a stride of the arrays): tells the graph the pool was forked with from another one."""
    rp, c = np.asarray(rowptr), np.asarray(col)
    step_r, step_c = max(1, len(rp) // 4096), max(1, len(c) // 4096)
    return (len(rp) - 1, int(rp[-1]) if len(rp) else 0, int(rp[::step_r].astype(np.int64).sum()), int(c[::step_c].astype(np.int64).sum()))


def start_pool(rowptr, col, procs=None):
    """Fork the worker processes (call before anything initialises the GPU).  The graph arrays are shared copy-on-write.
    A pool that exists for another graph is replaced."""
    global _POOL
    import multiprocessing as mp
    if _POOL is not None and _W.get('G_id') != _graph_identity(rowptr, col):
        stop_pool()
    _W['G'] = CSRGraph(rowptr, col)
    _W['G_id'] = _graph_identity(rowptr, col)
    procs = procs or os.cpu_count() or 1
    if procs > 1 and _POOL is None:
        _POOL = (mp.get_context('fork').Pool(procs), procs)
    return _POOL


def stop_pool():
    global _POOL
    if _POOL is not None:
        _POOL[0].terminate()
        _POOL = None


def _pmap(fn, items, timeout=600):
    """fn over items on the pool (order kept); in this process when there is no pool."""
    items = list(items)
    if _POOL is None or len(items) < 2:
        return [fn(x) for x in items]
    chunk = max(1, len(items) // (4 * _POOL[1]))
    return _POOL[0].map_async(fn, items, chunksize=chunk).get(timeout)


def _w_patch(a):
    i, walk_len, beta, seed = a
    return IH.triangular_walk(_W['G'], walk_len, beta, IH._Draws(seed, T.stream_id(T.STREAM_STRUCT_PATCH), i), 'graph')


def _w_walks(a):
    row, W, Tn, beta, seed = a
    G = _W['G']
    return (IH.perform_random_walks(G, row[None, :], W, Tn, beta, True, seed)[0],
            IH.perform_random_walks(G, row[None, :], W, Tn, beta, False, seed)[0])


def _w_components(s):
    return IH.connected_components(_W['G'], s)


def _w_border(r):
    G = _W['G']
    members = r[r != 0]
    if len(members) == 0:
        return np.zeros(0, dtype=np.int64)
    nb = np.unique(np.concatenate([G.neighbors(int(v)) for v in members]))
    return np.setdiff1d(nb, members).astype(np.int64)


def _w_anchors(a):
    r, ids, A, kind, width, seed = a
    out = np.zeros(A, dtype=np.int64)
    if len(ids):
        real = np.sort(ids)
        st = T.stream_id(kind, 'train', 0)
        for i in range(A):
            k = T.nanchor_pick(seed, st, r * A + i, len(real), len(real) < width)
            out[i] = 0 if k < 0 else real[k]
    return out


def _calibration():
    """The newest calibration file under profiles/ (round 6's, else round 4's) -> (dict, file name) or None."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
    for name in ('r06_cpu_calibration.json', 'r04_cpu_calibration.json'):
        f = os.path.join(d, name)
        if os.path.exists(f):
            with open(f) as fh:
                cal = json.load(fh)
            cal['_file'] = 'profiles/' + name
            return cal
    return None


def run(rowptr, col, subs, hp, emb, labels, n_sample, S_total):
    t_all = time.perf_counter()
    if _POOL is not None and _W.get('G_id') != _graph_identity(rowptr, col):
        # the pool's workers hold the graph start_pool() forked into them: a run() on ANOTHER graph must not use it
        stop_pool()
    G = _W['G'] = CSRGraph(rowptr, col) if _POOL is None else _W['G']
    n = G.n
    seed = int(hp.get('seed', 0))
    rng = np.random.default_rng(0)
    sample = [subs[i] for i in rng.choice(len(subs), n_sample, replace=False)]
    L = hp['n_layers']
    t = {}

    # ---- shared stages on a sub-sample, scaled ---------------------------------------------
    n_patches = hp['max_sim_epochs'] * hp['n_anchor_patches_structure'] * L
    np_s = n_patches                                  # round 4: every patch (dealt to the pool), nothing scaled
    t0 = time.perf_counter()
    patches = np.zeros((np_s, hp['sample_walk_len']), dtype=np.int64)
    for i, w in enumerate(_pmap(_w_patch, [(i, hp['sample_walk_len'], hp['rw_beta'], seed) for i in range(np_s)])):
        patches[i, :len(w)] = w
    both = _pmap(_w_walks, [(patches[i], hp['n_triangular_walks'], hp['random_walk_len'], hp['rw_beta'], seed) for i in range(np_s)])
    iw, bw = np.stack([b[0] for b in both]), np.stack([b[1] for b in both])
    t['shared_patches_walks'] = time.perf_counter() - t0
    pext = IH.position_anchors_border(G, hp['n_anchor_patches_pos_out'], seed, 0)
    allp = np.concatenate([patches] * (n_patches // np_s + 1))[:n_patches]
    iw = np.concatenate([iw] * (n_patches // np_s + 1))[:n_patches]
    bw = np.concatenate([bw] * (n_patches // np_s + 1))[:n_patches]
    pp, pf = cbind.ragged([[int(v) for v in row if v] for row in allp])
    t0 = time.perf_counter()
    pi, pe = cbind.degree_sequence(rowptr, col, None, pp, pf, True)
    t['shared_patch_degree_seq'] = time.perf_counter() - t0

    # ---- per-subgraph stages on the sample -------------------------------------------------
    t0 = time.perf_counter()
    ccs = _pmap(_w_components, sample)
    cc_ids = IH.pad_cc_ids(ccs)
    t['components'] = time.perf_counter() - t0
    S, C, Lc = cc_ids.shape
    rows = cc_ids.reshape(S * C, Lc)
    t0 = time.perf_counter()
    borders = _pmap(_w_border, list(rows))
    t['border'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    A_in, A_out = hp['n_anchor_patches_N_in'], hp['n_anchor_patches_N_out']
    maxb = max(len(b) for b in borders)
    n_int = np.stack(_pmap(_w_anchors, [(r, rows[r][rows[r] != 0], A_in, T.STREAM_N_INT, Lc, seed) for r in range(S * C)]))
    n_bor = np.stack(_pmap(_w_anchors, [(r, borders[r], A_out, T.STREAM_N_BOR, maxb, seed) for r in range(S * C)]))
    p_int = IH.position_anchors_internal(sample, hp['n_anchor_patches_pos_in'], seed, 'train', 0)
    t['anchors'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    cp, cf = cbind.ragged([[int(v) for v in r if v] for r in rows])
    ci, ce = cbind.degree_sequence(rowptr, col, None, cp, cf, True)
    t['degree_seq'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    int_sim = cbind.fastdtw_sim(cp, ci, pp, pi).reshape(S, C, -1)
    bor_sim = cbind.fastdtw_sim(cp, ce, pp, pe).reshape(S, C, -1)
    t['dtw'] = time.perf_counter() - t0
    # position channel: one BFS per P-border anchor over the whole graph (C, the sources shared among the cores) and the
    # min over every sampled component's members.  The BFS part does not depend on the number of subgraphs: a shared stage,
    # run in full; the per-component minimum is part of the same call and small beside it.
    t0 = time.perf_counter()
    p_out = cbind.bfs_min_hops_to_sets(rowptr, col, pext, cp, cf)
    t['shared_pext_bfs_all_sources'] = time.perf_counter() - t0

    # ---- float half: forward + backward + Adam on the sample ---------------------------------
    D = emb.shape[1]
    g = torch.Generator().manual_seed(0)
    E = torch.cat([torch.zeros(1, D), emb.float()], 0).requires_grad_(True)
    params = {'node_embeddings.weight': E}

    def lin(name, o, i):
        params[name + '.weight'] = (torch.randn(o, i, generator=g) / i ** 0.5).requires_grad_(True)
        params[name + '.bias'] = torch.zeros(o, requires_grad=True)
    for mod in ('neighborhood_mpns', 'position_mpns', 'structure_mpns'):
        for side in ('internal', 'border'):
            lin('%s.0.%s.linear' % (mod, side), D, 2 * D)
            lin('%s.0.%s.linear_position' % (mod, side), 1, D)
    for sfx in ('', '_reverse'):
        params['lstm.lstm.weight_ih_l0' + sfx] = (torch.randn(4 * D, D, generator=g) / D ** 0.5).requires_grad_(True)
        params['lstm.lstm.weight_hh_l0' + sfx] = (torch.randn(4 * D, D, generator=g) / D ** 0.5).requires_grad_(True)
        params['lstm.lstm.bias_ih_l0' + sfx] = torch.zeros(4 * D, requires_grad=True)
        params['lstm.lstm.bias_hh_l0' + sfx] = torch.zeros(4 * D, requires_grad=True)
    lin('lstm.linear', D, 2 * D)
    hid = D + 2 * D + hp['n_anchor_patches_pos_in'] + hp['n_anchor_patches_pos_out'] + 2 * hp['n_anchor_patches_structure']
    lin('lin', hp['linear_hidden_dim_1'], hid)
    lin('lin2', hp['linear_hidden_dim_2'], hp['linear_hidden_dim_1'])
    lin('lin3', 3, hp['linear_hidden_dim_2'])
    opt = torch.optim.Adam(list(params.values()), lr=hp['learning_rate'])
    idx = IH.structure_anchor_indices(n_patches, hp['n_anchor_patches_structure'], seed, 0)
    Tt = torch.from_numpy
    anchors = {'N_int': {'train': {0: Tt(n_int).view(S, C, -1)}}, 'N_bor': {'train': {0: Tt(n_bor).view(S, C, -1)}},
               'P_int': {'train': {0: Tt(p_int)}}, 'P_ext': {0: Tt(pext)},
               'S': {0: (Tt(allp[idx]), idx, Tt(iw[idx]), Tt(bw[idx]))}}
    real = Tt((cc_ids[:, :, 0] != 0))
    sims = {('N', 'in', 0): torch.zeros(S, C, A_in), ('N', 'out', 0): (Tt(n_bor).view(S, C, -1) != 0).float(),
            ('P', 'in', 0): torch.zeros(S, C, hp['n_anchor_patches_pos_in']),
            ('P', 'out', 0): Tt(p_out).float().view(S, C, -1) * real.unsqueeze(-1)}
    batch = {'cc_ids': Tt(cc_ids), 'subgraph_idx': torch.arange(S).view(-1, 1), 'NP_sim': sims,
             'I_S_sim': Tt(int_sim), 'B_S_sim': Tt(bor_sim)}
    lab = labels[:S].long()
    hp_f = dict(hp)
    hp_f['lstm_n_layers'] = 1
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        logits = FH.forward(params, hp_f, 'train', batch, anchors, None)
        loss = torch.nn.functional.cross_entropy(logits, lab)
        opt.zero_grad()
        loss.backward()
        opt.step()
    t_fb = (time.perf_counter() - t0) / steps
    # the dense (N+1, D) embedding gradient + Adam update is a per-STEP cost, not per subgraph:
    # time it alone and count it once per pass
    t0 = time.perf_counter()
    E.grad = torch.zeros_like(E)
    torch.optim.Adam([E], lr=1e-3).step()
    t_table = time.perf_counter() - t0
    t['fwd_bwd_adam'] = max(t_fb - t_table, 0.0)

    per_sample = sum(v for k, v in t.items() if not k.startswith('shared_'))
    shared = sum(v for k, v in t.items() if k.startswith('shared_')) + t_table
    est_pass = shared + per_sample * S_total / n_sample
    workers = _POOL[1] if _POOL is not None else 1
    res = {'value': S_total / est_pass, 'unit': 'subgraphs/s', 'cores': int(max(torch.get_num_threads(), workers)),
           'kind': 'port',
           'sample': ('oracle (C + numpy + torch-CPU, same sparse algorithm) on %d of %d subgraphs; per-subgraph stages '
                      'scaled x%.0f; shared stages counted once (all %d structure patches and their walks; one BFS '
                      'per P-border anchor for all %d sources; dense embedding-table Adam); DTW and BFS in C with OpenMP over '
                      'the cores, the Python / numpy integer stages on %d worker processes, torch stages on %d threads; %.1f s '
                      'of CPU work measured'
                      % (n_sample, S_total, S_total / n_sample, n_patches, len(pext), workers, torch.get_num_threads(),
                         time.perf_counter() - t_all)),
           'stage_seconds_measured': {k: round(v, 3) for k, v in t.items()},
           'estimated_full_pass_s': round(est_pass, 1)}
    cal = _calibration()
    if cal is not None:
        c = cal['all_density']
        res['calibration'] = {
            'reference_ms_per_step': c['reference_ms_per_step'], 'port_ms_per_step': c['port_ms_per_step'], 'ratio': c['ratio'],
            'reference_prepare_data_s': c['reference_prepare_data_s'], 'port_prepare_data_s': c['port_prepare_data_s'],
            'prepare_ratio': c['prepare_ratio'], 'cores': cal['cores'],
            'n_density_channel_only': {k: cal['n_density'][k] for k in ('reference_ms_per_step', 'port_ms_per_step', 'ratio')},
            'what': 'imported reference vs this port, same inputs (configs[0] dataset, all three channels on; neighbourhood only '
                    'beside it), build container, 8 cores: %s (tools/bench_density_n.py)' % cal['_file'],
            'taken': cal.get('taken', {'round': 4}),
            'value_calibrated_to_reference': S_total / est_pass / c['ratio'],
            'note': 'value / ratio (the training-step ratio; the reference\'s prepare_data is a further %sx slower than the '
                    'port\'s, with its pure-Python fastdtw stand-in) -- an ESTIMATE of the reference\'s own CPU path at a size '
                    'where its dense N x N structures cannot be built' % c['prepare_ratio']}
    return res
