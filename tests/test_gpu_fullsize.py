"""-m gpu: the kernels at BASELINE.json's full size (1M-node / 10M-edge base graph, 50k subgraphs of 20
nodes -- the oracle cannot replay that in seconds), checked through size-independent properties and
against the oracle on samples: checksums of degree sequences, sortedness, idempotence, exact border
sizes and tape-ranked anchors on sampled rows, BFS levels against scipy, DTW against the C oracle on
sampled pairs and under row reordering / de-duplication, walks that only ever follow edges."""
import numpy as np
import pytest
import torch

from oracle import tape as T, cbind

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
N, M, S, K = 1_000_000, 10, 50_000, 20


@pytest.fixture(scope='module')
def full():
    from subgnn_amd import ops, synthetic
    edges = synthetic.barabasi_albert_edges(N, M, seed=42)
    rowptr, col = synthetic.sorted_csr(edges, N)
    subs = synthetic.bfs_subgraphs(rowptr, col, S, K, seed=1000)
    g = ops.DeviceGraph(rowptr, col, np.arange(1, N + 1, dtype=np.int32), DEV)
    sets = ops.Ragged.from_lists(subs, DEV)
    return dict(ops=ops, rowptr=rowptr, col=col, subs=subs, g=g, sets=sets)


def test_degree_sequences_checksums_sorted_idempotent(full):
    ops, g, sets, rowptr = full['ops'], full['g'], full['sets'], full['rowptr']
    oi, oe = ops.degree_sequence(g, sets, sort=True, use_degree_dict=False)
    oi2, oe2 = ops.degree_sequence(g, sets, sort=True, use_degree_dict=False)
    assert torch.equal(oi, oi2) and torch.equal(oe, oe2)                          # idempotent (no atomics races)
    i2 = oi.view(S, K).long()
    e2 = oe.view(S, K).long()
    assert bool((i2[:, 1:] >= i2[:, :-1]).all()) and bool((e2[:, 1:] >= e2[:, :-1]).all())     # ascending per set
    deg = torch.from_numpy(np.diff(rowptr)).to(DEV)
    full_sum = deg[sets.nodes[:S * K].long()].view(S, K).sum(1)
    assert torch.equal(i2.sum(1) + e2.sum(1), full_sum)                           # internal + external = degree
    assert bool((i2.sum(1) % 2 == 0).all())                                       # every internal edge counted twice
    assert bool((i2.sum(1) >= 2 * (K - 1)).all())                                 # BFS subgraphs are connected
    # unsorted form is a permutation of the sorted one
    ui, ue = ops.degree_sequence(g, sets, sort=False, use_degree_dict=False)
    assert torch.equal(torch.sort(ui.view(S, K), dim=1).values, oi.view(S, K))
    # sampled sets against the C oracle
    idx = np.random.default_rng(0).choice(S, 300, replace=False)
    ptr, flat = cbind.ragged([full['subs'][i] for i in idx])
    ci, ce = cbind.degree_sequence(rowptr, full['col'], None, ptr, flat, True)
    assert np.array_equal(i2[idx].cpu().numpy().reshape(-1), ci) and np.array_equal(e2[idx].cpu().numpy().reshape(-1), ce)


def test_components_of_connected_subgraphs(full):
    ops, g, sets = full['ops'], full['g'], full['sets']
    from subgnn_amd.subgraph_utils import components_from_labels
    cc = components_from_labels(sets.ptr, sets.nodes, ops.cc_labels(g, sets), sets.max_len)
    assert tuple(cc.shape) == (S, 1, K)                                           # one component each
    assert torch.equal(cc.view(S, K), sets.nodes[:S * K].view(S, K).long())       # in subgraph order


def test_border_sizes_and_anchor_ranks_on_samples(full):
    ops, g, sets, rowptr, col = full['ops'], full['g'], full['sets'], full['rowptr'], full['col']
    A, seed, st = 43, 7, T.stream_id(T.STREAM_N_BOR, 'train', 0)
    anchors, sims, counts = ops.khop_border_sample(g, sets, 1, A, seed, st)
    a2, s2, c2 = ops.khop_border_sample(g, sets, 1, A, seed, st)
    assert torch.equal(anchors, a2) and torch.equal(counts, c2)                   # idempotent, workspace left clean
    an, cn, mx = anchors.cpu().numpy(), counts.cpu().numpy(), int(counts.max())
    assert bool(((sims == 1) == (anchors != 0)).all())                            # 1-hop border: hop level 1
    for i in np.random.default_rng(1).choice(S, 60, replace=False):
        members = np.asarray(full['subs'][i])
        border = np.setdiff1d(np.unique(np.concatenate([col[rowptr[v]:rowptr[v + 1]] for v in members])), members)
        assert cn[i] == len(border)
        for a in range(A):
            kk = T.nanchor_pick(seed, st, int(i) * A + a, len(border), len(border) < mx)
            assert an[i, a] == (0 if kk < 0 else border[kk])


def test_bfs_levels_against_scipy(full):
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order, shortest_path
    ops, g, rowptr, col = full['ops'], full['g'], full['rowptr'], full['col']
    src = np.random.default_rng(2).integers(1, N + 1, 183).astype(np.int32)
    dist = ops.bfs_hops(g, torch.from_numpy(src).to(DEV), max_hops=32)
    assert int(dist[torch.arange(183), torch.from_numpy(src).long()].max()) == 0  # a source is at distance 0
    d = dist[:, 1:]
    assert int(d.max()) < 255                                                     # BA graph: connected
    A = sp.csr_matrix((np.ones(len(col), dtype=np.int8), col.astype(np.int64) - 1, rowptr[1:] - rowptr[1]), shape=(N, N))
    ref = shortest_path(A, method='D', unweighted=True, indices=src[:3].astype(np.int64) - 1)
    assert np.array_equal(d[:3].cpu().numpy().astype(np.float64), ref)
    # fused min over members == two-step form on the full 50k sets
    two = ops.min_hops_to_sets(dist, full['sets'])
    assert torch.equal(ops.bfs_min_hops_to_sets(g, torch.from_numpy(src).to(DEV), full['sets'], max_hops=32), two)
    # the hinted form the pass runs (levels and push levels capped from an earlier search's status): same rows
    w, st = ops.bfs_min_hops_to_sets(g, torch.from_numpy(src).to(DEV), full['sets'], max_hops=32, want_status=True)
    last, more, first_pull, _ = st.tolist()
    assert more == 0 and first_pull >= 2 and torch.equal(w, two)
    for cap in (1, first_pull - 1, first_pull):
        w2, st2 = ops.bfs_min_hops_to_sets(g, torch.from_numpy(src).to(DEV), full['sets'], max_hops=last + 3, want_status=True,
                                           push_levels=cap)
        assert torch.equal(w2, two) and st2.tolist()[:2] == [last, 0], cap


def test_walks_follow_edges_and_dtw_properties(full):
    ops, g, sets, rowptr, col = full['ops'], full['g'], full['sets'], full['rowptr'], full['col']
    walks = ops.triangular_walks(g, 0, 210, 50, 0.65, 0, T.stream_id(T.STREAM_STRUCT_PATCH))
    w = walks.cpu().numpy()
    for row in w:
        nz = row[row != 0]
        for a, b in zip(nz[:-1], nz[1:]):
            nb = col[rowptr[a]:rowptr[a + 1]]
            assert nb[np.searchsorted(nb, b)] == b                                 # consecutive nodes are adjacent
    a_sets = ops.Ragged.from_padded(walks)
    ai, ae = ops.degree_sequence(g, a_sets)
    ci, ce = ops.degree_sequence(g, sets)
    sim = ops.dtw_similarity(sets.ptr, ce, K, a_sets.ptr, ae, 50)
    assert tuple(sim.shape) == (S, 210) and bool((sim > 0).all()) and bool((sim <= 1).all())
    # a series against itself: distance 0, similarity exactly 1
    self_sim = ops.dtw_similarity(a_sets.ptr, ae, 50, a_sets.ptr, ae, 50, dedupe=False)
    assert bool((torch.diagonal(self_sim) == 1).all())
    # processing order and de-duplication change nothing: a 3000-row slice recomputed plainly
    sub = ops.Ragged(sets.ptr[:3001].clone(), ce[:3000 * K].clone(), max_len=K)
    plain = ops.dtw_similarity(sub.ptr, sub.nodes, K, a_sets.ptr, ae, 50, order_rows=False, dedupe=False)
    assert torch.equal(plain, sim[:3000])
    int_sim = ops.dtw_similarity(sets.ptr, ci, K, a_sets.ptr, ai, 50)               # de-duplicated side
    plain_i = ops.dtw_similarity(sub.ptr, ci[:3000 * K].clone(), K, a_sets.ptr, ai, 50, order_rows=False, dedupe=False)
    assert torch.equal(plain_i, int_sim[:3000])
    # sampled pairs against the C oracle (fastdtw restatement)
    rng = np.random.default_rng(3)
    rows, cols_ = rng.choice(S, 40, replace=False), rng.choice(210, 5, replace=False)
    xs = ce.view(S, K)[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    yl, yv = a_sets.to_lists(), ae.cpu().numpy()
    yp = a_sets.ptr.cpu().numpy()
    ys = [yv[yp[c]:yp[c + 1]] for c in cols_]
    xp, xf = cbind.ragged([list(map(int, x)) for x in xs])
    ypp, yf = cbind.ragged([list(map(int, y)) for y in ys])
    ref = cbind.fastdtw_sim(xp, xf, ypp, yf).reshape(40, 5)          # (both sides: the default predecessor rule)
    got = sim[torch.from_numpy(rows).to(DEV)][:, torch.from_numpy(cols_).to(DEV)].cpu().numpy()
    assert np.array_equal(got, ref)


def test_one_shard_of_the_benchmark_reproduces_its_slice_of_the_unsharded_pass(full):
    """BASELINE configs[3] as worded ("50k subgraphs, sharded across 8 x MI355X"): ONE of the eight 6 250-subgraph shards
    of the 1M-node / 50k-subgraph workload (rank 3), prepared on its own with dist.Shard -- draws keyed by the global
    subgraph numbers, padded widths as the all-reduce over the eight ranks would give them -- against the same slice of
    the unsharded pass: component tensors, every anchor tensor, every similarity, bit for bit."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from bench import ALL_DENSITY_HP
    from subgnn_amd import hotpath, ops
    from subgnn_amd import dist as sdist
    from subgnn_amd.SubGNN import SubGNN
    g, subs = full['g'], full['subs']
    hp = dict(ALL_DENSITY_HP)
    emb = torch.randn(N, hp['node_embed_size'], generator=torch.Generator().manual_seed(0)).to(DEV)
    labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
    labels[:3] = torch.tensor([0, 1, 2])

    def model(sub_lists, lab):
        torch.manual_seed(0)
        return SubGNN.from_memory(dict(hp), g, {'train': sub_lists, 'val': [], 'test': []},
                                  {'train': lab, 'val': lab[:0], 'test': lab[:0]}, emb, num_classes=3)
    whole = model(subs, labels)
    hotpath.prepare_sparse(whole, 'train')
    Sx, C, L = whole.train_cc_ids.shape
    assert (Sx, C) == (S, 1)
    width = ops.khop_border(g, ops.Ragged.from_padded(whole.train_cc_ids.reshape(S * C, L)), 1).lengths.max().view(1)
    dims = torch.tensor([C, L], device=DEV)

    class PlayedShard(sdist.Shard):                     # the MAX over ranks, known here from the unsharded pass
        def reduce_max(self, t):
            return torch.maximum(t, (dims if t.numel() == 2 else width).to(t.dtype))
    sh = PlayedShard(S, 3, 8, collectives=False)
    a, b = sh.start, sh.stop
    assert b - a == 6250
    part = model(subs[a:b], labels[a:b])
    hotpath.prepare_sparse(part, 'train', shard=sh)
    assert torch.equal(part.train_cc_ids, whole.train_cc_ids[a:b])
    for l in range(hp['n_layers']):
        assert torch.equal(part.anchors_neigh_int['train'][l], whole.anchors_neigh_int['train'][l][a:b])
        assert torch.equal(part.anchors_neigh_border['train'][l], whole.anchors_neigh_border['train'][l][a:b])
        assert torch.equal(part.anchors_pos_int['train'][l], whole.anchors_pos_int['train'][l][a:b])
        assert torch.equal(part.anchors_pos_ext[l], whole.anchors_pos_ext[l])
        for key in (('N', 'out', l), ('P', 'out', l)):
            assert torch.equal(part.train_neigh_pos_similarities[key], whole.train_neigh_pos_similarities[key][a:b]), key
    assert torch.equal(part.structure_anchors, whole.structure_anchors)
    assert torch.equal(part.train_int_struc_similarities, whole.train_int_struc_similarities[a:b])
    assert torch.equal(part.train_bor_struc_similarities, whole.train_bor_struc_similarities[a:b])
    assert float(part.train_bor_struc_similarities.abs().sum()) > 0
    # ... and the same rank in the DEALT (strong-scaling) form, as ONE process with the peers' shares from recorded buffers
    # (dist.EmulatedPeers: what bench.py's strong_rank8 object runs): the rank walks an eighth of the structure patches and of
    # their walks, searches ITS 23 of the 183 position sources over all ranks' components, and still arrives at its slice of
    # the unsharded pass, bit for bit -- on the recording pass AND on the passes that replay the recorded buffers
    emu = sdist.EmulatedPeers(3, 8)
    emu.provided['cc_ids_all'] = whole.train_cc_ids.reshape(S * C, L)
    emu.maxima[2], emu.maxima[1] = dims.to(torch.int32), width
    dealt_shard = sdist.Shard(S, 3, 8, deal_shared=True, emulator=emu)
    assert dealt_shard.deal_shared and not dealt_shard.collectives
    dealt = model(subs[a:b], labels[a:b])
    for _pass in range(2):
        hotpath.prepare_sparse(dealt, 'train', shard=dealt_shard)
        assert torch.equal(dealt.train_cc_ids, whole.train_cc_ids[a:b])
        assert torch.equal(dealt.structure_anchors, whole.structure_anchors)
        assert torch.equal(dealt.int_structure_anchor_random_walks, whole.int_structure_anchor_random_walks)
        assert torch.equal(dealt.bor_structure_anchor_random_walks, whole.bor_structure_anchor_random_walks)
        for l in range(hp['n_layers']):
            assert torch.equal(dealt.anchors_pos_ext[l], whole.anchors_pos_ext[l])
            for key in (('N', 'out', l), ('P', 'out', l)):
                assert torch.equal(dealt.train_neigh_pos_similarities[key], whole.train_neigh_pos_similarities[key][a:b]), key
        assert torch.equal(dealt.train_int_struc_similarities, whole.train_int_struc_similarities[a:b])
        assert torch.equal(dealt.train_bor_struc_similarities, whole.train_bor_struc_similarities[a:b])
    assert set(k[0] if isinstance(k, tuple) else k for k in emu.received_bytes) >= {'S_patches', 'S_walks', 'cc_ids', 'P_out'}
    assert 0 < sum(emu.received_bytes.values()) < 64 << 20


@pytest.mark.parametrize('n_layers', [1, 2])
def test_training_half_at_shard_size_matches_the_oracle(full, n_layers):
    """The TRAINING half the benchmark times (SubGNN.py:225-348 forward + loss, :1156-1164 backward + clip + Adam) on the
    configs[3] workload at shard size: the 1M-node graph, ALL_DENSITY_HP, 8 192 subgraphs as one batch -- so that every
    shard-size branch runs (counted below): the shared-anchor layers as library GEMMs (ops._mpn_shared_gemm, rows >=
    SHARED_GEMM_MIN_ROWS), the head's tall linears (_LinearTallSkinny, rows >= 8192) with their split contractions
    (contract_rows) and column sums, the slot-fused read-out (ReadoutPiece / sgnn_readout_sum_*), the fused cross entropy,
    the edge plans and sorted id lists made with the prepared pass, the MFMA update layer, ClipAdam's two-launch tail
    (norm of every gradient; coefficient + Adam on every parameter, the table included).  n_layers = 1 is the benchmark's configuration (every shared-anchor layer is the channel's last one:
    read-out pieces only); n_layers = 2 adds a first layer whose shared-anchor bodies are the library contractions
    (ops._mpn_shared_gemm, rows >= SHARED_GEMM_MIN_ROWS) and whose update feeds a second layer.  Dropout 0.  Logits, loss, every gradient and every parameter after one clip + Adam step against
    oracle/float_half.py + clip_grad_norm_ + torch.optim.Adam fed the product's prepared state: element-wise 1e-4."""
    import sys, os, collections
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from bench import ALL_DENSITY_HP
    from helpers import assert_close, rel_err, oracle_inputs
    from oracle import float_half as FH
    from subgnn_amd import hotpath, ops, optim, _lib
    from subgnn_amd.SubGNN import SubGNN
    g, subs = full['g'], full['subs']
    B = 8192
    hp = dict(ALL_DENSITY_HP, lin_dropout=0.0, lstm_dropout=0.0, n_layers=n_layers)
    D = hp['node_embed_size']
    emb = torch.randn(N, D, generator=torch.Generator().manual_seed(0)).to(DEV)
    labels = torch.randint(0, 3, (B,), generator=torch.Generator().manual_seed(0))
    torch.manual_seed(0)
    m = SubGNN.from_memory(dict(hp), g, {'train': subs[:B], 'val': [], 'test': []},
                           {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3)
    m.train()
    lib = _lib.load()
    calls = collections.Counter()
    py_names = ('_mpn_shared_gemm', 'contract_rows', 'column_sum', 'mpn_edge_plan', 'presort_ids', 'update_layer',
                'cross_entropy_with_accuracy', 'subgraph_embedding')
    lib_names = ('sgnn_readout_sum_fwd', 'sgnn_readout_sum_bwd', 'sgnn_update_fwd', 'sgnn_update_bwd', 'sgnn_scatter_add_rows',
                 'sgnn_cross_entropy_fwd', 'sgnn_cross_entropy_bwd', 'sgnn_optim_sumsq', 'sgnn_optim_adam', 'sgnn_lstm_fwd', 'sgnn_lstm_bwd',
                 'sgnn_cc_embed_fwd', 'sgnn_mpn_fwd', 'sgnn_mpn_bwd', 'sgnn_head_fwd', 'sgnn_head_bwd', 'sgnn_contract_rows_partial',
                 'sgnn_reduce_partials', 'sgnn_readout_many_fwd', 'sgnn_readout_many_bwd')
    saved_py = {n: getattr(ops, n) for n in py_names}
    saved_lib = {n: getattr(lib, n) for n in lib_names if hasattr(lib, n)}
    tall = collections.Counter()
    real_tall = ops._LinearTallSkinny.apply

    def counted(name, fn):
        def w(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return w
    real_linear = ops.linear

    def linear(x, weight, bias):
        if x.dim() == 2 and x.shape[0] >= 8192:
            tall['rows>=8192'] += 1
        return real_linear(x, weight, bias)
    for n, f in saved_py.items():
        setattr(ops, n, counted(n, f))
    for n, f in saved_lib.items():
        setattr(lib, n, counted(n, f))
    ops.linear = linear
    try:
        hotpath.prepare_sparse(m, 'train')
        S_, C, Lc = m.train_cc_ids.shape
        assert (S_, C) == (B, 1) and B * C >= ops.SHARED_GEMM_MIN_ROWS
        opt = optim.ClipAdam(m.parameters(), hp['learning_rate'], max_norm=hp['grad_clip'])
        assert [tuple(p.shape) for p in opt.big] == [(N + 1, D)]
        before = {k: v.detach().clone() for k, v in m.state_dict().items() if v.dtype == torch.float32}
        batch = hotpath.full_split_batch(m, 'train')
        out = m.training_step(batch, 0)
        with torch.no_grad():
            logits = m._forward_batch('train', batch)
        m.backward(None, out['loss'], None, 0)
        # no two parameters' gradients may share memory (an in-place multi-tensor update -- clip_grad_norm_ -- would hit it twice)
        spans = sorted((p.grad.data_ptr(), p.grad.data_ptr() + p.grad.numel() * p.grad.element_size(), k)
                       for k, p in m.named_parameters() if p.grad is not None)
        assert all(a[1] <= b[0] for a, b in zip(spans[:-1], spans[1:])), [(a[2], b[2]) for a, b in zip(spans[:-1], spans[1:]) if a[1] > b[0]]
        grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        opt.step()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
    finally:
        for n, f in saved_py.items():
            setattr(ops, n, f)
        for n, f in saved_lib.items():
            setattr(lib, n, f)
        ops.linear = real_linear
    # ---- the shard-size branches ran -------------------------------------------------------------------------------
    assert calls['_mpn_shared_gemm'] >= (3 if n_layers > 1 else 0), calls      # first layer: P-border + both structure sides
    # the head + loss: one fused launch each way behind the first layer's GEMM (csrc/head.hip: two calls forward -- the training
    # step and the no-grad forward above --, one backward), its first weight gradient on the matrix cores, one reduction launch
    assert calls['sgnn_head_fwd'] == 2 and calls['sgnn_head_bwd'] == 1 and tall['rows>=8192'] == 0, (calls, tall)
    assert calls['sgnn_contract_rows_partial'] >= 1 and calls['sgnn_reduce_partials'] >= 1, calls
    assert calls['mpn_edge_plan'] >= 1 and calls['presort_ids'] >= 2, calls
    assert calls['cross_entropy_with_accuracy'] == 0 and calls['sgnn_cross_entropy_bwd'] == 0, calls      # (inside the head's launches)
    # the read-out pieces (P internal / border, S internal / border of the last layer): all of them in one call each way
    assert calls['subgraph_embedding'] >= 1 and calls['sgnn_readout_many_fwd'] == 2 and calls['sgnn_readout_many_bwd'] == 1, calls
    assert calls['sgnn_readout_sum_fwd'] == 0 and calls['sgnn_readout_sum_bwd'] == 0, calls
    assert calls['sgnn_update_fwd'] >= 1 and calls['sgnn_update_bwd'] >= 1, calls
    assert calls['sgnn_optim_sumsq'] == 1 and calls['sgnn_optim_adam'] == 1, calls      # the whole optimizer tail: two library calls
    assert calls['sgnn_lstm_fwd'] >= 1 and calls['sgnn_lstm_bwd'] >= 1, calls
    # ---- the oracle on the product's prepared state ----------------------------------------------------------------
    params, anchors, ob, ccp = oracle_inputs(m, batch, torch.arange(B))
    # the oracle in FLOAT64 on the state BEFORE the step (state_dict() above is after it): a float32 restatement carries
    # its own summation error on the small elements of a gradient contracted over 8192 x 183 terms
    for k in list(params):
        if params[k].dtype == torch.float32:
            params[k] = (before[k].cpu() if k in before else params[k].detach()).double().requires_grad_(True)
    ob = {k: ({kk: (vv.double() if torch.is_tensor(vv) and vv.dtype == torch.float32 else vv) for kk, vv in v.items()} if isinstance(v, dict)
              else (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v)) for k, v in ob.items()}
    ref_logits = FH.forward(params, hp, 'train', ob, anchors, ccp)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, labels)
    ref_loss.backward()
    assert_close(logits, ref_logits, 'logits at shard size')
    assert_close(out['loss'], ref_loss, 'loss at shard size')
    checked = 0
    for k, gp in grads.items():
        ref = params[k].grad
        if gp is None or ref is None:
            other = ref if gp is None else gp
            assert other is None or float(other.abs().max()) == 0, k
            continue
        assert_close(gp, ref, 'grad ' + k)
        checked += 1
    assert checked >= 12, checked
    assert float(grads['node_embeddings.weight'][0].abs().max()) == 0
    # clip_grad_norm_ + Adam as the caller runs them (train_config.py Trainer(gradient_clip_val); SubGNN.py:1156-1161)
    leaves = [v for k, v in params.items() if v.requires_grad and v.grad is not None]
    torch.nn.utils.clip_grad_norm_(leaves, hp['grad_clip'])
    torch.optim.Adam(leaves, lr=hp['learning_rate']).step()
    after = {k: v.detach() for k, v in m.state_dict().items() if v.dtype == torch.float32}
    # Parameters after the step against the oracle's.  Adam's first step moves an element by lr * g / (|g| + eps): where the
    # gradient is tiny (|g| ~ eps = 1e-8: dead units of the head) the step is sensitive to the 1e-4 the two gradients may
    # differ by -- an element of lin2.weight with g ~ 1e-8 moved by 0.95 lr here and by 1.0 lr in the oracle -- so an element
    # is allowed 1e-4 relative PLUS what the asserted gradient tolerance can move Adam's step: lr * dg / (|g| + eps) (<= 2 lr)
    lr_, eps_ = hp['learning_rate'], 1e-8
    for k, v in params.items():
        if not (v.requires_grad and v.grad is not None):
            continue
        gref = v.grad.double().abs()                                 # (clipped in place above: the gradient Adam saw)
        dg = 1e-4 * gref + 1e-6 * float(gref.max())                  # the element-wise gradient tolerance asserted above
        step_slack = (lr_ * dg / (gref + eps_)).clamp(max=2 * lr_)
        allowed = 1e-4 * v.detach().double().abs() + 1e-6 * float(v.detach().abs().max()) + step_slack
        err = (after[k].cpu().double() - v.detach().double()).abs()
        worst_i = int(torch.argmax(err / allowed))
        assert bool((err <= allowed).all()), ('parameter after clip + Adam: ' + k, float(err.view(-1)[worst_i]), float(allowed.view(-1)[worst_i]))
    # ... and the UPDATE itself (parameters of size ~1 moved by ~lr pass the line above whatever the step did).  Adam's first
    # step is lr * g / (|g| + eps): where |g| ~ eps = 1e-8 it amplifies the 1e-4 the two gradients may differ by (3 % on the
    # worst table element here), so the update is checked against Adam's rule applied in float64 to the PRODUCT's own
    # gradients (whose parity is asserted above): |update - rule| <= 1e-4 |rule| + one float32 rounding of the parameter
    g64 = {k: g.double() for k, g in grads.items() if g is not None}
    total = torch.sqrt(sum((g * g).sum() for g in g64.values()))
    coef = torch.clamp(hp['grad_clip'] / (total + 1e-6), max=1.0)
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, hp['learning_rate']
    worst = {}
    for k, g in g64.items():
        gc = g * coef
        rule = -lr * ((1 - b1) * gc / (1 - b1)) / (torch.sqrt((1 - b2) * gc * gc / (1 - b2)) + eps)
        got = after[k].double() - before[k].double()
        slack = 1e-4 * rule.abs() + 2 * 2.0 ** -24 * before[k].double().abs() + 1e-12
        worst[k] = float(((got - rule).abs() / slack).max())
        assert worst[k] <= 1.0, (k, worst[k])
    untouched = (grads['node_embeddings.weight'].abs().sum(1) == 0)
    assert torch.equal(after['node_embeddings.weight'][untouched], before['node_embeddings.weight'][untouched])
    print('shard-size training half: calls', dict(calls), 'worst update error / allowed', {k: '%.2f' % e for k, e in worst.items()})


def test_one_50k_batch_equals_its_chunks(full):
    """The float half on ALL 50 000 subgraphs as ONE batch -- the launch shape bench.py times (hotpath.full_split_batch) --
    against the same model run over seven chunks of <= 8 192 subgraphs (SubGNN.make_batch; the last chunk has 848 rows and
    takes the small-R kernels: K-split update layer, un-split contractions, short scatter runs).  The forward is row-wise in
    the subgraph (SubGNN.py:225-348 has no cross-subgraph term without batch norm) and the loss is a mean, so
        logits(full)[chunk] = logits(chunk)      and      grad(full) = sum_chunks  n_chunk / S * grad(chunk)
    hold exactly in real arithmetic: a size-independent property that covers the 50k launch itself, which the oracle cannot
    replay in seconds (the test above pins the 8 192-row shape to the float64 oracle).  Tolerances: logits 1e-5 of the
    largest logit; gradients element-wise 1e-4 relative + 1e-6 of the tensor's largest element (tests/helpers.assert_close)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from bench import ALL_DENSITY_HP
    from helpers import assert_close
    from subgnn_amd import hotpath
    from subgnn_amd.SubGNN import SubGNN
    g, subs = full['g'], full['subs']
    hp = dict(ALL_DENSITY_HP, lin_dropout=0.0, lstm_dropout=0.0)
    D = hp['node_embed_size']
    emb = torch.randn(N, D, generator=torch.Generator().manual_seed(1)).to(DEV)
    labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(1))
    torch.manual_seed(1)
    m = SubGNN.from_memory(dict(hp), g, {'train': subs, 'val': [], 'test': []},
                           {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3)
    m.train()
    hotpath.prepare_sparse(m, 'train')
    assert m.train_cc_ids.shape[0] == S
    batch = hotpath.full_split_batch(m, 'train')
    out = m.training_step(batch, 0)
    with torch.no_grad():
        logits = m._forward_batch('train', batch)
    m.backward(None, out['loss'], None, 0)
    torch.cuda.synchronize()
    full_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    full_loss = float(out['loss'].detach())
    for p in m.parameters():
        p.grad = None
    CH = 8192
    pieces, loss_sum = [], 0.0
    for lo in range(0, S, CH):
        idx = torch.arange(lo, min(S, lo + CH))
        b = m.make_batch('train', idx)
        o = m.training_step(b, 0)
        with torch.no_grad():
            pieces.append(m._forward_batch('train', b))
        w = idx.numel() / S
        m.backward(None, o['loss'] * w, None, 0)
        loss_sum += float(o['loss'].detach()) * w
    torch.cuda.synchronize()
    assert len(pieces) == 7 and pieces[-1].shape[0] == S - 6 * CH
    chunked = torch.cat(pieces)
    scale = float(logits.abs().max())
    assert float((logits - chunked).abs().max()) <= 1e-5 * scale, (float((logits - chunked).abs().max()), scale)
    assert abs(full_loss - loss_sum) <= 1e-5 * abs(full_loss), (full_loss, loss_sum)
    checked = 0
    for k, p in m.named_parameters():
        if k not in full_grads:
            assert p.grad is None or float(p.grad.abs().max()) == 0, k
            continue
        assert p.grad is not None, k
        assert_close(full_grads[k], p.grad, 'grad ' + k + ' (one batch vs chunks)')
        checked += 1
    assert checked >= 12, checked
    # the table's gradient touches exactly the rows some subgraph, border set or anchor patch reads; row 0 is the padding row
    assert float(full_grads['node_embeddings.weight'][0].abs().max()) == 0
    assert torch.equal(full_grads['node_embeddings.weight'].abs().sum(1) > 0, m.node_embeddings.weight.grad.abs().sum(1) > 0)
