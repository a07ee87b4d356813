"""The integer-half oracle against the golden vectors produced by the imported reference
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import graph as OG, integer_half as IH, tape as T, fastdtw_restate as FD


def _graph(g):
    return OG.from_edge_pairs([tuple(e) for e in g['edge_list']])


def _degdict(G):
    return {v - 1: G.degree(v) for v in G.node_order}


def test_g1_graph_order(golden):
    G = _graph(golden)
    rp, col = G.csr()
    assert G.node_order == list(golden['g1_node_order'])
    assert np.array_equal(rp, golden['g1_rowptr'])
    assert np.array_equal(col, golden['g1_col'])


def test_g2_connected_components(golden):
    G = _graph(golden)
    for sp in ('train', 'val'):
        cc, sub = golden['g2_cc_ids_' + sp], golden['subgraphs_' + sp]
        for s in range(cc.shape[0]):
            ref = {frozenset(int(v) for v in row if v != 0) for row in cc[s] if row[0] != 0}
            mine = {frozenset(c) for c in IH.connected_components(G, [v for v in sub[s] if v != 0])}
            assert ref == mine


def test_g3_border_sets(golden):
    G = _graph(golden)
    cc = golden['g2_cc_ids_train']
    for k in (1, 2, 3):
        rows = golden.ragged('g3_border_k%d_train' % k, -1)
        r = 0
        for s in range(cc.shape[0]):
            for c in range(cc.shape[1]):
                assert sorted(IH.component_border_set(G, cc[s, c], k, golden.has_ego)) == rows[r]
                r += 1
    gb = golden['g3_border_train']          # what prepare_data stored (k = 2), PAD-filled
    for s in range(cc.shape[0]):
        for c in range(cc.shape[1]):
            mine = sorted(IH.component_border_set(G, cc[s, c], 2, golden.has_ego))
            assert [m for m in mine if m != 0] == sorted(int(v) for v in gb[s, c] if v != 0)


def test_g4_shortest_path_similarities(golden):
    G = _graph(golden)
    assert np.array_equal(IH.bfs_all_pairs(G), golden['apsp'])
    for sp in ('train', 'val'):
        assert np.array_equal(IH.shortest_path_similarities(golden['apsp'], golden['g2_cc_ids_' + sp]),
                              golden['g4_np_sim_' + sp])


def test_g5_patches_and_walks(golden):
    G = _graph(golden)
    hp, seed = golden.hp, golden.seed
    sa = golden['g5_structure_anchors']
    n = hp['max_sim_epochs'] * hp['n_anchor_patches_structure'] * hp['n_layers']
    assert np.array_equal(IH.sample_structure_anchor_patches(G, n, hp['sample_walk_len'], hp['rw_beta'], seed), sa)
    views = golden.ragged('g5_views_int', 0)
    iw = IH.perform_random_walks(G, sa, hp['n_triangular_walks'], hp['random_walk_len'], hp['rw_beta'], True, seed,
                                 patch_orders=views)
    assert np.array_equal(iw, golden['g5_int_walks'])
    vb, inb = golden.ragged('g5_views_bor', 0), golden.ragged('g5_in_border', 0)
    assert all(IH.patch_in_border_nodes(G, v) == b for v, b in zip(vb, inb))
    bw = IH.perform_random_walks(G, sa, hp['n_triangular_walks'], hp['random_walk_len'], hp['rw_beta'], False, seed,
                                 patch_orders=vb, in_borders=inb)
    assert np.array_equal(bw, golden['g5_bor_walks'])


def test_g6_degree_sequences(golden):
    G = _graph(golden)
    dd = _degdict(G)
    sa = golden['g5_structure_anchors']
    cc = golden['g2_cc_ids_train'].reshape(-1, golden['g2_cc_ids_train'].shape[-1])
    for internal, key in ((True, 'int'), (False, 'ext')):
        ref = golden.ragged('g6_anchor_deg_' + key, -1)
        for a in range(sa.shape[0]):
            assert IH.degree_sequence(G, sa[a], dd, internal) == ref[a]
        ref = golden.ragged('g6_cc_deg_%s_train' % key, -1)
        ref2 = golden.ragged('g6_cc_deg_nodict_%s_train' % key, -1)
        for r in range(cc.shape[0]):
            assert IH.degree_sequence(G, cc[r], dd, internal) == ref[r]
            assert IH.degree_sequence(G, cc[r], None, internal) == ref2[r]


def test_g6_duplicates_are_kept(golden):
    """Walk-sampled patches revisit nodes; the degree sequence keeps one entry per visit."""
    G = _graph(golden)
    sa = golden['g5_structure_anchors']
    lens = [(row != 0).sum() for row in sa]
    uniq = [len(set(int(v) for v in row if v != 0)) for row in sa]
    assert any(l > u for l, u in zip(lens, uniq))
    for a in range(sa.shape[0]):
        assert len(IH.degree_sequence(G, sa[a], None, True)) == lens[a]


def test_g7_structure_similarities(golden):
    """PROVISIONAL: pinned only against the restated fastdtw (parity unpinned)."""
    G = _graph(golden)
    dd = _degdict(G)
    cc, sa = golden['g2_cc_ids_train'], golden['g5_structure_anchors']
    for internal, key in ((True, 'int'), (False, 'bor')):
        assert np.array_equal(IH.structure_similarities(G, cc, sa, dd, internal, tie_order=0), golden['g7_%s_struc_sim_train' % key])


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_g7_structure_similarities_other_tie_rules(name):
    """tests/golden/ties.npz: the same stage boundary under predecessor rules 1 and 2 (the product's default is 2), produced
    by the reference's own compute_structure_patch_similarities through the fastdtw stand-in (make_goldens_ties.py);
    a self-consistency pin like g7.  (On these small fixtures the three rules agree in value: the pins constrain, the
    discriminating cases are the random series of test_fastdtw_tie_orders_bound_exact_dtw / test_dtw_random.)"""
    from conftest import load_golden, GOLDEN_DIR
    import os
    golden = load_golden(name)
    ties = np.load(os.path.join(GOLDEN_DIR, 'ties.npz'))
    G = _graph(golden)
    dd = _degdict(G)
    cc, sa = golden['g2_cc_ids_train'], golden['g5_structure_anchors']
    for tie in (1, 2):
        for internal, key in ((True, 'int'), (False, 'bor')):
            assert np.array_equal(IH.structure_similarities(G, cc, sa, dd, internal, tie_order=tie),
                                  ties['%s/g7_tie%d_%s_struc_sim_train' % (name, tie, key)])
    assert FD.DEFAULT_TIE_ORDER == 2 and np.array_equal(
        IH.structure_similarities(G, cc, sa, dd, True), ties['%s/g7_tie2_int_struc_sim_train' % name])      # the default rule


def test_fastdtw_never_below_exact_dtw():
    rng = np.random.default_rng(0)
    for _ in range(200):
        x = sorted(rng.integers(0, 9, rng.integers(1, 26)).tolist())
        y = sorted(rng.integers(0, 31, rng.integers(1, 51)).tolist())
        for tie in (0, 1):
            d, _ = FD.fastdtw(x, y, dist=FD.calc_dist, tie_order=tie)
            assert d >= FD.exact_dtw(x, y, FD.calc_dist) - 1e-12


def test_g8_anchor_tensors(golden):
    G = _graph(golden)
    hp, seed = golden.hp, golden.seed
    for sp in ('train', 'val'):
        ccs, bs = golden['g2_cc_ids_' + sp], golden['g3_border_' + sp]
        subs = golden.ragged('subgraphs_' + sp, 0)
        for l in range(hp['n_layers']):
            assert np.array_equal(IH.sample_neighborhood_anchors(ccs, hp['n_anchor_patches_N_in'], seed, T.stream_id(T.STREAM_N_INT, sp, l)),
                                  golden['g8_N_int_%s_%d' % (sp, l)])
            assert np.array_equal(IH.sample_neighborhood_anchors(bs, hp['n_anchor_patches_N_out'], seed, T.stream_id(T.STREAM_N_BOR, sp, l)),
                                  golden['g8_N_bor_%s_%d' % (sp, l)])
            assert np.array_equal(IH.position_anchors_internal(subs, hp['n_anchor_patches_pos_in'], seed, sp, l),
                                  golden['g8_P_int_%s_%d' % (sp, l)])
    sa = golden['g5_structure_anchors']
    for l in range(hp['n_layers']):
        assert np.array_equal(IH.position_anchors_border(G, hp['n_anchor_patches_pos_out'], seed, l), golden['g8_P_ext_%d' % l])
        idx = IH.structure_anchor_indices(sa.shape[0], hp['n_anchor_patches_structure'], seed, l)
        assert idx == list(golden['g8_S_idx_%d' % l])
        assert np.array_equal(sa[idx], golden['g8_S_patches_%d' % l])
        assert np.array_equal(golden['g5_int_walks'][idx], golden['g8_S_int_rw_%d' % l])


def test_g8_pad_anchor_quirk(golden):
    """aps:178,190: padded slots hold 0, not -inf, so a short component can yield the PAD anchor."""
    seen_pad = False
    for sp in ('train', 'val'):
        for l in range(golden.hp['n_layers']):
            a = golden['g8_N_int_%s_%d' % (sp, l)]
            real = golden['g2_cc_ids_' + sp][:, :, 0] != 0
            seen_pad |= bool((a[real] == 0).any())
    assert seen_pad or golden.name == 'density'


def test_g12_collate_trim(golden):
    idx = golden['g12_idx']
    assert np.array_equal(IH.trim_zero_columns(golden['g2_cc_ids_train'][idx]), golden['g12_cc_ids'])
    assert np.array_equal(IH.trim_zero_columns(golden['g3_border_train'][idx]), golden['g12_N_border'])
    assert np.array_equal(golden['g4_np_sim_train'][idx], golden['g12_NP_sim'])


def test_tape_scalar_matches_vector():
    item = np.repeat(np.arange(7), 5)
    j = np.tile(np.arange(5), 7)
    st = T.stream_id(T.STREAM_N_BOR, 'val', 3)
    v = T.draw64_np(99, st, item, j)
    for a, b, x in zip(item, j, v):
        assert T.draw64(99, st, int(a), int(b)) == int(x)


def test_neighbourhood_anchor_law():
    """The two-draw law has the distribution of the reference's argmax over iid symmetric variates
    with PAD columns at 0: PAD with probability 2**-n on padded rows, else uniform over the entries."""
    st = T.stream_id(T.STREAM_N_INT, 'train', 0)
    assert T.nanchor_pick(5, st, 0, 0, True) == -1 and T.nanchor_pick(5, st, 0, 0, False) == -1
    for n in (1, 2, 3):
        picks = np.array([T.nanchor_pick(5, st, i, n, True) for i in range(20000)])
        assert abs((picks == -1).mean() - 2.0 ** -n) < 0.015
        for k in range(n):
            assert abs((picks == k).mean() - (1 - 2.0 ** -n) / n) < 0.015
        assert (np.array([T.nanchor_pick(5, st, i, n, False) for i in range(2000)]) >= 0).all()
    big = np.array([T.nanchor_pick(5, st, i, 1000, True) for i in range(5000)])
    assert big.min() >= 0 and big.max() < 1000 and abs(big.mean() - 499.5) < 15
    # the pick is defined on the ascending order of the entries: the oracle ignores the row order
    m = np.array([[[9, 4, 7, 0, 0]], [[3, 0, 0, 0, 0]]])
    a = IH.sample_neighborhood_anchors(m, 6, 11, st)
    b = IH.sample_neighborhood_anchors(m[:, :, [2, 0, 1, 3, 4]], 6, 11, st)
    assert np.array_equal(a, b) and set(a[0, 0]) <= {0, 4, 7, 9} and set(a[1, 0]) <= {0, 3}


def test_reciprocal_division_is_exact():
    """The register DTW kernel divides by multiplying with a correctly rounded reciprocal plus one
    fma correction (similarity.hip dtw_cost_rcp); on the operands it can meet that IS the IEEE
    quotient the reference's `/` produces (gamma.py:51-52) -- checked here on the CPU, exhaustively
    for integers up to 3000 and on 10^7 random dyadic rationals."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
    subprocess.check_call(['make', '-s', '-C', here, '_build/division_check'])
    out = subprocess.run([os.path.join(here, '_build', 'division_check'), '3000', '10000000'], capture_output=True,
                         text=True)
    assert out.returncode == 0 and 'bad 0' in out.stdout, out.stdout + out.stderr



# ---- extra.npz: structure patches with structure_patch_type == 'ego_graph' (aps:226-228) ----------------

def _extra():
    import os
    from conftest import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, 'extra.npz'), allow_pickle=False)


@pytest.mark.parametrize('radius', [1, 2])
def test_ego_graph_structure_patches_golden(tiny, radius):
    """The reference's ego-graph patches, as sets (the node order inside nx.ego_graph's node view is CPython-set
    order for small patches and graph order for large ones -- canonical here: graph order), and the border walks
    over them bit for bit given the reference's recorded views."""
    import json
    z = _extra()
    G = _graph(tiny)
    t = 'ego_r%d_' % radius
    hp = json.loads(str(z[t + 'hparams']))
    want = z[t + 'structure_anchors']
    got = IH.sample_structure_anchor_patches(G, want.shape[0], hp['sample_walk_len'], hp['rw_beta'], int(z['seed']),
                                             'ego_graph', radius)
    assert got.shape == want.shape
    for a, b in zip(got, want):
        assert sorted(a.tolist()) == sorted(b.tolist())
    pos = {v: i for i, v in enumerate(G.node_order)}
    for row in got:                                   # canonical order: the base graph's node order
        nz = [int(v) for v in row if v != 0]
        assert nz == sorted(nz, key=lambda v: pos[v])
    vb = [[int(v) for v in r if v != -1] for r in z[t + 'views_bor']]
    inb = [[int(v) for v in r if v != -1] for r in z[t + 'in_border']]
    assert all(IH.patch_in_border_nodes(G, v) == b for v, b in zip(vb, inb))
    bw = IH.perform_random_walks(G, want, hp['n_triangular_walks'], hp['random_walk_len'], hp['rw_beta'], False,
                                 int(z['seed']), patch_orders=vb, in_borders=inb)
    assert np.array_equal(bw, z[t + 'bor_rw'])


# ---- fastdtw restatement: the three predecessor rules --------------------------------------------------

def test_fastdtw_tie_orders_bound_exact_dtw():
    """Whatever the compiled fastdtw 0.3.4 does on ties, each plausible rule (oracle/fastdtw_restate.py) is a
    valid DTW recurrence: its cost is that of some warp path, hence >= the exact DTW distance, and equal to it
    when the window is the whole grid (a series shorter than radius + 2, or a window that covers it).  The C
    oracle agrees with the Python restatement for every rule, and the rules do differ on some inputs."""
    from oracle import cbind
    rng = np.random.default_rng(5)
    differ = 0
    xs, ys = [], []
    for _ in range(300):
        lx, ly = int(rng.integers(1, 24)), int(rng.integers(1, 40))
        x = np.sort(rng.integers(0, 12, lx)).tolist()
        y = np.sort(rng.integers(0, 30, ly)).tolist()
        xs.append(x)
        ys.append(y)
        exact = FD.exact_dtw(x, y, FD.calc_dist)
        costs = [FD.fastdtw(x, y, 1, FD.calc_dist, t)[0] for t in (0, 1, 2)]
        for c in costs:
            assert c >= exact - 1e-12
            if lx < 3 or ly < 3:
                assert c == exact
        differ += len(set(costs)) > 1
    assert differ > 0
    xp, xv = cbind.ragged(xs)
    yp, yv = cbind.ragged(ys)
    for t in (0, 1, 2):
        got = cbind.fastdtw_sim(xp, xv, yp, yv, t)
        for i in range(0, 300, 7):
            for j in range(0, 300, 11):
                assert got[i, j] == np.float32(FD.calc_dtw(xs[i], ys[j], t))


def test_c_bfs_min_hops_matches_the_apsp_golden(tiny):
    """oracle_c's per-source BFS + min over members (the cpu_baseline leg's position stage) against the reference's
    all-pairs matrix: the g4 position similarities of the golden, for every anchor node as a source."""
    from oracle import cbind
    G = _graph(tiny)
    rowptr, col = G.csr()
    cc = tiny['g2_cc_ids_train']
    S, C, L = cc.shape
    ptr, flat = cbind.ragged([[int(v) for v in r if v] for r in cc.reshape(S * C, L)])
    src = np.arange(1, len(rowptr) - 1, dtype=np.int32)              # every node id as a source: the dense slab's columns
    got = cbind.bfs_min_hops_to_sets(rowptr, col, src, ptr, flat)
    want = tiny['g4_np_sim_train'].reshape(S * C, -1)
    assert np.array_equal(got, want.astype(np.float32))
