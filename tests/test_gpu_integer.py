"""-m gpu parity tests of the integer-half HIP kernels (through the C ABI) against the oracle,
on the golden fixtures (reference-pinned) and on seeded random inputs.  Bit-exact."""
import numpy as np
import pytest
import torch

from oracle import graph as OG, integer_half as IH, tape as T, fastdtw_restate as FD, cbind

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ops():
    from subgnn_amd import ops
    return ops


def _graphs(golden):
    ops = _ops()
    G = OG.from_edge_pairs([tuple(e) for e in golden['edge_list']])
    rp, col = G.csr()
    deg = np.zeros(G.max_id() + 1, dtype=np.int32)
    for v in G.node_order:
        deg[v] = G.degree(v)
    return G, ops.DeviceGraph(rp, col, G.node_order, DEV, full_degree=deg)


def _rand_graph(n, m, seed):
    import networkx as nx
    Gx = nx.barabasi_albert_graph(n, m, seed=seed)
    rng = np.random.default_rng(seed)
    edges = list(Gx.edges())
    edges = [edges[i] for i in rng.permutation(len(edges))]
    edges += [(3, 3), (10, 10), (0, 0)]                   # self loops (one of them on a hub whose list is searched, not streamed)
    edges += [(0, i) for i in range(1, min(n, 1300))] + [(7, i) for i in range(8, min(n, 700))]   # hubs (deg >= 256 path)
    return OG.from_edge_pairs(edges)


def _dev_graph(G, with_deg=True):
    ops = _ops()
    rp, col = G.csr()
    deg = np.zeros(G.max_id() + 1, dtype=np.int32)
    for v in G.node_order:
        deg[v] = G.degree(v)
    return ops.DeviceGraph(rp, col, G.node_order, DEV, full_degree=deg if with_deg else None)


# ---- a10 degree sequence ------------------------------------------------------------------

def test_degree_sequence_golden(golden):
    ops = _ops()
    G, dg = _graphs(golden)
    sa = torch.from_numpy(golden['g5_structure_anchors']).to(DEV)
    cc = torch.from_numpy(golden['g2_cc_ids_train']).to(DEV)
    cc = cc.view(-1, cc.shape[-1])
    for ids, ki, ke in ((sa, 'g6_anchor_deg_int', 'g6_anchor_deg_ext'), (cc, 'g6_cc_deg_int_train', 'g6_cc_deg_ext_train')):
        r = ops.Ragged.from_padded(ids)
        for use_dict in (True, False):
            oi, oe = ops.degree_sequence(dg, r, sort=True, use_degree_dict=use_dict)
            got_i = ops.Ragged(r.ptr, oi).to_lists()
            got_e = ops.Ragged(r.ptr, oe).to_lists()
            assert got_i == golden.ragged(ki, -1)
            assert got_e == golden.ragged(ke, -1)        # degree dict == graph degree in the fixtures


@pytest.mark.parametrize('sizes', [(1, 64), (60, 70), (65, 300), (1000, 2048)])
def test_degree_sequence_random(sizes):
    """wave path (<=64), block path (65..2048), duplicates, self loops, empty sets, vs the C oracle."""
    ops = _ops()
    G = _rand_graph(3000, 6, 7)
    dg = _dev_graph(G, with_deg=False)
    rng = np.random.default_rng(sizes[1])
    sets = []
    for i in range(300):
        n = int(rng.integers(sizes[0], sizes[1] + 1))
        s = rng.integers(1, G.max_id() + 1, n).tolist()
        if i % 7 == 0:
            s = s + s[:3]                                  # duplicates
        if i % 11 == 0:
            s = [3, 10] + s
        if i % 5 == 0:
            s = [1, 8] + s                                 # the two hubs (ids are +1)
        s = s[:sizes[1]]
        sets.append(s)
    sets[5] = []
    r = ops.Ragged.from_lists(sets, DEV)
    rp, col = G.csr()
    hf = ops.heaviest_first(dg, r)                       # a permutation; the dispatch order changes nothing
    assert sorted(hf.cpu().tolist()) == list(range(len(sets)))
    a0, b0 = ops.degree_sequence(dg, r, use_degree_dict=False)
    a1, b1 = ops.degree_sequence(dg, r, use_degree_dict=False, order=hf)
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    hub = dg.hub_tables()                                # the two hub lists (1300 and 700 entries) have membership bitmaps
    assert hub is not None and hub[1].shape == (dg.max_id + 1, 1) and hub[2] == 1 and int((hub[0] >= 0).sum()) == 2
    for table in (True, False):                          # hub lists streamed / searched for the members / answered from their bitmaps
        a2, b2 = ops.degree_sequence(dg, r, use_degree_dict=False, use_self_loop_table=table, search_long_lists=False)
        a3, b3 = ops.degree_sequence(dg, r, use_degree_dict=False, use_self_loop_table=table, search_long_lists=True, hub_bitmaps=False)
        a4, b4 = ops.degree_sequence(dg, r, use_degree_dict=False, use_self_loop_table=table, search_long_lists=True, hub_bitmaps=True)
        assert torch.equal(a2, a3) and torch.equal(b2, b3)
        assert torch.equal(a2, a4) and torch.equal(b2, b4)
    # the bitmap of a hub IS its neighbour list
    rp_, col_ = G.csr()
    bits = hub[1].cpu().numpy().view(np.uint32)                              # (rows = node ids, bit = the hub's number)
    for v in (1, 8):
        h = int(hub[0][v])
        ids = np.nonzero((bits[:, h >> 5] >> (h & 31)) & 1)[0]
        assert sorted(ids.tolist()) == sorted(set(col_[rp_[v]:rp_[v + 1]].tolist()))
    for srt in (True, False):
        for table in (True, False):        # self loops from the per-node table / counted while streaming
            oi, oe = ops.degree_sequence(dg, r, sort=srt, use_degree_dict=False, use_self_loop_table=table)
            ptr, flat = cbind.ragged(sets)
            ci, ce = cbind.degree_sequence(rp, col, None, ptr, flat, srt)
            n = int(ptr[-1])
            assert np.array_equal(oi.cpu().numpy()[:n], ci)
            assert np.array_equal(oe.cpu().numpy()[:n], ce)


def test_hub_bitmaps_respect_their_memory_budget():
    """DeviceGraph.hub_tables builds nothing beyond HUB_BITMAP_BYTES (and nothing for a graph without long lists): the
    degree-sequence call then searches the long lists as before -- same results."""
    ops = _ops()
    G = _rand_graph(3000, 6, 7)
    small = _dev_graph(G, with_deg=False)
    small.HUB_BITMAP_BYTES = 1024                      # (3002 nodes x 1 word = 12 KB would be needed)
    assert small.hub_tables() is None and small.node_records() is None
    full = _dev_graph(G, with_deg=False)
    assert full.hub_tables() is not None and full.node_records().shape == (full.max_id + 1, 4)
    rng = np.random.default_rng(1)
    sets = [[1, 8] + rng.integers(1, G.max_id() + 1, 18).tolist() for _ in range(200)]
    r = ops.Ragged.from_lists(sets, DEV)
    a, b = ops.degree_sequence(small, r, use_degree_dict=False)
    c, d = ops.degree_sequence(full, r, use_degree_dict=False)
    assert torch.equal(a, c) and torch.equal(b, d)
    import networkx as nx
    flat = OG.from_edge_pairs(list(nx.path_graph(50).edges()))
    assert _dev_graph(flat, with_deg=False).hub_tables() is None       # no list of >= 512 entries


def test_degree_sequence_of_a_set_beyond_2048_is_served():
    """Rounds 1-2 answered SGNN_ERR_SET_TOO_LARGE here; since round 3 such a set takes the workspace-backed kernel."""
    ops = _ops()
    G = _rand_graph(3000, 3, 1)
    dg = _dev_graph(G)
    rp, col = G.csr()
    members = list(range(1, 2500))
    r = ops.Ragged.from_lists([members], DEV)
    oi, oe = ops.degree_sequence(dg, r, use_degree_dict=False)
    ptr, flat = cbind.ragged([members])
    ci, ce = cbind.degree_sequence(rp, col, None, ptr, flat, True)
    assert np.array_equal(oi.cpu().numpy(), ci) and np.array_equal(oe.cpu().numpy(), ce)


# ---- a7 connected components --------------------------------------------------------------

def _labels_to_sets(nodes, labels):
    out = {}
    for v, l in zip(nodes, labels):
        out.setdefault(l, set()).add(v)
    return {frozenset(s) for s in out.values()}


def test_cc_labels_golden(golden):
    ops = _ops()
    G, dg = _graphs(golden)
    for sp in ('train', 'val'):
        subs = golden.ragged('subgraphs_' + sp, 0)
        r = ops.Ragged.from_lists(subs, DEV)
        lab = ops.Ragged(r.ptr, ops.cc_labels(dg, r)).to_lists()
        cc = golden['g2_cc_ids_' + sp]
        for s in range(len(subs)):
            ref = {frozenset(int(v) for v in row if v != 0) for row in cc[s] if row[0] != 0}
            assert _labels_to_sets(subs[s], lab[s]) == ref
            # the label is the smallest position in the component
            for i, l in enumerate(lab[s]):
                assert l <= i and lab[s][l] == l


def test_cc_labels_random():
    ops = _ops()
    G = _rand_graph(2000, 2, 3)
    dg = _dev_graph(G)
    rng = np.random.default_rng(0)
    subs = [rng.integers(1, G.max_id() + 1, int(rng.integers(1, 200))).tolist() for _ in range(100)]
    r = ops.Ragged.from_lists(subs, DEV)
    lab = ops.Ragged(r.ptr, ops.cc_labels(dg, r)).to_lists()
    for s in range(len(subs)):
        ref = {frozenset(c) for c in IH.connected_components(G, subs[s])}
        assert _labels_to_sets(subs[s], lab[s]) == ref


def test_components_from_labels(golden):
    """labels (smallest position per component) -> padded (S,C,L) tensor in canonical order (HIP compaction)."""
    _ops()
    from subgnn_amd.subgraph_utils import components_from_labels
    G, _dg = _graphs(golden)
    subs = golden.ragged('subgraphs_train', 0)
    subs[0] = subs[0] + subs[0][:2]                         # duplicates must collapse
    ptr = np.zeros(len(subs) + 1, dtype=np.int64)
    flat, labels = [], []
    for i, s in enumerate(subs):
        comps = IH.connected_components(G, s)
        where = {}
        for c in comps:
            first = min(s.index(v) for v in c)
            for v in c:
                where[v] = first
        flat += s
        labels += [where[v] for v in s]
        ptr[i + 1] = ptr[i] + len(s)
    out = components_from_labels(torch.from_numpy(ptr).to(DEV), torch.tensor(flat, dtype=torch.int32, device=DEV),
                                 torch.tensor(labels, dtype=torch.int32, device=DEV)).cpu().numpy()
    for i, s in enumerate(subs):
        ref = [c for c in IH.connected_components(G, s)]
        got = [[int(v) for v in row if v != 0] for row in out[i] if row[0] != 0]
        assert got == ref                                   # canonical order == the oracle's canonical order

def test_cc_compact_large_and_duplicates():
    """Subgraphs beyond 64 nodes (hash-based duplicate detection, several 64-node chunks per
    component), many components, duplicates, an empty subgraph: cc_labels + cc_compact == the oracle."""
    ops = _ops()
    from subgnn_amd.subgraph_utils import components_from_labels
    G = _rand_graph(2500, 2, 13)
    dg = _dev_graph(G)
    rng = np.random.default_rng(2)
    subs = []
    for i in range(60):
        n = int(rng.integers(1, 400)) if i % 3 else int(rng.integers(1, 40))
        s = rng.integers(1, G.max_id() + 1, n).tolist()
        if i % 4 == 0:
            s = s + s[:5]
        subs.append(s)
    subs[7] = []
    r = ops.Ragged.from_lists(subs, DEV)
    out = components_from_labels(r.ptr, r.nodes, ops.cc_labels(dg, r), r.max_len).cpu().numpy()
    assert out.shape[0] == 60
    for i, s in enumerate(subs):
        ref = IH.connected_components(G, s) if s else []
        got = [[int(v) for v in row if v != 0] for row in out[i] if row[0] != 0]
        assert got == ref, i


def test_cc_of_subgraphs_beyond_the_lds_tables():
    """Subgraphs of 5000 and 3000 nodes (rounds 1-2 refused more than 2048; the reference pads to any size,
    SubGNN.py:575-607) next to small ones, with repeated ids: labels and the padded component tensor from the
    workspace-backed kernels (sgnn_cc_labels_huge / sgnn_cc_compact_huge) == the oracle, and the padded shape is tight."""
    ops = _ops()
    from subgnn_amd.subgraph_utils import components_from_labels
    G = _rand_graph(9000, 2, 21)
    dg = _dev_graph(G)
    rng = np.random.default_rng(5)
    ids = np.arange(1, G.max_id() + 1)
    subs = [rng.choice(ids, 5000, replace=False).tolist(), rng.integers(1, G.max_id() + 1, 30).tolist(),
            rng.choice(ids, 3000, replace=False).tolist(), [], rng.integers(1, G.max_id() + 1, 300).tolist()]
    subs[0] = subs[0] + subs[0][:7]                            # repeated ids in a huge subgraph
    subs[2][100] = subs[2][5]
    r = ops.Ragged.from_lists(subs, DEV)
    assert r.max_len > 2048
    lab = ops.Ragged(r.ptr, ops.cc_labels(dg, r)).to_lists()
    for s in (0, 1, 2, 4):
        ref = {frozenset(c) for c in IH.connected_components(G, subs[s])}
        assert _labels_to_sets(subs[s], lab[s]) == ref, s
        for i, l in enumerate(lab[s]):
            assert l <= i and lab[s][l] == l                   # the label is the smallest position in the component
    out = components_from_labels(r.ptr, r.nodes, ops.cc_labels(dg, r), r.max_len).cpu().numpy()
    refs = [IH.connected_components(G, s) if s else [] for s in subs]
    assert out.shape == (5, max(len(c) for c in refs), max(len(cc) for c in refs for cc in c))
    for i, s in enumerate(subs):
        got = [[int(v) for v in row if v != 0] for row in out[i] if row[0] != 0]
        assert got == refs[i], i
    # a kept shape (no statistics launch) gives the same tensor
    again = components_from_labels(r.ptr, r.nodes, ops.cc_labels(dg, r), r.max_len, dims=out.shape[1:]).cpu().numpy()
    assert np.array_equal(again, out)


def test_degree_sequences_beyond_the_lds_tables():
    """Sets of 5000 / 2600 entries (with a repeated member and self loops in the graph) next to small ones: internal and
    external degree sequences, sorted and unsorted, with and without the degree dictionary == the C oracle."""
    ops = _ops()
    G = _rand_graph(9000, 3, 29)
    dg = _dev_graph(G)
    rowptr, col = G.csr()
    rng = np.random.default_rng(8)
    ids = np.arange(1, G.max_id() + 1)
    big = rng.choice(ids, 5000, replace=False).tolist()
    big[3], big[9], big[11] = 4, 1, 11                         # nodes with self loops (networkx degree counts them twice) and a hub
    big[40] = big[41]
    sets = [big, rng.integers(1, G.max_id() + 1, 50).tolist(), rng.choice(ids, 2600, replace=False).tolist(), [],
            rng.integers(1, G.max_id() + 1, 700).tolist()]
    r = ops.Ragged.from_lists(sets, DEV)
    ptr, flat = cbind.ragged(sets)
    for srt in (True, False):
        ci, ce = cbind.degree_sequence(rowptr, col, None, ptr, flat, srt)
        gi, ge = ops.degree_sequence(dg, r, sort=srt, use_degree_dict=False)
        assert np.array_equal(gi.cpu().numpy()[:len(ci)], ci) and np.array_equal(ge.cpu().numpy()[:len(ce)], ce), srt


def test_patch_in_border_beyond_the_lds_table():
    """A patch of 5000 nodes (an ego-graph patch around hubs) next to small ones: in-border flags from the
    workspace-backed kernel == the oracle (the reference's id - 1 / node-order quirk included)."""
    ops = _ops()
    G = _rand_graph(8000, 3, 17)
    dg = _dev_graph(G)
    rng = np.random.default_rng(3)
    ids = np.arange(1, G.max_id() + 1)
    views = [rng.choice(ids, 5000, replace=False).tolist(), rng.choice(ids, 40, replace=False).tolist(),
             rng.choice(ids, 2500, replace=False).tolist()]
    vr = ops.Ragged.from_lists(views, DEV)
    flags = ops.Ragged(vr.ptr, ops.patch_in_border(dg, vr).to(torch.int32)).to_lists()
    for v, f in zip(views, flags):
        want = set(IH.patch_in_border_nodes(G, v))
        assert [x for x, b in zip(v, f) if b] == [x for x in v if x in want]
        assert set(f) <= {0, 1}


# ---- a8 k-hop border ----------------------------------------------------------------------

def test_khop_border_golden(golden):
    ops = _ops()
    G, dg = _graphs(golden)
    cc = torch.from_numpy(golden['g2_cc_ids_train']).to(DEV)
    r = ops.Ragged.from_padded(cc.view(-1, cc.shape[-1]))
    for k, lds in ((1, True), (2, True), (3, True), (1, False), (2, False), (3, False)):
        b, hops = ops.khop_border(dg, r, k, ego_dict_mode=golden.has_ego, want_hops=True, bitmap_in_lds=lds)
        b, hops = ops.sort_ragged(b, hops)
        assert b.to_lists() == golden.ragged('g3_border_k%d_train' % k, -1)
        if not golden.has_ego:
            ccn = golden['g2_cc_ids_train'].reshape(-1, golden['g2_cc_ids_train'].shape[-1])
            hl = ops.Ragged(b.ptr, hops.to(torch.int32)).to_lists()
            for row in range(ccn.shape[0]):
                lev = IH.border_hop_levels(G, ccn[row], k)
                assert hl[row] == [lev[v] for v in b.to_lists()[row]]


def test_khop_border_workspace_left_clean():
    """Many more sets than workgroups: a workgroup's bitmap must be clean for its next set."""
    ops = _ops()
    G = _rand_graph(500, 3, 5)
    dg = _dev_graph(G)
    rng = np.random.default_rng(1)
    sets = [list({int(v) for v in rng.integers(1, G.max_id() + 1, int(rng.integers(1, 6)))}) for _ in range(3000)]
    r = ops.Ragged.from_lists(sets, DEV)
    for lds in (True, False):
        b = ops.sort_ragged(ops.khop_border(dg, r, 2, bitmap_in_lds=lds)).to_lists()
        for i in range(0, 3000, 37):
            assert b[i] == sorted(IH.component_border_set(G, sets[i], 2))


@pytest.mark.parametrize('lds', [True, False])
def test_khop_border_one_pass_arena(lds):
    """The arena variant (BFS queue = caller's slice) materialises the same 1-hop borders."""
    ops = _ops()
    G = _rand_graph(400, 3, 21)
    dg = _dev_graph(G)
    rng = np.random.default_rng(8)
    sets = [list({int(v) for v in rng.integers(1, G.max_id() + 1, int(rng.integers(1, 6)))}) for _ in range(900)]
    r = ops.Ragged.from_lists(sets, DEV)
    arena, off, counts = ops.khop_border_one_pass(dg, r, bitmap_in_lds=lds)
    ref = ops.sort_ragged(ops.khop_border(dg, r, 1, bitmap_in_lds=lds)).to_lists()
    a, o, c = arena.cpu().numpy(), off.cpu().numpy(), counts.cpu().numpy()
    for i in range(900):
        assert sorted(a[o[i]:o[i] + c[i]].tolist()) == ref[i]


@pytest.mark.parametrize('lds', [True, False, 32])          # 32: an LDS bitmap of 32 bytes = 256 ids per slice (k = 1: sliced)
@pytest.mark.parametrize('k', [1, 2])
def test_khop_border_sample_equals_materialised_draw(k, lds):
    """Fused BFS + anchor draw == (materialise the border, pad it, run the reference-shaped
    padded sampler), including the PAD rule and the hop level of every drawn anchor."""
    ops = _ops()
    G = _rand_graph(600, 2, 11)
    dg = _dev_graph(G)
    rng = np.random.default_rng(4)
    sets = [list({int(v) for v in rng.integers(1, G.max_id() + 1, int(rng.integers(1, 5)))}) for _ in range(700)]
    sets[3] = []
    sets[5] = list(range(1, 151))                   # more members than one 64-lane tile
    sets[6] = list(range(1, G.max_id() + 1))        # everything: empty border
    r = ops.Ragged.from_lists(sets, DEV)
    A, seed, st = 7, 99, T.stream_id(T.STREAM_N_BOR, 'val', 1)
    anchors, sims, counts = ops.khop_border_sample(dg, r, k, A, seed, st, bitmap_in_lds=lds)
    b, hops = ops.sort_ragged(*ops.khop_border(dg, r, k, want_hops=True))
    padded = b.to_padded()
    ref = ops.sample_anchors_padded(padded, A, seed, st)
    assert torch.equal(anchors, ref)
    assert torch.equal(counts, b.lengths)
    bl, hl = b.to_lists(), ops.Ragged(b.ptr, hops.to(torch.int32)).to_lists()
    an, sm = anchors.cpu().numpy(), sims.cpu().numpy()
    for i in range(0, 700, 13):
        lev = dict(zip(bl[i], hl[i]))
        for a in range(A):
            assert sm[i, a] == (lev[an[i, a]] if an[i, a] != 0 else 0)
    assert (an == 0).any()          # the PAD rule fires on small borders
    # and against the oracle (ascending border, tape pick) for a sample of rows
    mx = int(counts.max())
    for i in range(0, 700, 29):
        real = sorted(bl[i])
        for a in range(A):
            kk = T.nanchor_pick(seed, st, i * A + a, len(real), len(real) < mx)
            assert an[i, a] == (0 if kk < 0 else real[kk])


def test_khop_border_sample_large_border_rank_query():
    """Hubs: borders of thousands of nodes spread over the whole id range, many anchor slots (more
    than one 256-slot pass) -- every drawn anchor is the tape-ranked element of the sorted border."""
    from subgnn_amd import synthetic
    ops = _ops()
    n = 40000
    rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, 8, seed=3), n)
    dg = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), DEV)
    sets = synthetic.bfs_subgraphs(rowptr, col, 300, 12, seed=5)
    r = ops.Ragged.from_lists(sets, DEV)
    A, seed, st = 300, 7, T.stream_id(T.STREAM_N_BOR, 'train', 0)
    for lds in (True, False, 1024):                  # 1024 bytes of bitmap: 8192 ids per slice, 5 slices
        for k in (1, 2):
            anchors, sims, counts = ops.khop_border_sample(dg, r, k, A, seed, st, bitmap_in_lds=lds)
            b, hops = ops.sort_ragged(*ops.khop_border(dg, r, k, want_hops=True))
            assert torch.equal(counts, b.lengths)
            bl, hl = b.to_lists(), ops.Ragged(b.ptr, hops.to(torch.int32)).to_lists()
            an, sm, mx = anchors.cpu().numpy(), sims.cpu().numpy(), int(counts.max())
            for i in range(0, 300, 17):
                lev = dict(zip(bl[i], hl[i]))
                for a in range(0, A, 7):
                    kk = T.nanchor_pick(seed, st, i * A + a, len(bl[i]), len(bl[i]) < mx)
                    assert an[i, a] == (0 if kk < 0 else bl[i][kk])
                    assert sm[i, a] == (lev[an[i, a]] if an[i, a] != 0 else 0)


def test_khop_border_sample_beyond_the_lds_bitmap():
    """1.5 M ids: more than the 150 KB LDS bitmap holds (1.22 M).  The one-hop draw then processes the id
    range in two LDS-sized slices (rows ascending; pass A counts per slice, pass B answers the slots);
    results equal the workspace-bitmap kernel's and the tape-ranked sorted border, and item_base shifts
    the tape items of a shard of the rows."""
    from subgnn_amd import synthetic, _lib
    ops = _ops()
    n = 1_500_000
    rowptr, col = synthetic.barabasi_albert_csr_device(n, 4, 9, torch.device(DEV))
    dg = ops.DeviceGraph.from_device_csr(rowptr, col)
    lib = _lib.load()
    assert not lib.sgnn_khop_border_bitmap_fits_lds(dg.max_id)
    assert lib.sgnn_khop_border_sample_workspace_bytes(dg.max_id, 100, 1, 1, 1) == 16          # sliced LDS path
    assert lib.sgnn_khop_border_sample_workspace_bytes(dg.max_id, 100, 2, 1, 1) > 1 << 20      # k = 2: bitmap in workspace
    rp, cl = rowptr.cpu().numpy(), col.cpu().numpy()
    sets = synthetic.bfs_subgraphs(rp, cl, 1500, 20, seed=2)
    r = ops.Ragged.from_lists(sets, DEV)
    A, seed, st = 43, 5, T.stream_id(T.STREAM_N_BOR, 'train', 0)
    a1, s1, c1 = ops.khop_border_sample(dg, r, 1, A, seed, st)                     # sliced, LDS
    a0, s0, c0 = ops.khop_border_sample(dg, r, 1, A, seed, st, bitmap_in_lds=False)
    assert torch.equal(a1, a0) and torch.equal(s1, s0) and torch.equal(c1, c0)
    an, cn, mx = a1.cpu().numpy(), c1.cpu().numpy(), int(c1.max())
    for i in range(0, 1500, 97):
        members = np.asarray(sets[i])
        border = np.setdiff1d(np.unique(np.concatenate([cl[rp[v]:rp[v + 1]] for v in members])), members)
        assert cn[i] == len(border)
        for a in range(A):
            kk = T.nanchor_pick(seed, st, i * A + a, len(border), len(border) < mx)
            assert an[i, a] == (0 if kk < 0 else border[kk])
    # a shard of the rows with item_base reproduces those rows of the full call (same padded width)
    sub = ops.Ragged.from_lists(sets[700:900], DEV)
    width = c1.max().view(1)
    a2, s2, c2 = ops.khop_border_sample(dg, sub, 1, A, seed, st, item_base=700, count_reduce=lambda t: torch.maximum(t, width))
    assert torch.equal(a2, a1[700:900]) and torch.equal(c2, c1[700:900])


# ---- a4 neighbourhood anchors -------------------------------------------------------------

def test_sample_anchors_golden(golden):
    ops = _ops()
    hp, seed = golden.hp, golden.seed
    for sp in ('train', 'val'):
        for l in range(hp['n_layers']):
            for mat, A, kind, key in ((golden['g2_cc_ids_' + sp], hp['n_anchor_patches_N_in'], T.STREAM_N_INT, 'g8_N_int_%s_%d'),
                                      (golden['g3_border_' + sp], hp['n_anchor_patches_N_out'], T.STREAM_N_BOR, 'g8_N_bor_%s_%d')):
                S, C, L = mat.shape
                ids = torch.from_numpy(mat).to(DEV).view(S * C, L).contiguous()
                st = T.stream_id(kind, sp, l)
                got = ops.sample_anchors_padded(ids, A, seed, st).view(S, C, A).cpu().numpy()
                assert np.array_equal(got, golden[key % (sp, l)])
                # ragged form: same law, same tape, row_has_pad from the padded width
                r = ops.Ragged.from_padded(ids)
                has_pad = (r.lengths < L).to(torch.uint8)
                got2 = ops.sample_anchors_ragged(r, A, seed, st, has_pad).view(S, C, A).cpu().numpy()
                assert np.array_equal(got2, golden[key % (sp, l)])


def test_sample_anchors_order_independent():
    """the pick is a rank among the ascending entries: permuting a row leaves the sample unchanged."""
    ops = _ops()
    rng = np.random.default_rng(3)
    ids = np.zeros((50, 40), dtype=np.int64)
    for r in range(50):
        n = int(rng.integers(1, 41))
        ids[r, :n] = rng.choice(np.arange(1, 5000), n, replace=False)
    perm = ids.copy()
    for r in range(50):
        n = int((ids[r] != 0).sum())
        perm[r, :n] = rng.permutation(ids[r, :n])
    a = ops.sample_anchors_padded(torch.from_numpy(ids).to(DEV), 9, 5, 77)
    b = ops.sample_anchors_padded(torch.from_numpy(perm).to(DEV), 9, 5, 77)
    assert torch.equal(a, b)


# ---- a5/a6 choices ------------------------------------------------------------------------

def test_choice_golden(golden):
    ops = _ops()
    G, dg = _graphs(golden)
    hp, seed = golden.hp, golden.seed
    for sp in ('train', 'val'):
        subs = ops.Ragged.from_lists(golden.ragged('subgraphs_' + sp, 0), DEV)
        for l in range(hp['n_layers']):
            got = ops.choice_ragged(subs, hp['n_anchor_patches_pos_in'], seed, T.stream_id(T.STREAM_P_INT, sp, l))
            assert np.array_equal(got.cpu().numpy(), golden['g8_P_int_%s_%d' % (sp, l)])
    order = ops.Ragged.from_lists([G.node_order], DEV)
    npatch = golden['g5_structure_anchors'].shape[0]
    rng_list = ops.Ragged.from_lists([list(range(npatch))], DEV)
    for l in range(hp['n_layers']):
        got = ops.choice_ragged(order, hp['n_anchor_patches_pos_out'], seed, T.stream_id(T.STREAM_P_EXT, 0, l))
        assert np.array_equal(got.cpu().numpy()[0], golden['g8_P_ext_%d' % l])
        got = ops.choice_ragged(rng_list, hp['n_anchor_patches_structure'], seed, T.stream_id(T.STREAM_S_PICK, 0, l))
        assert np.array_equal(got.cpu().numpy()[0], golden['g8_S_idx_%d' % l])


def test_sample_position_anchor_patches_per_subgraph(golden):
    """aps:200-208 with the reference's own signature: one subgraph at a time (internal) or none (the
    shared border anchors) -> the python lists the reference returns, equal to the reference's draws
    under the tape (g8) -- and hence to the batched init_anchors_pos_int."""
    from subgnn_amd import anchor_patch_samplers as aps
    G, dg = _graphs(golden)
    hp = dict(golden.hp)
    hp['seed'] = golden.seed
    for l in range(hp['n_layers']):
        got = aps.sample_position_anchor_patches(hp, dg, None, layer=l)
        assert isinstance(got, list) and got == golden['g8_P_ext_%d' % l].tolist()
        for sp in ('train', 'val'):
            subs = golden.ragged('subgraphs_' + sp, 0)
            for i in (0, len(subs) // 2, len(subs) - 1):
                got = aps.sample_position_anchor_patches(hp, dg, subs[i], split=sp, layer=l, item=i)
                assert got == golden['g8_P_int_%s_%d' % (sp, l)][i].tolist()


# ---- a1-a3 walks --------------------------------------------------------------------------

@pytest.fixture(params=['workgroup', 'wavefront'])
def walk_kernel(request):
    """Both walk kernels: a workgroup per walk with the LDS adjacency bitmap (the default when the
    id range fits) and a wavefront per walk with binary searches -- the ``kernel`` argument of the call."""
    return 1 if request.param == 'wavefront' else 0


def test_walks_golden(golden, walk_kernel):
    ops = _ops()
    G, dg = _graphs(golden)
    hp, seed = golden.hp, golden.seed
    sa = golden['g5_structure_anchors']
    n = sa.shape[0]
    got = ops.triangular_walks(dg, 0, n, hp['sample_walk_len'], hp['rw_beta'], seed, T.stream_id(T.STREAM_STRUCT_PATCH), kernel=walk_kernel)
    assert np.array_equal(got.cpu().numpy()[:, :sa.shape[1]], sa)
    assert (got.cpu().numpy()[:, sa.shape[1]:] == 0).all()
    W, Tn = hp['n_triangular_walks'], hp['random_walk_len']
    views = ops.Ragged.from_lists(golden.ragged('g5_views_int', 0), DEV)
    iw = ops.triangular_walks(dg, 1, n * W, Tn, hp['rw_beta'], seed, T.stream_id(T.STREAM_WALK_INT), patches=views,
                              walks_per_patch=W, kernel=walk_kernel)
    assert np.array_equal(iw.view(n, W, Tn).cpu().numpy(), golden['g5_int_walks'])
    vb = ops.Ragged.from_lists(golden.ragged('g5_views_bor', 0), DEV)
    flags = ops.patch_in_border(dg, vb)
    tot = int(vb.ptr[-1].item())
    inb_lists = ops.Ragged(vb.ptr, (vb.nodes * flags.to(torch.int32))).to_lists()
    inb_lists = [[v for v in row if v != 0] for row in inb_lists]
    assert inb_lists == golden.ragged('g5_in_border', 0)
    inb = ops.Ragged.from_lists(inb_lists, DEV)
    bw = ops.triangular_walks(dg, 2, n * W, Tn, hp['rw_beta'], seed, T.stream_id(T.STREAM_WALK_BOR), patches=vb,
                              in_border=inb, walks_per_patch=W, kernel=walk_kernel)
    assert np.array_equal(bw.view(n, W, Tn).cpu().numpy(), golden['g5_bor_walks'])


def test_walks_random_vs_oracle(walk_kernel):
    ops = _ops()
    G = _rand_graph(400, 4, 9)
    dg = _dev_graph(G)
    patches = IH.sample_structure_anchor_patches(G, 40, 30, 0.4, 123)
    got = ops.triangular_walks(dg, 0, 40, 30, 0.4, 123, T.stream_id(T.STREAM_STRUCT_PATCH), kernel=walk_kernel).cpu().numpy()
    assert np.array_equal(got[:, :patches.shape[1]], patches)
    views = [IH.patch_unique_nodes(p) for p in patches]
    inb = [IH.patch_in_border_nodes(G, v) for v in views]
    vr, ir = ops.Ragged.from_lists(views, DEV), ops.Ragged.from_lists(inb, DEV)
    for inside in (True, False):
        ref = IH.perform_random_walks(G, patches, 4, 12, 0.4, inside, 123)
        st = T.stream_id(T.STREAM_WALK_INT if inside else T.STREAM_WALK_BOR)
        got = ops.triangular_walks(dg, 1 if inside else 2, 160, 12, 0.4, 123, st, patches=vr, in_border=ir,
                                   walks_per_patch=4, kernel=walk_kernel).view(40, 4, 12).cpu().numpy()
        assert np.array_equal(got, ref)


def test_index_rows_many_equals_index_select():
    """sgnn_gather_rows_many (ops.index_rows_many: a batch's rows of every per-split tensor in one launch) against
    torch.index_select: int64 / int32 / float32 / uint8 tensors, rows of 1 byte to 1.2 MB, odd widths (byte and 4-byte paths),
    repeated and unordered indices, a non-contiguous tensor (falls back), 30 tensors (two launches)."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    S = 57
    ts = [torch.randint(0, 1000, (S, 7, 8), generator=g), torch.randint(0, 1000, (S, 5), generator=g).to(torch.int32),
          torch.randn(S, 3, 100003, generator=g), torch.randint(0, 255, (S,), generator=g).to(torch.uint8),
          torch.randint(0, 255, (S, 3), generator=g).to(torch.uint8), torch.randn(S, generator=g), torch.randn(S, 13, generator=g),
          torch.randn(S, 20, 6, generator=g)[:, ::2]]
    ts += [torch.randn(S, k + 1, generator=g) for k in range(22)]
    ts = [t.to(DEV) for t in ts]
    ts[7] = torch.randn(S, 20, 6, generator=g).to(DEV)[:, ::2]
    assert not ts[7].is_contiguous()
    idx = torch.tensor([5, 0, 56, 5, 5, 31, 2, 2, 40], dtype=torch.int64, device=DEV)
    got = ops.index_rows_many(ts, idx)
    for t, o in zip(ts, got):
        assert torch.equal(o, t.index_select(0, idx)) and o.dtype == t.dtype
    assert ops.index_rows_many([], idx) == []
    one = ops.index_rows_many(ts[:1], idx)                           # a single tensor: the library call
    assert torch.equal(one[0], ts[0].index_select(0, idx))


def test_internal_and_border_walks_in_one_launch_equal_the_oracle():
    """sgnn_triangular_walks_both: the internal and the border walks of the same patches in ONE launch against the oracle's two
    calls (aps:118-158 with inside = True / False), and a share of it (item_base) against the whole."""
    ops = _ops()
    G = _rand_graph(400, 4, 9)
    dg = _dev_graph(G)
    patches = IH.sample_structure_anchor_patches(G, 40, 30, 0.4, 123)
    views = [IH.patch_unique_nodes(p) for p in patches]
    inb = [IH.patch_in_border_nodes(G, v) for v in views]
    vr, ir = ops.Ragged.from_lists(views, DEV), ops.Ragged.from_lists(inb, DEV)
    both = ops.triangular_walks_both(dg, 160, 12, 0.4, 123, T.stream_id(T.STREAM_WALK_INT), T.stream_id(T.STREAM_WALK_BOR), vr, ir, 4)
    assert tuple(both.shape) == (2, 160, 12)
    for k, inside in enumerate((True, False)):
        ref = IH.perform_random_walks(G, patches, 4, 12, 0.4, inside, 123)
        assert np.array_equal(both[k].view(40, 4, 12).cpu().numpy(), ref), inside
    lo, hi = 13, 29                                                  # patches 13..28 as a share: same walks
    vs, is_ = ops.Ragged.from_lists(views[lo:hi], DEV), ops.Ragged.from_lists(inb[lo:hi], DEV)
    part = ops.triangular_walks_both(dg, (hi - lo) * 4, 12, 0.4, 123, T.stream_id(T.STREAM_WALK_INT), T.stream_id(T.STREAM_WALK_BOR),
                                     vs, is_, 4, item_base=lo * 4)
    assert torch.equal(part, both[:, lo * 4:hi * 4])


def test_a_share_of_the_walks_draws_what_the_whole_launch_draws(walk_kernel):
    """item_base: a rank that runs walks [lo, hi) of a launch -- with the node views / in-border sets of ITS patches only
    for the patch walks -- gets rows lo..hi-1 of the whole launch, bit for bit (strong scaling deals the shared patches'
    walks this way: hotpath._deal_rows)."""
    ops = _ops()
    G = _rand_graph(500, 4, 21)
    dg = _dev_graph(G)
    sid = T.stream_id(T.STREAM_STRUCT_PATCH)
    whole = ops.triangular_walks(dg, 0, 37, 25, 0.5, 77, sid, kernel=walk_kernel)
    for lo, hi in ((0, 10), (10, 29), (29, 37)):
        part = ops.triangular_walks(dg, 0, hi - lo, 25, 0.5, 77, sid, kernel=walk_kernel, item_base=lo)
        assert torch.equal(part, whole[lo:hi])
    patches = whole.cpu().numpy()
    views = [IH.patch_unique_nodes(p) for p in patches]
    inb = [IH.patch_in_border_nodes(G, v) for v in views]
    W = 3
    for inside in (True, False):
        st = T.stream_id(T.STREAM_WALK_INT if inside else T.STREAM_WALK_BOR)
        mode = 1 if inside else 2
        full = ops.triangular_walks(dg, mode, 37 * W, 11, 0.5, 77, st, patches=ops.Ragged.from_lists(views, DEV),
                                    in_border=ops.Ragged.from_lists(inb, DEV), walks_per_patch=W, kernel=walk_kernel).view(37, W, 11)
        for lo, hi in ((0, 19), (19, 37)):
            part = ops.triangular_walks(dg, mode, (hi - lo) * W, 11, 0.5, 77, st, patches=ops.Ragged.from_lists(views[lo:hi], DEV),
                                        in_border=ops.Ragged.from_lists(inb[lo:hi], DEV), walks_per_patch=W, kernel=walk_kernel,
                                        item_base=lo * W).view(hi - lo, W, 11)
            assert torch.equal(part, full[lo:hi])


def test_walks_hubs_both_kernels_agree():
    """Long walks over hubs (lists longer than one 64-entry chunk per wavefront, several workgroup
    passes over the items): the two kernels return the same walks."""
    from subgnn_amd import synthetic
    ops = _ops()
    n = 30000
    rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, 12, seed=4), n)
    dg = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), DEV)
    out = {}
    for kernel in (0, 1):
        out[kernel] = ops.triangular_walks(dg, 0, 3000, 40, 0.65, 5, T.stream_id(T.STREAM_STRUCT_PATCH), kernel=kernel)
    assert torch.equal(out[0], out[1]) and int((out[0] != 0).sum()) > 3000 * 20


# ---- a9 shortest-path similarities --------------------------------------------------------

def test_sp_similarity_dense_golden(golden):
    ops = _ops()
    apsp = torch.from_numpy(golden['apsp']).to(DEV)
    for sp in ('train', 'val'):
        cc = torch.from_numpy(golden['g2_cc_ids_' + sp]).to(DEV)
        S, C, L = cc.shape
        got = ops.sp_similarity_dense(apsp, ops.Ragged.from_padded(cc.view(S * C, L))).view(S, C, -1)
        assert np.array_equal(got.cpu().numpy(), golden['g4_np_sim_' + sp])


@pytest.fixture
def bfs_alpha(request):
    """Direction switch of the multi-source BFS (the ``pull_alpha`` argument of the calls)."""
    return request.param


@pytest.mark.parametrize('n_src', [70, 300])
def test_bfs_hops_push_pull_agree_and_match_scipy(n_src):
    """The direction switch never changes a hop count: always-push, always-pull and the default mix
    agree on a scale-free graph (300 sources = 5 words: two register chunks in the pull pass), and
    equal scipy's BFS distances; isolated ids stay unreached."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import shortest_path
    from subgnn_amd import synthetic
    ops = _ops()
    n = 6000
    edges = synthetic.barabasi_albert_edges(n - 50, 6, seed=9)          # ids n-49..n are isolated
    rowptr, col = synthetic.sorted_csr(edges, n)
    dg = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), DEV)
    src = np.random.default_rng(n_src).integers(1, n + 1, n_src).astype(np.int32)
    src[0] = n                                                           # an isolated source
    out = {}
    for alpha in (0, 256, 1 << 30):
        out[alpha] = ops.bfs_hops(dg, torch.from_numpy(src).to(DEV), max_hops=32, pull_alpha=alpha)
        out[(alpha, 't')] = ops.bfs_hops(dg, torch.from_numpy(src).to(DEV), max_hops=32, node_major=True, pull_alpha=alpha)
    assert torch.equal(out[0], out[256]) and torch.equal(out[0], out[1 << 30])
    for alpha in (0, 256, 1 << 30):
        assert torch.equal(out[(alpha, 't')].t().contiguous(), out[0])
    # fused BFS + min over members == the two-step form (sets with isolated members, an empty set)
    rng = np.random.default_rng(1)
    sets_l = [rng.integers(1, n + 1, int(rng.integers(1, 30))).tolist() for _ in range(400)]
    sets_l[5] = []
    sets = ops.Ragged.from_lists(sets_l, DEV)
    two_step = ops.min_hops_to_sets(out[256], sets)
    assert torch.equal(ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=32), two_step)
    # round 5: a pull level is one launch that writes the next VERSION of the seen rows (three rotating buffers), and the
    # caller may cap the levels that can still push (``push_levels``: beyond them the device pulls whatever the frontier).
    # Neither changes a value: every cap x every direction switch against the two-step form; status[2] = first pull level
    firsts = set()
    for alpha in (0, 1, 8, 256, 1 << 30):
        for cap in (-1, 1, 2, 3, 5):
            w, st = ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=32, want_status=True,
                                             pull_alpha=alpha, push_levels=cap)
            assert torch.equal(w, two_step), (alpha, cap)
            last, more, first_pull, _ = st.tolist()
            assert more == 0 and last == int(out[0][out[0] != 255].max())
            if alpha == 0:
                assert first_pull == 0                                    # never pulls: the caller's choice wins over the cap
            elif cap > 0:
                assert 2 <= first_pull <= cap + 1
            firsts.add(first_pull)
    assert len(firsts) >= 2                                               # pushed-only and pulled searches were both among them
    A = sp.csr_matrix((np.ones(len(col), dtype=np.int8), col.astype(np.int64) - 1, rowptr[1:] - rowptr[1]), shape=(n, n))
    ref = shortest_path(A, method='D', unweighted=True, indices=src[:40].astype(np.int64) - 1)
    got = out[0][:40, 1:].cpu().numpy().astype(np.float64)
    got[got == 255] = np.inf
    assert np.array_equal(got, ref)
    assert (got[0][:n - 1] == np.inf).all() and got[0][n - 1] == 0


@pytest.mark.parametrize('bfs_alpha', [0, 256, 1 << 30], indirect=True)
def test_bfs_hops_matches_apsp(golden, bfs_alpha):
    """Sparse form == dense form on the columns of the chosen sources (the graph is connected
    enough; unreachable pairs are 0 in both conventions); pushed, mixed and pulled expansion."""
    ops = _ops()
    G, dg = _graphs(golden)
    rng = np.random.default_rng(2)
    src = rng.choice(np.array(G.node_order), 70, replace=True).astype(np.int32)       # > 64: two words
    dist = ops.bfs_hops(dg, torch.from_numpy(src).to(DEV), max_hops=32, pull_alpha=bfs_alpha)
    apsp = golden['apsp']
    d = dist.cpu().numpy().astype(np.int64)
    for i, s in enumerate(src):
        row = d[i, 1:]
        ref = apsp[s - 1, :]
        reach = row != 255
        assert np.array_equal(row[reach], ref[reach].astype(np.int64))
        assert (ref[~reach] == 0).all()
    cc = torch.from_numpy(golden['g2_cc_ids_train']).to(DEV)
    S, C, L = cc.shape
    sets = ops.Ragged.from_padded(cc.view(S * C, L))
    got = ops.min_hops_to_sets(dist, sets).cpu().numpy()
    ref = golden['g4_np_sim_train'].reshape(S * C, -1)[:, src - 1]
    assert np.array_equal(got, ref)
    dist_t = ops.bfs_hops(dg, torch.from_numpy(src).to(DEV), max_hops=32, node_major=True, pull_alpha=bfs_alpha)      # (ids, sources)
    assert torch.equal(dist_t.t().contiguous(), dist)
    assert np.array_equal(ops.min_hops_to_sets(dist_t, sets, node_major=True).cpu().numpy(), ref)
    # fused form: BFS + min over members in one call, no hop table
    assert np.array_equal(ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=32, pull_alpha=bfs_alpha).cpu().numpy(), ref)
    # status: the last productive level == the largest finite hop count from these sources; enqueueing exactly that
    # many levels is reported as possibly incomplete (the last level found something), one more is complete
    _, st = ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=32, want_status=True, pull_alpha=bfs_alpha)
    depth = int(d[d != 255].max())
    assert st.tolist()[:2] == [depth, 0]
    w1, st1 = ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=depth, want_status=True)
    assert st1.tolist()[:2] == [depth, 1] and np.array_equal(w1.cpu().numpy(), ref)
    w2, st2 = ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=depth + 1, want_status=True)
    assert st2.tolist()[:2] == [depth, 0] and np.array_equal(w2.cpu().numpy(), ref)
    _, st3 = ops.bfs_min_hops_to_sets(dg, torch.from_numpy(src).to(DEV), sets, max_hops=depth - 1, want_status=True)
    assert st3.tolist()[:2] == [depth - 1, 1]


# ---- a11 DTW ------------------------------------------------------------------------------

def test_dtw_golden(golden):
    """PROVISIONAL pin (restated fastdtw): HIP == golden produced through the restatement."""
    ops = _ops()
    G, dg = _graphs(golden)
    sa = torch.from_numpy(golden['g5_structure_anchors']).to(DEV)
    cc = torch.from_numpy(golden['g2_cc_ids_train']).to(DEV)
    S, C, L = cc.shape
    ra, rc = ops.Ragged.from_padded(sa), ops.Ragged.from_padded(cc.view(S * C, L))
    ai, ae = ops.degree_sequence(dg, ra)
    ci, ce = ops.degree_sequence(dg, rc)
    for xa, xc, key in ((ai, ci, 'g7_int_struc_sim_train'), (ae, ce, 'g7_bor_struc_sim_train')):
        got = ops.dtw_similarity(rc.ptr, xc, rc.max_len, ra.ptr, xa, ra.max_len, 0).view(S, C, -1)      # (g7: rule 0)
        assert np.array_equal(got.cpu().numpy(), golden[key])
    if golden.name in ('tiny', 'density'):
        # the same boundary under rules 1 and 2 (tests/golden/ties.npz); no rule given = config.DTW_TIE_ORDER = 2
        import os
        from conftest import GOLDEN_DIR
        from subgnn_amd import config
        ties = np.load(os.path.join(GOLDEN_DIR, 'ties.npz'))
        assert config.DTW_TIE_ORDER == 2
        for xa, xc, key in ((ai, ci, 'int'), (ae, ce, 'bor')):
            for tie in (1, 2, None):
                got = ops.dtw_similarity(rc.ptr, xc, rc.max_len, ra.ptr, xa, ra.max_len, tie).view(S, C, -1)
                assert np.array_equal(got.cpu().numpy(), ties['%s/g7_tie%d_%s_struc_sim_train' % (golden.name, tie or 2, key)])


@pytest.mark.parametrize('tie', [0, 1, 2])
def test_dtw_random(tie):
    ops = _ops()
    rng = np.random.default_rng(10 + tie)
    xs = [sorted(rng.integers(0, 12, int(rng.integers(0, 70))).tolist()) for _ in range(120)]
    ys = [sorted(rng.integers(0, 40, int(rng.integers(1, 51))).tolist()) for _ in range(37)]
    xp, xv = cbind.ragged(xs)
    yp, yv = cbind.ragged(ys)
    ref = cbind.fastdtw_sim(xp, xv, yp, yv, tie)
    # spot-check the C oracle itself against the pure-Python restatement
    for i in range(0, 120, 17):
        for j in range(0, 37, 5):
            assert ref[i, j] == (np.float32(FD.calc_dtw(xs[i], ys[j], tie)) if len(xs[i]) else 0)
    t = lambda a: torch.from_numpy(a).to(DEV)
    got = ops.dtw_similarity(t(xp), t(xv), 70, t(yp), t(yv), 51, tie).cpu().numpy()
    assert np.array_equal(got, ref)
    # size-independent properties: similarity in (0, 1], identical sequences -> 1
    assert (got[[len(x) > 0 for x in xs]] > 0).all() and (got <= 1).all()
    same = ops.dtw_similarity(t(yp), t(yv), 51, t(yp), t(yv), 51, tie).cpu().numpy()
    assert (np.diag(same) == 1).all()


@pytest.mark.parametrize('tie', [0, 1, 2])
def test_dtw_register_kernel_equals_general_kernel(tie):
    """x rows of <= 32 entries take the register-resident column-major kernel; it must agree bit for
    bit with the C oracle and with the general kernel (``kernel=1``)."""
    ops = _ops()
    rng = np.random.default_rng(50 + tie)
    xs = [sorted(rng.integers(0, 30, int(rng.integers(0, 33))).tolist()) for _ in range(300)]
    xs[0], xs[1], xs[2] = [5], [1, 2], list(range(32))
    ys = [sorted(rng.integers(0, 200, int(rng.integers(1, 61))).tolist()) for _ in range(41)]
    ys[0], ys[1] = [7], [3, 3]
    xp, xv = cbind.ragged(xs)
    yp, yv = cbind.ragged(ys)
    ref = cbind.fastdtw_sim(xp, xv, yp, yv, tie)
    t = lambda a: torch.from_numpy(a).to(DEV)
    fast = ops.dtw_similarity(t(xp), t(xv), 32, t(yp), t(yv), 60, tie).cpu().numpy()
    general = ops.dtw_similarity(t(xp), t(xv), 32, t(yp), t(yv), 60, tie, kernel=1).cpu().numpy()
    assert np.array_equal(fast, ref)
    assert np.array_equal(general, ref)


def test_dtw_row_dedupe_changes_nothing():
    """Identical x rows are computed once and gathered back: same matrix as the plain call."""
    ops = _ops()
    rng = np.random.default_rng(77)
    base = [sorted(rng.integers(0, 6, int(rng.integers(0, 21))).tolist()) for _ in range(40)]
    xs = [base[int(i)] for i in rng.integers(0, 40, 3000)]
    ys = [sorted(rng.integers(0, 50, int(rng.integers(1, 51))).tolist()) for _ in range(23)]
    xp, xv = cbind.ragged(xs)
    yp, yv = cbind.ragged(ys)
    t = lambda a: torch.from_numpy(a).to(DEV)
    a = ops.dtw_similarity(t(xp), t(xv), 20, t(yp), t(yv), 50, 0, dedupe=True)
    b = ops.dtw_similarity(t(xp), t(xv), 20, t(yp), t(yv), 50, 0, dedupe=False, order_rows=False)
    assert torch.equal(a, b)
    ref = cbind.fastdtw_sim(xp, xv, yp, yv, 0)
    assert np.array_equal(a.cpu().numpy(), ref)


@pytest.mark.parametrize('max_len', [1, 9, 16, 20, 32, 47, 64, 65, 300, 1024])
def test_sort_sets_matches_stable_sort(max_len):
    """sgnn_sort_sets: every set ascending, equal ids in their original order (out_pos carries the
    payload), empty sets untouched; the three group widths and the multi-pass wave path."""
    ops = _ops()
    rng = np.random.default_rng(max_len)
    n_sets = 700
    lens = rng.integers(0, max_len + 1, n_sets)
    lens[:3] = (0, max_len, max_len)
    lists = [rng.integers(1, max(4, max_len // 2) + 1, l).tolist() for l in lens]       # many repeats
    r = ops.Ragged.from_lists(lists, DEV)
    r._max_len = max_len
    payload = torch.arange(int(lens.sum()), device=DEV, dtype=torch.int32)
    out, pay = ops.sort_ragged(r, payload)
    got = out.to_lists()
    pay = pay.cpu().numpy()
    off = 0
    for l, g in zip(lists, got):
        order = np.argsort(np.array(l, dtype=np.int64), kind='stable')
        assert g == [l[i] for i in order]
        assert np.array_equal(pay[off:off + len(l)], off + order)
        off += len(l)
    # the device-wide key sort (sets of unknown size) gives the same sets
    r2 = ops.Ragged.from_lists(lists, DEV)
    r2._max_len = None
    assert ops.sort_ragged(r2).to_lists() == got


def test_sort_sets_rejects_large_sets():
    import ctypes
    from subgnn_amd import _lib
    lib = _lib.load()
    ptr = torch.tensor([0, 2000], dtype=torch.int64, device=DEV)
    nodes = torch.arange(2000, dtype=torch.int32, device=DEV)
    out = torch.empty_like(nodes)
    rc = lib.sgnn_sort_sets(ctypes.c_void_p(ptr.data_ptr()), ctypes.c_void_p(nodes.data_ptr()), 1, 2000,
                            ctypes.c_void_p(out.data_ptr()), None, None)
    assert rc == -2


def test_dtw_row_grouping_survives_hash_collisions(monkeypatch):
    """The row grouping of the DTW dedupe trusts no hash: with every row hashing to the same value
    (coefficients forced to zero) rows that differ from their group's representative represent
    themselves, and the similarities are still the plain ones."""
    ops = _ops()
    rng = np.random.default_rng(5)
    base = [sorted(rng.integers(0, 6, int(rng.integers(0, 21))).tolist()) for _ in range(25)]
    xs = [base[int(i)] for i in rng.integers(0, 25, 1500)]
    ys = [sorted(rng.integers(0, 50, int(rng.integers(1, 51))).tolist()) for _ in range(9)]
    xp, xv = cbind.ragged(xs)
    yp, yv = cbind.ragged(ys)
    t = lambda a: torch.from_numpy(a).to(DEV)
    plain = ops.dtw_similarity(t(xp), t(xv), 20, t(yp), t(yv), 50, 0, dedupe=False, order_rows=False)
    monkeypatch.setattr(ops, '_HASH_COEF', {(20, torch.device(DEV)): torch.zeros(20, dtype=torch.int64, device=DEV)})
    rows = ops.Ragged(t(xp), t(xv), max_len=20).to_padded(width=20, fill=-1, dtype=torch.int32)
    rep = ops._row_representatives(rows)
    assert bool((rows[rep] == rows).all()) and bool((rep[rep] == rep).all())
    assert torch.equal(ops.dtw_similarity(t(xp), t(xv), 20, t(yp), t(yv), 50, 0, dedupe=True), plain)
    monkeypatch.setattr(ops, '_HASH_COEF', {})
    rep2 = ops._row_representatives(rows)                          # the real hash: one representative per distinct row
    assert int(rep2.unique().numel()) == len({tuple(x) for x in xs})


def test_ragged_packing_without_host_round_trips():
    """Ragged.from_mask / to_padded (scatter-pack and gather forms) against plain python lists,
    including empty rows, empty matrices and rows of full width."""
    ops = _ops()
    rng = np.random.default_rng(11)
    ids = torch.from_numpy(rng.integers(1, 1000, (300, 17))).to(DEV)
    mask = torch.from_numpy(rng.random((300, 17)) < 0.4).to(DEV)
    mask[0] = False
    mask[1] = True
    r = ops.Ragged.from_mask(ids, mask)
    want = [[int(v) for v, m in zip(row, mr) if m] for row, mr in zip(ids.cpu().tolist(), mask.cpu().tolist())]
    assert r.to_lists() == want
    back = r.to_padded(width=17, fill=-1)
    for row, w in zip(back.cpu().tolist(), want):
        assert row == w + [-1] * (17 - len(w))
    padded = ids * mask                                                      # PAD = 0 where dropped
    assert ops.Ragged.from_padded(padded).to_lists() == [[v for v in row if v != 0] for row in padded.cpu().tolist()]
    assert ops.Ragged.from_padded(torch.zeros((4, 5), dtype=torch.int64, device=DEV)).to_lists() == [[], [], [], []]
    wide = torch.from_numpy(rng.integers(0, 3, (20, 300))).to(DEV)           # wider than the triangular-product form
    assert ops.Ragged.from_padded(wide).to_lists() == [[v for v in row if v != 0] for row in wide.cpu().tolist()]


def test_patch_node_views_both_forms(monkeypatch):
    """First-occurrence unique ids per patch: the L x L comparison form and the sort-based form of
    anchor_patch_samplers.patch_node_views agree with python."""
    from subgnn_amd import anchor_patch_samplers as aps
    rng = np.random.default_rng(2)
    ids = torch.from_numpy(rng.integers(0, 12, (40, 30))).to(DEV)            # many repeats and PADs
    want = [list(dict.fromkeys(v for v in row if v != 0)) for row in ids.cpu().tolist()]
    assert aps.patch_node_views(ids).to_lists() == want
    monkeypatch.setattr(aps, 'VIEW_PAIRWISE_MAX', 0)                       # force the sort-based form
    assert aps.patch_node_views(ids).to_lists() == want


def test_degree_sequence_multigraph_rows_are_streamed():
    """A CSR with an id twice in a row is not a simple graph: the search form of the degree-sequence
    launch (which would count the repeated neighbour once) must not be taken."""
    ops = _ops()
    n = 700
    rowptr = np.zeros(n + 2, dtype=np.int64)
    hub = list(range(2, n + 1)) + [5, 5]                 # node 1: neighbours 2..n, and 5 two more times
    cols = [hub] + [[1] for _ in range(2, n + 1)]
    cols[5 - 1] = [1, 1, 1]                              # node 5 lists the hub three times (kept symmetric)
    col = np.concatenate([np.array(c, dtype=np.int32) for c in cols])
    rowptr[2:] = np.cumsum([len(c) for c in cols])
    g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), DEV)
    assert not g.simple_rows
    r = ops.Ragged.from_lists([[1, 5, 9], [5]], DEV)
    oi, oe = ops.degree_sequence(g, r, sort=False, use_degree_dict=False)
    assert ops.Ragged(r.ptr, oi).to_lists() == [[4, 3, 1], [0]]          # hub: 5 (x3) + 9; node 5: hub x3; node 9: hub


@pytest.mark.parametrize('radius', [1, 2])
def test_ego_graph_structure_patches_golden(tiny, radius):
    """structure_patch_type == 'ego_graph' (aps:226-228) through the product's sampler: the reference's patches
    as sets, in the base graph's node order, equal to the oracle's; border walks over the reference's recorded
    views bit for bit."""
    import json
    import os
    from conftest import GOLDEN_DIR
    from subgnn_amd import anchor_patch_samplers as aps
    ops = _ops()
    z = np.load(os.path.join(GOLDEN_DIR, 'extra.npz'), allow_pickle=False)
    G, dg = _graphs(tiny)
    t = 'ego_r%d_' % radius
    hp = json.loads(str(z[t + 'hparams']))
    hp['seed'] = int(z['seed'])
    want = z[t + 'structure_anchors']
    got = aps.sample_structure_anchor_patches(hp, dg, DEV, hp['max_sim_epochs']).cpu().numpy()
    ref = IH.sample_structure_anchor_patches(G, want.shape[0], hp['sample_walk_len'], hp['rw_beta'], hp['seed'], 'ego_graph', radius)
    assert np.array_equal(got, ref)
    assert [sorted(r.tolist()) for r in got] == [sorted(r.tolist()) for r in want]
    W, Tn = hp['n_triangular_walks'], hp['random_walk_len']
    vb = ops.Ragged.from_lists([[int(v) for v in r if v != -1] for r in z[t + 'views_bor']], DEV)
    inb = ops.Ragged.from_lists([[int(v) for v in r if v != -1] for r in z[t + 'in_border']], DEV)
    n = want.shape[0]
    bw = ops.triangular_walks(dg, 2, n * W, Tn, hp['rw_beta'], hp['seed'], T.stream_id(T.STREAM_WALK_BOR), patches=vb,
                              in_border=inb, walks_per_patch=W)
    assert np.array_equal(bw.view(n, W, Tn).cpu().numpy(), z[t + 'bor_rw'])
    # and the whole structure channel runs on ego patches (degree sequences, DTW, walks from canonical views)
    views = aps.patch_node_views(torch.from_numpy(got).to(DEV))
    iw = aps.perform_random_walks(hp, dg, torch.from_numpy(got).to(DEV), True, views)
    assert tuple(iw.shape) == (n, W, Tn)


def test_degree_sequence_understated_set_bound_is_loud():
    """A caller that promises sets of at most 64 entries and hands over a longer one gets INT32_MIN in that set's
    outputs (not whatever the buffer held); the other sets are unaffected."""
    from subgnn_amd import _lib
    ops = _ops()
    G = _rand_graph(300, 3, 2)
    dg = _dev_graph(G)
    sets = [list(range(1, 21)), list(range(1, 101)), list(range(30, 45))]
    r = ops.Ragged.from_lists(sets, DEV)
    good_i, good_e = ops.degree_sequence(dg, r)                                   # true bound: the block kernel for the long set
    lib = _lib.load()
    oi = torch.full((r.nodes.numel(),), 7, dtype=torch.int32, device=DEV)
    oe = torch.full_like(oi, 7)
    p = ops._ptr
    _lib.check(lib.sgnn_degree_sequence(p(dg.rowptr), p(dg.col), dg.nnz, p(dg.full_degree), p(dg.self_loops), p(r.ptr), p(r.nodes),
                                        r.n, 64, 1, p(oi), p(oe), None, ops._stream()), 'sgnn_degree_sequence')
    assert bool((oi[20:120] == torch.iinfo(torch.int32).min).all()) and bool((oe[20:120] == torch.iinfo(torch.int32).min).all())
    assert torch.equal(oi[:20], good_i[:20]) and torch.equal(oi[120:135], good_i[120:135])


def test_dtw_similarity_kept_row_preparation(golden):
    """x_prep: the grouping of repeated x rows and the processing order are kept by the caller and reused; same values."""
    ops = _ops()
    rng = np.random.default_rng(5)
    base = [sorted(rng.integers(0, 9, size=rng.integers(1, 12)).tolist()) for _ in range(40)]
    xs = [base[i] for i in rng.integers(0, 40, size=3000)]                  # many repeats: the grouping branch
    ys = [sorted(rng.integers(0, 30, size=rng.integers(3, 25)).tolist()) for _ in range(9)]
    X, Y = ops.Ragged.from_lists(xs, DEV), ops.Ragged.from_lists(ys, DEV)
    want = ops.dtw_similarity(X.ptr, X.nodes, 12, Y.ptr, Y.nodes, 25)
    keep = {}
    for _ in range(3):
        got = ops.dtw_similarity(X.ptr, X.nodes, 12, Y.ptr, Y.nodes, 25, x_prep=keep)
        assert torch.equal(got, want)
    assert 'dedupe' in keep and 'order' in keep['grouped']
    keep2 = {}
    for _ in range(2):
        assert torch.equal(ops.dtw_similarity(X.ptr, X.nodes, 12, Y.ptr, Y.nodes, 25, dedupe=False, x_prep=keep2), want)
    assert 'order' in keep2


# ---- padded rows -> ragged sets, node views, in-border filter (the set plumbing every stage goes through) -----------------

@pytest.mark.parametrize('n,L', [(1, 1), (37, 20), (2000, 7), (210, 50), (3, 300)])
def test_ragged_from_padded_and_from_mask_strip_entries_in_order(n, L):
    """sgnn_pack_rows_count / _write against plain python: PAD entries (or masked-out ones) removed, order kept, rows that
    are all PAD give empty sets (gamma.py:27, aps:131, S.py:769)."""
    ops = _ops()
    rng = np.random.default_rng(n * 131 + L)
    ids = rng.integers(0, 9, size=(n, L)).astype(np.int64) * rng.integers(0, 2, size=(n, L))
    ids[0] = 0
    r = ops.Ragged.from_padded(torch.from_numpy(ids).to(DEV))
    assert r.to_lists() == [[int(v) for v in row if v != 0] for row in ids]
    mask = rng.integers(0, 2, size=(n, L)).astype(bool)
    r = ops.Ragged.from_mask(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV))
    assert r.to_lists() == [[int(v) for v, k in zip(row, mk) if k] for row, mk in zip(ids, mask)]


def test_first_occurrence_mask_and_filter_sets():
    """sgnn_first_occurrence_mask = the node view of a walk (unique ids, first occurrence first, PAD dropped: aps:131-138);
    sgnn_filter_sets = the flagged entries of every set in order (the in-border nodes among a patch's view nodes)."""
    ops = _ops()
    rng = np.random.default_rng(5)
    ids = rng.integers(0, 12, size=(300, 40)).astype(np.int64)
    keep = ops.first_occurrence_mask(torch.from_numpy(ids).to(DEV)).cpu().numpy().astype(bool)
    for row, k in zip(ids, keep):
        seen, want = set(), []
        for v in row:
            want.append(v != 0 and v not in seen)
            seen.add(v)
        assert k.tolist() == want
    lists = [rng.integers(1, 1000, size=rng.integers(0, 30)).tolist() for _ in range(500)]
    sets = ops.Ragged.from_lists(lists, DEV)
    flags = [rng.integers(0, 2, size=len(l)).tolist() for l in lists]
    flat = torch.tensor([f for fl in flags for f in fl] + [0] * (sets.nodes.numel() - sum(map(len, lists))), dtype=torch.uint8, device=DEV)
    got = ops.filter_sets(sets, flat)
    assert got.to_lists() == [[v for v, f in zip(l, fl) if f] for l, fl in zip(lists, flags)]



@pytest.mark.parametrize('n,L', [(1, 1), (7, 5), (210, 50), (350, 70), (8192, 3), (1228, 20), (1500, 70), (9000, 20)])
def test_fused_packings_equal_the_reference_packing(n, L):
    """sgnn_pack_rows_fused / sgnn_filter_sets_fused (one launch for up to 8192 rows) against numpy: PAD stripping (gamma.py:27),
    masks, first occurrences (a patch's node view, aps:131-138: rows longer than the 64 mask bits a thread keeps included),
    flagged entries of ragged sets (su:126-144); ptr, the packed entries and the ZEROED tail of the arena.  Inputs beyond 8192 rows
    or 24 576 entries (the LDS staging) take the count / scan / write launches: same results."""
    from subgnn_amd import ops
    rng = np.random.default_rng(n * 131 + L)
    ids = rng.integers(0, 12, size=(n, L)).astype(np.int64)          # many repeats and PADs
    ids[rng.random((n, L)) < 0.2] = 0
    mask = (rng.random((n, L)) < 0.5).astype(np.uint8)
    t = torch.from_numpy(ids).to(DEV)

    def check(r, keep):
        ptr, nodes = r.ptr.cpu().numpy(), r.nodes.cpu().numpy()
        want = [ids[i][keep[i]] for i in range(n)]
        lens = np.array([len(w) for w in want])
        assert np.array_equal(ptr, np.concatenate([[0], np.cumsum(lens)]))
        flat = np.concatenate(want) if lens.sum() else np.zeros(0, dtype=np.int64)
        assert np.array_equal(nodes[:len(flat)], flat)
        assert nodes.shape[0] == n * L + 1 and not nodes[len(flat):].any()
    check(ops.Ragged.from_padded(t), ids != 0)
    check(ops.Ragged.from_mask(t, torch.from_numpy(mask).to(DEV)), mask != 0)
    first = np.zeros((n, L), dtype=bool)
    for i in range(n):
        seen = set()
        for j in range(L):
            v = ids[i, j]
            if v != 0 and v not in seen:
                first[i, j] = True
                seen.add(v)
    check(ops.Ragged.from_first_occurrence(t), first)
    # ragged sets + flags
    sets = ops.Ragged.from_padded(t)
    flags = (rng.random(int(sets.nodes.numel())) < 0.4).astype(np.uint8)
    got = ops.filter_sets(sets, torch.from_numpy(flags).to(DEV))
    p, v = sets.ptr.cpu().numpy(), sets.nodes.cpu().numpy()
    want = [v[p[i]:p[i + 1]][flags[p[i]:p[i + 1]] != 0] for i in range(n)]
    lens = np.array([len(w) for w in want])
    assert np.array_equal(got.ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(lens)]))
    gv = got.nodes.cpu().numpy()
    assert np.array_equal(gv[:lens.sum()], np.concatenate(want) if lens.sum() else np.zeros(0, dtype=np.int32))
    assert not gv[lens.sum():].any()
