"""Integer / graph half of the SubGNN hot path, restated in plain Python + numpy.

Test infrastructure only (see oracle/__init__.py).  Each function cites the reference
file:line it follows.  Randomness comes from the draw tape (oracle/tape.py).

Conventions: node ids are 1-based, 0 = PAD (config.py:9, SubGNN.py:554-559).
Ragged containers are python lists; padded views are produced only by the ``pad_*``
helpers at the edge, exactly where the reference pads.
"""
import numpy as np

from . import tape as T
from . import fastdtw_restate as FD

PAD = 0


# ---------------------------------------------------------------------------------------
# a7  connected components  (SubGNN.py:575-607)
# ---------------------------------------------------------------------------------------

def connected_components(G, nodes):
    """nx.connected_components of the induced subgraph (SubGNN.py:590-591).

    Canonical order of THIS restatement (the reference's is CPython-set order,
    SURVEY.md Appendix A.4, compared as sets of sets): components in order of their first
    node in ``nodes``; nodes inside a component in ``nodes`` order.  Duplicates collapse.
    """
    uniq = list(dict.fromkeys(int(v) for v in nodes))
    inset = set(uniq)
    label = {}
    comps = []
    for s in uniq:
        if s in label:
            continue
        cid = len(comps)
        label[s] = cid
        stack = [s]
        while stack:
            v = stack.pop()
            for w in G.neighbors(v):
                if w in inset and w not in label:
                    label[w] = cid
                    stack.append(w)
        comps.append(None)
    comps = [[] for _ in comps]
    for v in uniq:
        comps[label[v]].append(v)
    return comps


def pad_cc_ids(cc_lists):
    """SubGNN.py:594-605: pad #CC with [PAD] rows, then pad lengths -> (S, C, L) int64."""
    S = len(cc_lists)
    C = max(len(c) for c in cc_lists)
    L = max(max((len(cc) for cc in c), default=1) for c in cc_lists)
    L = max(L, 1)
    out = np.zeros((S, C, L), dtype=np.int64)
    for s, comps in enumerate(cc_lists):
        for c, cc in enumerate(comps):
            out[s, c, :len(cc)] = cc
    return out


# ---------------------------------------------------------------------------------------
# a8  k-hop border set of a component  (subgraph_utils.py:146-176, SubGNN.py:673-700)
# ---------------------------------------------------------------------------------------

def component_border_set(G, component, k, ego_dict_mode=False):
    """Returns the border as a python set.

    ego_dict_mode=False: union of nx.ego_graph(G, v, radius=k) over v in CC, minus the CC
        (su:165-166,174).
    ego_dict_mode=True : ``ego_graphs.txt`` present.  The dict maps the 0-based id to the
        list of *0-based* 1-hop neighbour ids (precompute_graph_metrics.py:36-42); only the
        key is shifted at su:168, so the union holds (true id - 1) values which are then
        differenced against the 1-based CC (su:174).  k is ignored.  Id 0 (= PAD) can appear.
    """
    cc = {int(v) for v in component if int(v) != PAD}
    if ego_dict_mode:
        hood = set()
        for v in cc:
            hood.update(w - 1 for w in G.neighbors(v))
        return hood - cc
    dist = {v: 0 for v in cc}
    frontier = list(cc)
    for h in range(k):
        nxt = []
        for v in frontier:
            for w in G.neighbors(v):
                if w not in dist:
                    dist[w] = h + 1
                    nxt.append(w)
        frontier = nxt
    return {v for v, d in dist.items() if d > 0}


def border_hop_levels(G, component, k):
    """hop distance (1..k) of every k-hop border node -- what the sparse N-border
    similarity needs (equals the APSP row-min of a9 on those columns)."""
    cc = {int(v) for v in component if int(v) != PAD}
    dist = {v: 0 for v in cc}
    frontier = list(cc)
    for h in range(k):
        nxt = []
        for v in frontier:
            for w in G.neighbors(v):
                if w not in dist:
                    dist[w] = h + 1
                    nxt.append(w)
        frontier = nxt
    return {v: d for v, d in dist.items() if d > 0}


def pad_border_sets(border_sets):
    """SubGNN.py:690-696 (S, C, Lb) int64; entries sorted ascending (canonical order of this
    restatement; the reference stores ``list(set)``)."""
    S = len(border_sets)
    C = max(len(b) for b in border_sets)
    Lb = max(max((len(x) for x in b), default=0) for b in border_sets)
    out = np.zeros((S, C, max(Lb, 0)), dtype=np.int64)
    for s, bs in enumerate(border_sets):
        for c, b in enumerate(bs):
            b = sorted(b)
            out[s, c, :len(b)] = b
    return out


# ---------------------------------------------------------------------------------------
# a9  shortest-path similarities  (SubGNN.py:752-781)
# ---------------------------------------------------------------------------------------

def shortest_path_similarities(apsp, cc_ids):
    """sims[s,c,:] = min over v in cc of apsp[v-1,:]; padded CC rows -> PAD (:762-778).
    apsp is the dense float64 (N,N) matrix of precompute_graph_metrics.py:20-25,69."""
    S, C, _ = cc_ids.shape
    N = apsp.shape[1]
    out = np.zeros((S, C, N), dtype=np.float32)
    for s in range(S):
        for c in range(C):
            comp = cc_ids[s, c][cc_ids[s, c] != PAD]
            if len(comp) > 0:
                out[s, c, :] = np.min(apsp[comp - 1, :], axis=0).astype(np.float32)
    mask = cc_ids[:, :, 0] != PAD
    out[~mask] = PAD
    return out


def bfs_all_pairs(G):
    """precompute_graph_metrics.py:20-25: row (id-1) = hop counts from node id, 0 where
    unreachable (and on the diagonal), float64, indexed by 0-based id."""
    m = G.max_id()
    out = np.zeros((m, m), dtype=np.float64)
    for s in G.node_order:
        dist = {s: 0}
        frontier = [s]
        while frontier:
            nxt = []
            for v in frontier:
                for w in G.neighbors(v):
                    if w not in dist:
                        dist[w] = dist[v] + 1
                        nxt.append(w)
            frontier = nxt
        for v, d in dist.items():
            out[s - 1, v - 1] = d
    return out


# ---------------------------------------------------------------------------------------
# a10 degree sequences  (gamma.py:21-49)
# ---------------------------------------------------------------------------------------

def degree_sequence(G, nodes, degree_dict=None, internal=True, sort=True):
    """``nodes`` is the padded id vector.  Duplicates are kept (subgraph.degree(nodes)
    iterates the raw list, gamma.py:27-30,43-45); a self loop counts twice (networkx)."""
    nodes = [int(v) for v in nodes if int(v) != PAD]
    inset = set(nodes)
    internal_seq = []
    for v in nodes:
        d = 0
        for w in G.neighbors(v):
            if w in inset:
                d += 2 if w == v else 1
        internal_seq.append(d)
    if internal:
        seq = internal_seq
    else:
        if degree_dict is None:
            full = [G.degree(v) for v in nodes]
        else:
            full = [degree_dict[v - 1] for v in nodes]
        seq = [f - i for f, i in zip(full, internal_seq)]
    return sorted(seq) if sort else seq


# ---------------------------------------------------------------------------------------
# a11 structure similarities  (gamma.py:51-59, SubGNN.py:783-833)
# ---------------------------------------------------------------------------------------

def structure_similarities(G, cc_ids, structure_anchors, degree_dict, internal, tie_order=None):
    S, C, _ = cc_ids.shape
    P = structure_anchors.shape[0]
    aseq = [degree_sequence(G, structure_anchors[a], degree_dict, internal) for a in range(P)]
    out = np.zeros((S, C, P), dtype=np.float32)
    for s in range(S):
        for c in range(C):
            cseq = degree_sequence(G, cc_ids[s, c], degree_dict, internal)
            if cc_ids[s, c, 0] == PAD:
                continue                                   # SubGNN.py:831
            for a in range(P):
                out[s, c, a] = np.float32(FD.calc_dtw(cseq, aseq[a], tie_order))
    return out


# ---------------------------------------------------------------------------------------
# a1  triangular random walk  (anchor_patch_samplers.py:20-113)
# ---------------------------------------------------------------------------------------

class _Draws:
    def __init__(self, seed, stream, item):
        self.seed, self.stream, self.item, self.j = seed, stream, item, 0

    def choice(self, seq):
        i = T.choice_index(self.seed, self.stream, self.item, self.j, len(seq))
        self.j += 1
        return seq[i]

    def uniform(self):
        u = T.uniform01(self.seed, self.stream, self.item, self.j)
        self.j += 1
        return u


def triangular_walk(G, walk_len, beta, draws, mode, patch_nodes=None, in_border=None):
    """mode 'graph'  : aps:231 -- inside=True with the whole graph as the 'subgraph'
       mode 'inside' : aps:150 inside=True  -- restricted to the patch's induced subgraph;
                       ``patch_nodes`` = list(anchor_patch_subgraph.nodes()) in view order
       mode 'border' : aps:150 inside=False -- start in ``in_border``; neighbours filtered to
                       all_valid = in_border U (V \\ patch) (aps:143); triangles on the full graph
    Returns the visited list (aps:113)."""
    if mode == 'graph':
        node_list = G.node_order
        member = None
    else:
        member = set(patch_nodes)
        node_list = patch_nodes
    if mode == 'border':
        bset = set(in_border)

        def ok(n):
            return (n not in member) or (n in bset)
        if len(in_border) == 0:
            return []           # reference: np.random.choice([]) raises ValueError (aps:78)
        prev = draws.choice(in_border)                                       # aps:78
        nbrs = [n for n in G.neighbors(prev) if ok(n)]                       # aps:79
    else:
        def ok(n):
            return member is None or n in member
        prev = draws.choice(node_list)                                       # aps:70
        nbrs = [n for n in G.neighbors(prev) if ok(n)]                       # aps:72
    if len(nbrs) == 0:
        return [prev]                                                        # aps:83-84
    curr = draws.choice(nbrs)                                                # aps:74,80
    visited = [prev, curr]
    for _ in range(walk_len - 2):
        cand = [n for n in G.neighbors(curr) if ok(n)]                       # aps:35,38
        # is_triangle (aps:20-24): n adjacent to prev (inside the same restricted graph; n is
        # already a valid neighbour of curr, prev is a valid node)
        tri = [n for n in cand if G.has_edge(prev, n)]
        non = [n for n in cand if not G.has_edge(prev, n)]
        if len(cand) == 0:
            break                                                            # aps:94
        if len(tri) == 0:
            nxt = draws.choice(non)                                          # aps:98
        elif len(non) == 0:
            nxt = draws.choice(tri)                                          # aps:100
        elif draws.uniform() <= beta:                                        # aps:102
            nxt = draws.choice(tri)
        else:
            nxt = draws.choice(non)                                          # aps:106
        prev, curr = curr, nxt
        visited.append(nxt)
    return visited


def patch_unique_nodes(patch):
    """Canonical node order of a patch's induced subgraph in this restatement: first
    occurrence in the id list (the reference's is networkx-view / CPython-set order)."""
    return list(dict.fromkeys(int(v) for v in patch if int(v) != PAD))


def patch_in_border_nodes(G, patch_nodes):
    """su:126-144 with its indexing quirk: the dense adjacency has rows in G.nodes() order but
    is indexed by id-1 (su:139), i.e. id x is read as the node at position x-1.  With a
    sorted node order this is the plain 'has an edge leaving the patch' test."""
    member = set(patch_nodes)
    order = G.node_order
    out = []
    for x in patch_nodes:
        px = order[x - 1]
        hit = False
        for w in G.neighbors(px):
            y = G.pos[w] + 1             # the id whose (id-1) indexes column of w
            if y not in member:
                hit = True
                break
        if hit:
            out.append(x)
    return out


def ego_graph_nodes(G, center, radius):
    """list(nx.ego_graph(G, center, radius).nodes) (aps:228): the nodes within ``radius`` hops of the
    centre, centre included, in the order of the base graph's node view (the ego graph is a copy of a
    subgraph VIEW, which iterates G's nodes filtered by membership)."""
    dist = {center: 0}
    frontier = [center]
    for h in range(radius):
        nxt = []
        for v in frontier:
            for w in G.neighbors(v):
                if w not in dist:
                    dist[w] = h + 1
                    nxt.append(w)
        frontier = nxt
    return [v for v in G.node_order if v in dist]


def sample_structure_anchor_patches(G, n_samples, sample_walk_len, beta, seed, patch_type='triangular_random_walk',
                                    radius=1):
    """aps:210-243.  'triangular_random_walk': the start nodes drawn at aps:222 are ignored (the walk
    re-draws its own, aps:70).  'ego_graph': patch i = the ego graph around the i-th start node
    (aps:226-228), one np.random.choice call for all starts (tape item 0, draw i)."""
    patches = []
    if patch_type == 'ego_graph':
        order = list(G.node_order)
        st = T.stream_id(T.STREAM_STRUCT_START)
        for i in range(n_samples):
            patches.append(ego_graph_nodes(G, order[T.choice_index(seed, st, 0, i, len(order))], radius))
    else:
        for i in range(n_samples):
            d = _Draws(seed, T.stream_id(T.STREAM_STRUCT_PATCH), i)
            patches.append(triangular_walk(G, sample_walk_len, beta, d, 'graph'))
    L = max(len(p) for p in patches)
    out = np.zeros((n_samples, L), dtype=np.int64)
    for i, p in enumerate(patches):
        out[i, :len(p)] = p
    return out


def perform_random_walks(G, anchor_patch_ids, n_walks, walk_len, beta, inside, seed,
                         patch_orders=None, in_borders=None):
    """aps:118-158 -> (n_patches, n_walks, walk_len) int64, PAD filled.
    patch_orders / in_borders: optional per-patch overrides carrying the reference's own
    node-view order (goldens); default = this restatement's canonical order."""
    P = anchor_patch_ids.shape[0]
    out = np.zeros((P, n_walks, walk_len), dtype=np.int64)
    kind = T.STREAM_WALK_INT if inside else T.STREAM_WALK_BOR
    for p in range(P):
        nodes = patch_orders[p] if patch_orders is not None else patch_unique_nodes(anchor_patch_ids[p])
        if len(nodes) == 0:
            continue                                                         # aps:134-135
        inb = None
        if not inside:
            inb = in_borders[p] if in_borders is not None else patch_in_border_nodes(G, nodes)
        for w in range(n_walks):
            d = _Draws(seed, T.stream_id(kind), p * n_walks + w)
            walk = triangular_walk(G, walk_len, beta, d, 'inside' if inside else 'border', nodes, inb)
            out[p, w, :len(walk)] = walk
    return out


# ---------------------------------------------------------------------------------------
# a4  neighbourhood anchors  (aps:163-198)
# ---------------------------------------------------------------------------------------

def sample_neighborhood_anchors(id_matrix, n_slots, seed, stream):
    """id_matrix: (S, C, L) padded ids (cc_ids for inside, border sets for border), any order.
    Slot i of row r (tape item r*n_slots+i) takes the tape's neighbourhood-anchor pick
    (oracle/tape.py nanchor_pick) among the row's non-PAD entries in ascending order, or PAD --
    the law of aps:177-179 / 189-191 (argmax of iid symmetric variates with PAD columns at 0)."""
    S, C, L = id_matrix.shape
    rows = id_matrix.reshape(S * C, L)
    out = np.zeros((S * C, n_slots), dtype=np.int64)
    for r in range(S * C):
        real = np.sort(rows[r][rows[r] != PAD])
        n = len(real)
        for i in range(n_slots):
            k = T.nanchor_pick(seed, stream, r * n_slots + i, n, n < L)
            out[r, i] = PAD if k < 0 else real[k]
    return out.reshape(S, C, n_slots)


# ---------------------------------------------------------------------------------------
# a5  position anchors  (aps:200-208, 281-314);  a6 structure anchors (aps:316-328)
# ---------------------------------------------------------------------------------------

def position_anchors_internal(subgraphs, n_in, seed, split, layer):
    st = T.stream_id(T.STREAM_P_INT, split, layer)
    out = np.zeros((len(subgraphs), n_in), dtype=np.int64)
    for s, sg in enumerate(subgraphs):
        for j in range(n_in):
            out[s, j] = sg[T.choice_index(seed, st, s, j, len(sg))]
    return out


def position_anchors_border(G, n_out, seed, layer):
    st = T.stream_id(T.STREAM_P_EXT, 0, layer)
    order = G.node_order
    return np.array([order[T.choice_index(seed, st, 0, j, len(order))] for j in range(n_out)], dtype=np.int64)


def structure_anchor_indices(n_presampled, n_structure, seed, layer):
    st = T.stream_id(T.STREAM_S_PICK, 0, layer)
    return [T.choice_index(seed, st, 0, j, n_presampled) for j in range(n_structure)]


# ---------------------------------------------------------------------------------------
# a17 collate trimming  (SubGNN.py:1098-1099, 1109-1110)
# ---------------------------------------------------------------------------------------

def trim_zero_columns(x):
    B, C, L = x.shape
    flat = x.reshape(B * C, L)
    keep = np.abs(flat).sum(axis=0) != 0
    return flat[:, keep].reshape(B, C, -1)


def bfs_hops_numpy(rowptr, col, src, n):
    """Hop distance from node id ``src`` to every id of a CSR graph (row v = neighbours of id v), uint8, 255 = not
    reached: a level-synchronous numpy BFS for the parity tests of graphs too large for the all-pairs matrix."""
    dist = np.full(n + 1, 255, dtype=np.uint8)
    dist[src] = 0
    frontier = np.array([src], dtype=np.int64)
    level = 0
    while len(frontier) and level < 254:
        level += 1
        starts, ends = rowptr[frontier], rowptr[frontier + 1]
        lens = ends - starts
        idx = np.repeat(starts - np.cumsum(lens) + lens, lens) + np.arange(int(lens.sum()))
        nb = np.unique(col[idx])
        nb = nb[dist[nb] == 255]
        dist[nb] = level
        frontier = nb.astype(np.int64)
    return dist
