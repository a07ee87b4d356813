"""Synthetic DENSITY-style inputs for benchmarks and smoke tests (host side, numpy).

The reference builds its synthetic datasets with networkx (prepare_dataset/prepare_dataset.py:
26-327: Barabasi-Albert base graph, BFS subgraphs of N_SUBGRAPH_NODES nodes, recipe in
prepare_dataset/config_prepare_dataset.py:15-31).  networkx needs minutes for a 1M-node /
10M-edge graph, so this module generates the same kind of input with vectorised numpy:
preferential attachment by sampling the running endpoint list (the classic BA construction),
resolved for all nodes at once by pointer jumping.
"""
import numpy as np


def barabasi_albert_edges(n, m, seed):
    """(E, 2) int64 0-based edge list, node t >= m attaches to m endpoints drawn from the
    endpoint list of all earlier edges (multi-edges collapse later in the CSR builder)."""
    rng = np.random.default_rng(seed)
    # seed star: nodes 0..m-1 all linked to node m (as nx.barabasi_albert_graph starts)
    n_new = n - m
    # endpoint list layout: first 2*m entries for the initial star, then per new node t (>m):
    # m pairs (t, target)
    src = np.repeat(np.arange(m + 1, n, dtype=np.int64), m)
    k = len(src)
    # position of each draw in the endpoint list: uniform over the prefix available to node t
    t_idx = (src - (m + 1))
    prefix = 2 * m + 2 * m * t_idx
    pos = (rng.random(k) * prefix).astype(np.int64)
    tgt = np.full(k, -1, dtype=np.int64)
    init = np.empty(2 * m, dtype=np.int64)
    init[0::2] = np.arange(m)
    init[1::2] = m
    # resolve: pos < 2m -> initial list; else entry e = pos - 2m: even -> src[e//2], odd -> tgt[e//2]
    unresolved = np.arange(k)
    ref = pos.copy()
    for _ in range(64):
        if len(unresolved) == 0:
            break
        r = ref[unresolved]
        is_init = r < 2 * m
        e = r - 2 * m
        is_src = (~is_init) & (e % 2 == 0)
        done_val = np.where(is_init, init[np.clip(r, 0, 2 * m - 1)], src[np.clip(e // 2, 0, k - 1)])
        dep = np.clip(e // 2, 0, k - 1)
        dep_val = tgt[dep]
        can = is_init | is_src | (dep_val >= 0)
        val = np.where(is_init | is_src, done_val, dep_val)
        tgt[unresolved[can]] = val[can]
        unresolved = unresolved[~can]
    assert len(unresolved) == 0
    star = np.stack([np.arange(m, dtype=np.int64), np.full(m, m, dtype=np.int64)], 1)
    edges = np.concatenate([star, np.stack([src, tgt], 1)], 0)
    edges = edges[edges[:, 0] != edges[:, 1]]
    return edges


def sorted_csr(edges, n):
    """Undirected simple-graph CSR for 1-based ids from a 0-based edge list: rowptr int64[n+2],
    col int32 (rows ascending).  With ascending ids and ascending rows the networkx node order
    is 1..n; the networkx neighbour order would differ, which only the walks care about -- the
    synthetic benchmark defines its graph BY this CSR."""
    a = np.concatenate([edges[:, 0], edges[:, 1]]) + 1
    b = np.concatenate([edges[:, 1], edges[:, 0]]) + 1
    key = np.unique(a * (n + 2) + b)
    a, b = key // (n + 2), key % (n + 2)
    rowptr = np.zeros(n + 2, dtype=np.int64)
    np.add.at(rowptr, a + 1, 1)
    return np.cumsum(rowptr), b.astype(np.int32)


def bfs_subgraphs(rowptr, col, n_subgraphs, n_nodes_each, seed):
    """DENSITY-style subgraphs: breadth-first from a random start, first ``n_nodes_each`` nodes
    (prepare_dataset.py:288-327).  Returns a list of python lists of 1-based ids."""
    rng = np.random.default_rng(seed)
    n = len(rowptr) - 2
    out = []
    starts = rng.integers(1, n + 1, n_subgraphs)
    for s in starts:
        seen = [int(s)]
        have = {int(s)}
        qi = 0
        while len(seen) < n_nodes_each and qi < len(seen):
            v = seen[qi]
            qi += 1
            for w in col[rowptr[v]:rowptr[v + 1]][:n_nodes_each]:
                w = int(w)
                if w not in have:
                    have.add(w)
                    seen.append(w)
                    if len(seen) == n_nodes_each:
                        break
        out.append(seen)
    return out


def barabasi_albert_csr_device(n, m, seed, device):
    """The same preferential-attachment construction as ``barabasi_albert_edges`` + ``sorted_csr``,
    carried out with torch on the GPU (a graph whose CSR exceeds the 256 MiB Infinity Cache --
    n = 8M, m = 16: 0.25 G directed entries -- takes minutes and tens of GB in numpy).  Returns
    (rowptr int64[n+2], col int32[nnz]) on ``device``, rows ascending, ids 1-based.  Its own random
    stream (torch's generator): the graph is defined by (n, m, seed) for THIS function."""
    import torch
    g = torch.Generator(device=device).manual_seed(int(seed))
    src = torch.arange(m + 1, n, dtype=torch.int64, device=device).repeat_interleave(m)
    k = src.numel()
    prefix = 2 * m + 2 * m * (src - (m + 1))
    pos = (torch.rand(k, generator=g, dtype=torch.float64, device=device) * prefix).to(torch.int64)
    del prefix
    # entry r of the endpoint list: r < 2m -> the initial star; else e = r - 2m: even -> src[e // 2],
    # odd -> the target of draw e // 2, i.e. follow that draw's own reference (pointer jumping)
    ref = pos.clone()
    for _ in range(200):
        follow = (ref >= 2 * m) & (((ref - 2 * m) & 1) == 1)
        if not bool(follow.any()):
            break
        ref = torch.where(follow, pos[((ref - 2 * m) >> 1).clamp_(min=0)], ref)
    else:
        raise RuntimeError('pointer jumping did not converge')
    init = torch.empty(2 * m, dtype=torch.int64, device=device)
    init[0::2] = torch.arange(m, device=device)
    init[1::2] = m
    e = (ref - 2 * m).clamp_(min=0) >> 1
    tgt = torch.where(ref < 2 * m, init[ref.clamp(max=2 * m - 1)], src[e])
    del ref, pos, e
    star_a = torch.arange(m, dtype=torch.int64, device=device)
    a = torch.cat([star_a, src]) + 1
    b = torch.cat([torch.full((m,), m, dtype=torch.int64, device=device), tgt]) + 1
    del src, tgt
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * (n + 2) + b, b * (n + 2) + a]))
    del a, b
    rows = key // (n + 2)
    col = (key - rows * (n + 2)).to(torch.int32)
    del key
    rowptr = torch.zeros(n + 2, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n + 1), 0)
    return rowptr, col
