"""Base-graph container with networkx-identical node order and neighbour order.

Test infrastructure.  Restates what ``nx.read_edgelist`` followed by
``nx.relabel_nodes(G, {n: int(n)+1})`` produce at SubGNN/SubGNN.py:525,555-556, because the
triangular walks depend on ``list(G.nodes())`` and ``list(G.neighbors(v))`` orders
(anchor_patch_samplers.py:35,70,72,79).  Pinned by golden g1.
"""
import numpy as np


class OracleGraph:
    def __init__(self, node_order, adj):
        self.node_order = list(node_order)          # ids (1-based) in G.nodes() order
        self.adj = adj                              # id -> list of neighbour ids, nx order
        self.n = len(self.node_order)
        self.pos = {v: i for i, v in enumerate(self.node_order)}
        self._adjset = {v: set(a) for v, a in adj.items()}

    def neighbors(self, v):
        return self.adj[v]

    def has_edge(self, u, v):
        return v in self._adjset[u]

    def degree(self, v):
        """networkx degree: a self loop counts twice."""
        a = self.adj[v]
        return len(a) + (1 if v in self._adjset[v] else 0)

    def max_id(self):
        return max(self.node_order)

    def csr(self, sort=False):
        """rowptr int64[max_id+2], col int32 -- row v = neighbours of node id v (row 0 = PAD, empty)."""
        m = self.max_id()
        rowptr = np.zeros(m + 2, dtype=np.int64)
        for v, a in self.adj.items():
            rowptr[v + 1] = len(a)
        rowptr = np.cumsum(rowptr)
        col = np.zeros(int(rowptr[-1]), dtype=np.int32)
        for v, a in self.adj.items():
            aa = sorted(a) if sort else a
            col[rowptr[v]:rowptr[v + 1]] = aa
        return rowptr, col


def _add_edge(order, adj, u, v):
    for w in (u, v):
        if w not in adj:
            adj[w] = {}
            order.append(w)
    adj[u][v] = None
    adj[v][u] = None


def from_edge_pairs(pairs, relabel_plus_one=True):
    """pairs: iterable of (u, v) ints in file order (0-based ids as written by
    nx.write_edgelist, prepare_dataset.py:822)."""
    order, adj = [], {}
    for u, v in pairs:
        _add_edge(order, adj, u, v)
    if not relabel_plus_one:
        return OracleGraph(order, {k: list(d) for k, d in adj.items()})
    # nx.relabel_nodes(copy=True): nodes added in G order, then edges in G.edges order
    order2 = [n + 1 for n in order]
    adj2 = {n: {} for n in order2}
    seen = set()
    for n in order:
        for nbr in adj[n]:
            if nbr not in seen:
                a, b = n + 1, nbr + 1
                adj2[a][b] = None
                adj2[b][a] = None
        seen.add(n)
    return OracleGraph(order2, {k: list(d) for k, d in adj2.items()})


def read_edgelist(path, relabel_plus_one=True):
    pairs = []
    with open(path) as f:
        for line in f:
            line = line.split('#')[0].strip()
            if not line:
                continue
            s = line.split()
            pairs.append((int(s[0]), int(s[1])))
    return from_edge_pairs(pairs, relabel_plus_one)


class _IdentityPos:
    """pos[v] = v - 1 for a graph whose node order is 1..n."""

    def __getitem__(self, v):
        return int(v) - 1


class CSRGraph:
    """Same interface as OracleGraph, backed by CSR arrays (rows sorted ascending, node order
    1..n): for graphs too large for python dicts (the 1M-node benchmark graph)."""

    def __init__(self, rowptr, col):
        self.rowptr, self.col = rowptr, col
        self.n = len(rowptr) - 2
        self.node_order = range(1, self.n + 1)
        self.pos = _IdentityPos()

    def neighbors(self, v):
        return self.col[self.rowptr[v]:self.rowptr[v + 1]]

    def has_edge(self, u, v):
        a = self.neighbors(u)
        i = int(np.searchsorted(a, v))
        return i < len(a) and a[i] == v

    def degree(self, v):
        a = self.neighbors(v)
        return len(a) + (1 if self.has_edge(v, v) else 0)

    def max_id(self):
        return self.n
