"""Base-graph loading for the hot path: edge list -> CSR resident in HBM.

The walks depend on ``list(G.nodes())`` and ``list(G.neighbors(v))`` orders of the graph the
reference builds with ``nx.read_edgelist`` + ``nx.relabel_nodes(G, {n: int(n)+1})``
(reference SubGNN/SubGNN.py:525,555-556), so the CSR reproduces both, computed here with
vectorised numpy (a 10M-edge list takes seconds) instead of a python dict simulation:

  * node order  = order of first appearance in the file;
  * after the relabel copy, the neighbours of v are: first the neighbours that precede v in
    node order, sorted by their position; then the remaining ones (v itself for a self
    loop included) in the order their edge first appeared in the file.
"""
import json

import numpy as np

from .ops import DeviceGraph


def parse_edge_list(path):
    """-> int64 array (E, 2) of the ids as written (0-based, nx.write_edgelist(data=False))."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.split('#', 1)[0].split()
            if len(line) >= 2:
                rows.append((int(line[0]), int(line[1])))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 2)


def networkx_order_csr(edges):
    """edges: (E,2) 0-based ids in file order.  Returns (rowptr int64[max_id+2], col int32[nnz],
    node_order int32[n]) for the graph relabelled to 1-based ids."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if edges.size == 0:
        return np.zeros(2, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    flat = edges.reshape(-1)
    ids, first = np.unique(flat, return_index=True)
    node_order0 = ids[np.argsort(first, kind='stable')]              # first appearance
    max_id = int(ids.max()) + 1                                       # after the +1 relabel
    pos = np.full(max_id + 1, -1, dtype=np.int64)
    pos[node_order0 + 1] = np.arange(len(node_order0))
    # both directions, time-stamped by edge number; keep the first occurrence of each (a, b)
    a = np.concatenate([edges[:, 0], edges[:, 1]]) + 1
    b = np.concatenate([edges[:, 1], edges[:, 0]]) + 1
    t = np.concatenate([np.arange(len(edges)), np.arange(len(edges))])
    key = a * (max_id + 1) + b
    order = np.lexsort((t, key))
    key_s = key[order]
    keep = np.ones(len(order), dtype=bool)
    keep[1:] = key_s[1:] != key_s[:-1]
    sel = order[keep]
    a, b, t = a[sel], b[sel], t[sel]
    earlier = pos[b] < pos[a]
    k1 = np.where(earlier, 0, 1)
    k2 = np.where(earlier, pos[b], t)
    o = np.lexsort((k2, k1, a))
    a, b = a[o], b[o]
    rowptr = np.zeros(max_id + 2, dtype=np.int64)
    np.add.at(rowptr, a + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, b.astype(np.int32), (node_order0 + 1).astype(np.int32)


def load_graph(edge_list_path, device, degree_dict_path=None):
    """reference read_data (SubGNN/SubGNN.py:524-525,554-556) -> DeviceGraph.  If a
    ``degree_sequence.txt`` json exists (0-based string keys, precompute_graph_metrics.py:59) it
    supplies the full degrees used by the border structure channel (gamma.py:44-45)."""
    edges = parse_edge_list(edge_list_path)
    rowptr, col, node_order = networkx_order_csr(edges)
    full = None
    if degree_dict_path is not None:
        try:
            with open(str(degree_dict_path)) as f:
                dd = json.load(f)
            full = np.zeros(len(rowptr) - 1, dtype=np.int32)
            for k, v in dd.items():
                if int(k) + 1 < len(full):
                    full[int(k) + 1] = int(v)
        except FileNotFoundError:
            full = None
    return DeviceGraph(rowptr, col, node_order, device, full_degree=full)
