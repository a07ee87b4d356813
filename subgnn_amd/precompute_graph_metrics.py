"""GPU counterpart of the reference's prepare_dataset/precompute_graph_metrics.py (which needs
snap-stanford): all-pairs hop counts, degree dict and 1-hop ego graphs of the base graph, in the
on-disk formats SubGNN reads (SubGNN.py:718-722,847-848,882-886):

  shortest_path_matrix.npy  float64 (N, N), entry [s, t] = hop count, 0 on the diagonal and for
                            unreachable pairs (precompute_graph_metrics.py:20-25,66-70)
  degree_sequence.txt       json {str(0-based id): degree}          (:47-59)
  ego_graphs.txt            json {str(0-based id): [0-based 1-hop neighbour ids]}   (:31-45)

The all-pairs part reuses the bit-parallel multi-source BFS kernel (sgnn_bfs_hops), 64 sources per
machine word and ``chunk`` sources per launch sequence.  Only sensible where N x N float64 fits in
memory -- the dense matrix is the reference's design; the large-graph path never builds it
(hotpath.prepare_sparse).
"""
import json
from pathlib import Path

import numpy as np
import torch

from . import ops
from .graph import load_graph


def all_pairs_hops(graph, chunk=1024, max_hops=254):
    """(N, N) float64 numpy matrix indexed by 0-based id."""
    n = graph.max_id
    out = np.zeros((n, n), dtype=np.float64)
    for s0 in range(1, n + 1, chunk):
        src = torch.arange(s0, min(s0 + chunk, n + 1), dtype=torch.int32, device=graph.device)
        d = ops.bfs_hops(graph, src, max_hops=max_hops)[:, 1:]
        d = torch.where(d == 255, torch.zeros_like(d), d)
        out[s0 - 1:s0 - 1 + src.numel()] = d.cpu().numpy().astype(np.float64)
    return out


def degree_dict(graph):
    rp = graph.rowptr.cpu().numpy()
    col = graph.col.cpu().numpy()
    deg = np.diff(rp)[1:]
    ids = np.repeat(np.arange(1, graph.max_id + 1), deg)
    self_loops = np.bincount(ids[col[:len(ids)] == ids], minlength=graph.max_id + 1)[1:]
    return {str(i): int(deg[i] + self_loops[i]) for i in range(graph.max_id)}


def ego_graphs(graph):
    rp = graph.rowptr.cpu().numpy()
    col = graph.col.cpu().numpy()
    return {str(v - 1): [int(w) - 1 for w in col[rp[v]:rp[v + 1]]] for v in range(1, graph.max_id + 1)}


def calculate_stats(dataset_dir, device=None, shortest_paths=True, degree_sequence=True, ego=True, override=False):
    """Writes the three files next to ``edge_list.txt`` (skipping existing ones unless override)."""
    d = Path(dataset_dir)
    device = device or torch.device('cuda')
    g = load_graph(d / 'edge_list.txt', device)
    (d / 'similarities').mkdir(exist_ok=True)
    if ego and (override or not (d / 'ego_graphs.txt').exists()):
        with open(d / 'ego_graphs.txt', 'w') as f:
            json.dump(ego_graphs(g), f)
    if degree_sequence and (override or not (d / 'degree_sequence.txt').exists()):
        with open(d / 'degree_sequence.txt', 'w') as f:
            json.dump(degree_dict(g), f)
    if shortest_paths and (override or not (d / 'shortest_path_matrix.npy').exists()):
        np.save(d / 'shortest_path_matrix.npy', all_pairs_hops(g))
    return g
