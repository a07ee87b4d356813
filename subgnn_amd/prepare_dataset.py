"""Synthetic DENSITY-style dataset directories in the reference's on-disk formats.

The reference generator (prepare_dataset/prepare_dataset.py:26-831, recipe
prepare_dataset/config_prepare_dataset.py:15-41) builds a Barabasi-Albert base graph, BFS
subgraphs of N_SUBGRAPH_NODES nodes, EDITS the graph until the subgraph densities fall into
N_BINS target ranges, and trains GIN/GraphSAINT node embeddings.  This module is the reduced
counterpart the hot path needs to be runnable end to end without networkx/PyG:

  edge_list.txt          "u v" per line, 0-based ids              (prepare_dataset.py:822)
  subgraphs.pth          "n1-n2-...\\tlabel\\tsplit\\t" per line      (prepare_dataset.py:781-799)
  <type>_embeddings.pth  torch.save of an (N, D) float tensor     (train_node_emb.py)
  + the graph-metric files via precompute_graph_metrics.calculate_stats

Differences, on purpose: the base graph is NOT edited -- labels are the density terciles of the BFS
subgraphs as they are; embeddings are random N(0,1) (pre-training is out of scope).  80/10/10 split
(prepare_dataset.py:756-778).
"""
import argparse
from pathlib import Path

import numpy as np
import torch

from . import synthetic


def write_density_dataset(out_dir, n_nodes=1000, m=5, n_subgraphs=250, subgraph_nodes=20, embed_dim=32, seed=42,
                          embedding_type='gin', n_bins=3):
    out = Path(out_dir)
    (out / 'similarities').mkdir(parents=True, exist_ok=True)
    edges = synthetic.barabasi_albert_edges(n_nodes, m, seed)
    rowptr, col = synthetic.sorted_csr(edges, n_nodes)
    und = np.unique(np.sort(edges, axis=1), axis=0)
    with open(out / 'edge_list.txt', 'w') as f:
        for u, v in und:
            f.write('%d %d\n' % (u, v))
    subs = synthetic.bfs_subgraphs(rowptr, col, n_subgraphs, subgraph_nodes, seed + 1)
    dens = []
    for s in subs:
        ids = np.asarray(s)
        member = set(s)
        e = sum(1 for v in ids for w in col[rowptr[v]:rowptr[v + 1]] if int(w) in member) / 2
        k = len(ids)
        dens.append(e / (k * (k - 1) / 2) if k > 1 else 0.0)
    cuts = np.quantile(dens, np.linspace(0, 1, n_bins + 1)[1:-1])
    labels = np.searchsorted(cuts, dens, side='right')
    rng = np.random.default_rng(seed + 2)
    order = rng.permutation(n_subgraphs)
    split = np.empty(n_subgraphs, dtype=object)
    n_tr, n_va = int(0.8 * n_subgraphs), int(0.1 * n_subgraphs)
    split[order[:n_tr]] = 'train'
    split[order[n_tr:n_tr + n_va]] = 'val'
    split[order[n_tr + n_va:]] = 'test'
    with open(out / 'subgraphs.pth', 'w') as f:
        for s, lab, sp in zip(subs, labels, split):
            f.write('-'.join(str(v - 1) for v in s) + '\t' + str(int(lab)) + '\t' + sp + '\t\n')
    g = torch.Generator().manual_seed(seed + 3)
    torch.save(torch.randn(n_nodes, embed_dim, generator=g), out / ('%s_embeddings.pth' % embedding_type))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--nodes', type=int, default=1000)
    ap.add_argument('--m', type=int, default=5)
    ap.add_argument('--subgraphs', type=int, default=250)
    ap.add_argument('--subgraph-nodes', type=int, default=20)
    ap.add_argument('--embed', type=int, default=32)
    ap.add_argument('--no-metrics', action='store_true', help='skip the GPU graph-metric precompute')
    a = ap.parse_args(argv)
    d = write_density_dataset(a.out, a.nodes, a.m, a.subgraphs, a.subgraph_nodes, a.embed)
    if not a.no_metrics:
        from .precompute_graph_metrics import calculate_stats
        calculate_stats(d)
    print(d)


if __name__ == '__main__':
    main()
