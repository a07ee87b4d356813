"""Synthetic dataset generators in the reference's on-disk formats: DENSITY, CUT RATIO, CORENESS,
COMPONENT (reference prepare_dataset/prepare_dataset.py:26-831, recipes
prepare_dataset/README.md:59-125, constants prepare_dataset/config_prepare_dataset.py:15-41).

What the reference does, recipe by recipe, and what this module keeps:

  density    Barabasi-Albert base graph; subgraphs = first ``n_subgraph_nodes`` nodes of a BFS from a
             random start (prepare_dataset.py:288-327); then the GRAPH is edited, one subgraph after
             the other, until that subgraph's density is within DENSITY_EPSILON of a target drawn from
             DENSITY_RANGE: remove a random internal edge when too dense, add an edge between two
             random members when too sparse, at most MAX_TRIES edits (prepare_dataset.py:567-588).
  cut_ratio  BA base graph; a generated subgraph (complete graph) is PLANTED on randomly chosen base
             nodes (prepare_dataset.py:469-517); boundary edges are removed / added until the cut
             ratio  boundary / (|S| (N - |S|))  reaches a target from CUT_RATIO_RANGE +- epsilon
             (prepare_dataset.py:590-616).
  coreness   duplication-divergence base graph; for every core number k present, ``n_subgraphs``
             duplication-divergence subgraphs are planted on nodes of that core
             (prepare_dataset.py:227-286,469-517); label = bin of the average core number inside the
             subgraph (prepare_dataset.py:541-544,690-696).
  component  BA base graph; every subgraph is a set of generated components (extended BA graphs)
             STAPLED to the base graph by one edge each (prepare_dataset.py:404-467), the number of
             components drawn from CC_RANGE; label = one component vs. several
             (prepare_dataset.py:698-705).

Afterwards, as in the reference: keep the largest connected component and relabel nodes
consecutively (prepare_dataset.py:625-639,651-654), label = np.digitize of the property over
equal-count bins turned into letters 'A', 'B', ... (prepare_dataset.py:712-753), 80/10/10 split
(prepare_dataset.py:756-778), ``edge_list.txt`` + ``subgraphs.pth`` (prepare_dataset.py:781-799,822).

This is host-side Python on networkx like the reference (the graphs are ~10^3-10^4 nodes).  It is a
restatement of the recipes, not a replay of the reference's random stream: same distributions,
different draws.  Node embeddings are N(0,1) (pre-training them is out of scope).
"""
import argparse
import random
from pathlib import Path

import networkx as nx
import numpy as np
import torch

DENSITY_EPSILON = 0.01
DENSITY_RANGE = [0.05, 0.25, 0.45]
CUT_RATIO_EPSILON = 0.001
CUT_RATIO_RANGE = [0.005, 0.0125, 0.02]
CC_RANGE = [1, 1, 1, 1, 5, 6, 7, 8, 9, 10]
MAX_TRIES = 100

RECIPES = {      # prepare_dataset/README.md:59-125
    'density': dict(base='barabasi_albert', subgraph_type='bfs', n_subgraphs=250, n_subgraph_nodes=20, n=5000, m=5,
                    p=0.5, generator='complete', n_bins=3),
    'cut_ratio': dict(base='barabasi_albert', subgraph_type='plant', n_subgraphs=250, n_subgraph_nodes=20, n=5000, m=5,
                      p=0.5, generator='complete', n_bins=3),
    'coreness': dict(base='duplication_divergence_graph', subgraph_type='plant', n_subgraphs=30, n_subgraph_nodes=20,
                     n=5000, m=1, p=0.7, generator='duplication_divergence_graph', n_bins=3),
    'cc': dict(base='barabasi_albert', subgraph_type='staple', n_subgraphs=250, n_subgraph_nodes=15, n=1000, m=5,
               p=0.5, generator='extended_barabasi_albert', n_bins=2),
}


def _graph(kind, n, m, p, rng):
    seed = rng.randrange(1 << 30)
    if kind == 'barabasi_albert':
        return nx.barabasi_albert_graph(n, m, seed=seed)
    if kind == 'duplication_divergence_graph':
        return nx.duplication_divergence_graph(n, p, seed=seed)
    if kind == 'extended_barabasi_albert':
        return nx.extended_barabasi_albert_graph(n, 1, 0.0, 0.0, seed=seed)        # a random tree-like component
    if kind == 'complete':
        return nx.complete_graph(n)
    raise ValueError('unknown graph type %r' % (kind,))


def density(G, nodes):
    return nx.density(G.subgraph(nodes))


def cut_ratio(G, nodes):
    s = set(nodes)
    boundary = sum(1 for u in s for w in G[u] if w not in s)
    return boundary / (len(s) * (G.number_of_nodes() - len(s)))


def coreness(G, nodes):
    return float(np.mean(list(nx.core_number(G.subgraph(nodes)).values())))


def n_components(G, nodes):
    return nx.number_connected_components(G.subgraph(nodes))


PROPERTY = {'density': density, 'cut_ratio': cut_ratio, 'coreness': coreness, 'cc': n_components}


def bfs_subgraph(G, n_nodes, rng):
    """First n_nodes nodes reached breadth-first from a random start (one component)."""
    start = rng.choice(list(G.nodes))
    seen, order = {start}, [start]
    qi = 0
    while len(order) < n_nodes and qi < len(order):
        for w in G[order[qi]]:
            if w not in seen:
                seen.add(w)
                order.append(w)
                if len(order) == n_nodes:
                    break
        qi += 1
    return order


def plant(G, component, nodes):
    """Merge the edges of ``component`` (any graph with len(nodes) nodes) into G on ``nodes``."""
    mapping = dict(zip(component.nodes, nodes))
    G.add_edges_from((mapping[u], mapping[v]) for u, v in component.edges)


def staple(G, component, rng, base_nodes):
    """Disjoint union of G and ``component`` + one edge from a random node of ``base_nodes`` (the
    original base graph: attaching to an earlier stapled component could merge two components of
    one subgraph) to a random component node; returns the component's new node ids."""
    first = G.number_of_nodes()
    ids = list(range(first, first + component.number_of_nodes()))
    mapping = dict(zip(component.nodes, ids))
    G.add_nodes_from(ids)
    G.add_edges_from((mapping[u], mapping[v]) for u, v in component.edges)
    G.add_edge(rng.choice(base_nodes), rng.choice(ids))
    return ids


def edit_towards(G, nodes, prop, target, eps, rng, max_tries=MAX_TRIES):
    """prepare_dataset.py:567-616: edit G until the subgraph's property is within eps of target."""
    s = list(nodes)
    sset = set(s)
    for _ in range(max_tries):
        cur = PROPERTY[prop](G, s)
        if abs(cur - target) < eps:
            break
        if prop == 'density':
            if cur > target:
                G.remove_edge(*rng.choice(list(G.subgraph(s).edges)))
            else:
                G.add_edge(*rng.sample(s, 2))
        else:
            if cur > target:
                G.remove_edge(*rng.choice([(u, w) for u in s for w in G[u] if w not in sset]))
            else:
                out = rng.choice(list(G.nodes))
                while out in sset:
                    out = rng.choice(list(G.nodes))
                G.add_edge(rng.choice(s), out)


def equal_count_bins(values, n_bins):
    """prepare_dataset.py:712-728: cut points at the 1/n, 2/n, ... order statistics (last dropped)."""
    v = sorted(values)
    idx = (len(v) / float(n_bins)) * np.arange(1, n_bins + 1)
    cuts = np.unique(np.array([v[int(b) - 1] for b in idx]))
    return np.delete(cuts, len(cuts) - 1)


def letters(bin_ids):
    """prepare_dataset.py:730-753: bins -> 'A', 'B', ... in ascending bin order."""
    names = {b: chr(65 + i) for i, b in enumerate(sorted(set(int(x) for x in bin_ids)))}
    return [names[int(b)] for b in bin_ids]


def split_mask(n, rng):
    """prepare_dataset.py:756-778: 80 % train, the rest halved into val / test."""
    idx = list(range(n))
    rng.shuffle(idx)
    n_tr = int(n * 0.8)
    n_va = (n - n_tr) // 2
    mask = [''] * n
    for i in idx[:n_tr]:
        mask[i] = 'train'
    for i in idx[n_tr:n_tr + n_va]:
        mask[i] = 'val'
    for i in idx[n_tr + n_va:]:
        mask[i] = 'test'
    return mask


def generate(prop, seed=42, **overrides):
    """-> (graph with nodes 0..N-1, subgraphs (lists of node ids), labels (letters), property values)."""
    cfg = dict(RECIPES[prop])
    cfg.update(overrides)
    rng = random.Random(seed)
    G = _graph(cfg['base'], cfg['n'], cfg['m'], cfg['p'], rng)
    k, ns = cfg['n_subgraph_nodes'], cfg['n_subgraphs']
    subs = []
    if prop == 'density':
        subs = [bfs_subgraph(G, k, rng) for _ in range(ns)]
        for s in subs:
            edit_towards(G, s, 'density', rng.choice(DENSITY_RANGE), DENSITY_EPSILON, rng)
    elif prop == 'cut_ratio':
        for _ in range(ns):
            nodes = rng.sample(list(G.nodes), k)
            plant(G, _graph(cfg['generator'], k, cfg['m'], cfg['p'], rng), nodes)
            subs.append(nodes)
        for s in subs:
            edit_towards(G, s, 'cut_ratio', rng.choice(CUT_RATIO_RANGE), CUT_RATIO_EPSILON, rng)
    elif prop == 'coreness':
        for core in sorted(set(nx.core_number(G).values())):
            for _ in range(ns):
                pool = [v for v, c in nx.core_number(G).items() if c == core]
                if len(pool) < k:
                    break
                nodes = rng.sample(pool, k)
                plant(G, _graph(cfg['generator'], k, cfg['m'], cfg['p'], rng), nodes)
                subs.append(nodes)
    elif prop == 'cc':
        base_nodes = list(G.nodes)
        for _ in range(ns):
            nodes = []
            for _c in range(rng.choice(CC_RANGE)):
                nodes.extend(staple(G, _graph(cfg['generator'], k, cfg['m'], cfg['p'], rng), rng, base_nodes))
            subs.append(nodes)
    else:
        raise ValueError('unknown property %r' % (prop,))
    # largest connected component, consecutive ids (prepare_dataset.py:625-639)
    keep = max(nx.connected_components(G), key=len)
    G = G.subgraph(keep)
    mapping = {v: i for i, v in enumerate(G.nodes)}
    G = nx.relabel_nodes(G, mapping)
    subs = [[mapping[v] for v in s if v in keep] for s in subs]
    subs = [s for s in subs if s]
    values = [PROPERTY[prop](G, s) for s in subs]
    if prop == 'cc':
        labels = letters(np.digitize(values, bins=[1, 5]))            # one component vs. several
    else:
        labels = letters(np.digitize(values, bins=equal_count_bins(values, cfg['n_bins'])))
    return G, subs, labels, values


def write_dataset(out_dir, prop, seed=42, embed_dim=32, embedding_type='gin', **overrides):
    """Generates and writes edge_list.txt, subgraphs.pth and <type>_embeddings.pth under out_dir."""
    out = Path(out_dir)
    (out / 'similarities').mkdir(parents=True, exist_ok=True)
    G, subs, labels, values = generate(prop, seed, **overrides)
    nx.write_edgelist(G, str(out / 'edge_list.txt'), data=False)
    mask = split_mask(len(subs), random.Random(seed + 1))
    with open(out / 'subgraphs.pth', 'w') as f:
        for s, lab, sp in zip(subs, labels, mask):
            f.write('\t'.join(['-'.join(str(v) for v in s), str(lab), sp, '\n']))
    g = torch.Generator().manual_seed(seed + 3)
    torch.save(torch.randn(G.number_of_nodes(), embed_dim, generator=g), out / ('%s_embeddings.pth' % embedding_type))
    return out, dict(n_nodes=G.number_of_nodes(), n_edges=G.number_of_edges(), n_subgraphs=len(subs), labels=labels,
                     values=values)


def write_density_dataset(out_dir, n_nodes=1000, m=5, n_subgraphs=250, subgraph_nodes=20, embed_dim=32, seed=42,
                          embedding_type='gin', n_bins=3):
    """BASELINE.json configs[0]: the DENSITY recipe at the ~1k-node scale of config_prepare_dataset.py:15-31."""
    return write_dataset(out_dir, 'density', seed, embed_dim, embedding_type, n=n_nodes, m=m, n_subgraphs=n_subgraphs,
                         n_subgraph_nodes=subgraph_nodes, n_bins=n_bins)[0]


def main(argv=None):
    ap = argparse.ArgumentParser(description='Generate a synthetic SubGNN dataset directory')
    ap.add_argument('--out', required=True)
    ap.add_argument('--property', choices=sorted(RECIPES), default='density')
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--nodes', type=int, default=None, help='base graph size (default: the recipe\'s)')
    ap.add_argument('--subgraphs', type=int, default=None)
    ap.add_argument('--embed', type=int, default=32)
    ap.add_argument('--no-metrics', action='store_true', help='skip the GPU graph-metric precompute')
    a = ap.parse_args(argv)
    over = {}
    if a.nodes:
        over['n'] = a.nodes
    if a.subgraphs:
        over['n_subgraphs'] = a.subgraphs
    d, info = write_dataset(a.out, a.property, a.seed, a.embed, **over)
    if not a.no_metrics:
        from .precompute_graph_metrics import calculate_stats
        calculate_stats(d)
    print(d, {k: v for k, v in info.items() if k not in ('labels', 'values')})


if __name__ == '__main__':
    main()
