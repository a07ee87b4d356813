"""Synthetic dataset generators in the reference's on-disk formats: DENSITY, CUT RATIO, CORENESS,
COMPONENT (reference prepare_dataset/prepare_dataset.py:26-831, recipes
prepare_dataset/README.md:59-125, constants prepare_dataset/config_prepare_dataset.py:15-41).

Round 3: the recipes consume the random stream CALL FOR CALL like the reference does.  The reference draws
everything from Python's global ``random`` (``random.sample`` on node / edge views and sets; the unseeded
networkx generators of the CORENESS recipe fall back on it too) and seeds only the base-graph generators
with ``config.RANDOM_SEED``.  Here one ``random.Random(seed)`` plays the part of the global generator after
``random.seed(seed)``: every population is built by the same container operations (view -> tuple, the same
set differences, the same list comprehensions over the same networkx iterators), so the same networkx
version yields the same draws -- and the same edited graph, subgraphs, labels and split
(tests/test_host_logic.py::test_dataset_recipes_replay_the_reference_stream against
tests/golden/recipes.npz, produced by importing the reference with its global generator seeded).

Recipe by recipe (what is drawn, in order):

  density    BA base graph (seeded by number); per subgraph one start node ``sample(nodes, 1)`` and the
             first ``n_subgraph_nodes`` of ``nx.bfs_edges(depth_limit=3)`` (prepare_dataset.py:288-327);
             then per subgraph a target ``sample(DENSITY_RANGE, 1)`` and up to MAX_TRIES edits of the GRAPH:
             ``sample(subgraph edges, 1)`` removed when too dense, ``sample(subgraph nodes, 2)`` joined when
             too sparse (prepare_dataset.py:567-588).
  cut_ratio  BA base graph; per subgraph ``sample(nodes, n)`` and a complete graph composed onto them
             (prepare_dataset.py:469-517); then a target ``sample(CUT_RATIO_RANGE, 1)`` and edits of boundary
             edges: ``sample(boundary, 1)`` removed, or ``sample(members, 1)`` + ``sample(rest, 1)`` joined
             (prepare_dataset.py:590-616).
  coreness   duplication-divergence base graph (seeded); core numbers once; per core number, ``n_subgraphs``
             times: an UNSEEDED duplication-divergence component (it draws from the same stream),
             ``sample(nodes of that core not yet used, n)``, composed (prepare_dataset.py:227-286).
  component  BA base graph; per subgraph a hop count ``sample(k_hops_range, 1)``, a base node
             ``sample(base nodes, 1)``, a seeded extended-BA component stapled by one edge to
             ``sample(new ids, 1)``; a component count ``sample(CC_RANGE, 1)`` and further components stapled at
             ``sample(candidates k hops away, 1)`` (prepare_dataset.py:160-225).

Afterwards, as in the reference: largest connected component, nodes relabelled consecutively
(prepare_dataset.py:618-639), label = np.digitize of the property over equal-count bins turned into letters
(prepare_dataset.py:641-753), 80/10/10 split by three ``sample`` calls on shrinking sets
(prepare_dataset.py:756-778), ``edge_list.txt`` + ``subgraphs.pth`` (prepare_dataset.py:781-799,822).

One reference behaviour is reproduced by ``generate`` and repaired by ``write_dataset``: the relabelled subgraph
lists that ``_relabel_nodes`` returns are dropped by its caller (prepare_dataset.py:111-113), so when the edits
disconnect a node the subgraphs keep ids of the graph BEFORE relabelling (and the labels are computed on those).
``generate`` returns exactly that (it is what the fixtures pin); ``write_dataset(repair_ids=True)``, the default,
writes the relabelled lists and labels recomputed on them whenever the relabelling is not the identity -- a dataset
with ids that no longer exist cannot be read back.  Node embeddings are N(0,1) (pre-training them is out of scope,
SURVEY section 8f).
"""
import argparse
import random
from collections import Counter
from pathlib import Path

import networkx as nx
import numpy as np
import torch

DENSITY_EPSILON = 0.01
DENSITY_RANGE = [0.05, 0.25, 0.45]
CUT_RATIO_EPSILON = 0.001
CUT_RATIO_RANGE = [0.005, 0.0125, 0.02]
K_HOPS_RANGE = [0.12, 0.5, 1.0]
BA_P_RANGE = [0.1, 0.5, 0.9]
CC_RANGE = [1, 1, 1, 1, 5, 6, 7, 8, 9, 10]
MAX_TRIES = 100
BFS_MAX_DEPTH = 3

RECIPES = {      # prepare_dataset/README.md:59-125
    'density': dict(base='barabasi_albert', subgraph_type='bfs', n_subgraphs=250, n_subgraph_nodes=20, n=5000, m=5,
                    p=0.5, q=0, generator='complete', n_bins=3, n_components=1),
    'cut_ratio': dict(base='barabasi_albert', subgraph_type='plant', n_subgraphs=250, n_subgraph_nodes=20, n=5000, m=5,
                      p=0.5, q=0, generator='complete', n_bins=3, n_components=1),
    'coreness': dict(base='duplication_divergence_graph', subgraph_type='plant', n_subgraphs=30, n_subgraph_nodes=20,
                     n=5000, m=1, p=0.7, q=0, generator='duplication_divergence_graph', n_bins=3, n_components=1),
    'cc': dict(base='barabasi_albert', subgraph_type='staple', n_subgraphs=250, n_subgraph_nodes=15, n=1000, m=5,
               p=0.5, q=0, generator='extended_barabasi_albert', n_bins=2, n_components=None),
}


def _pick(rng, population, k):
    """``random.sample`` as the reference calls it on Python <= 3.10: a Set (networkx node / edge views, python sets)
    is turned into a tuple in its iteration order first; a list is sampled as it is."""
    if not isinstance(population, (list, tuple)):
        population = tuple(population)
    return rng.sample(population, k)


def _component(kind, n_nodes, m, p, q, seed, rng):
    """generate_subgraph (prepare_dataset.py:329-364): the seeded generators take ``seed``, the duplication-divergence
    component is unseeded in the reference and therefore draws from the shared stream."""
    if kind == 'complete':
        return nx.complete_graph(n_nodes)
    if kind == 'extended_barabasi_albert':
        return nx.extended_barabasi_albert_graph(n_nodes, m, p, q, seed=seed)
    if kind == 'duplication_divergence_graph':
        return nx.duplication_divergence_graph(n_nodes, p, seed=rng)
    if kind == 'barabasi_albert':
        return nx.barabasi_albert_graph(n_nodes, m, seed=seed)
    if kind == 'cycle':
        return nx.cycle_graph(n_nodes)
    if kind == 'path':
        return nx.path_graph(n_nodes)
    if kind == 'star':
        return nx.star_graph(n_nodes)
    raise ValueError('unknown component generator %r' % (kind,))


# ---- properties (prepare_dataset.py:519-550) -----------------------------------------------------------------

def density(G, nodes):
    return nx.density(G.subgraph(nodes))


def cut_ratio(G, nodes):
    view = G.subgraph(nodes)
    rest = set(G.nodes).difference(set(view.nodes))
    boundary = len(list(nx.edge_boundary(G, view.nodes, rest)))
    n, k = len(list(G.nodes)), len(list(view.nodes))
    return boundary / (k * (n - k))


def coreness(G, nodes):
    return float(np.average(list(nx.core_number(G.subgraph(nodes).copy()).values())))


def n_components(G, nodes):
    return nx.number_connected_components(G.subgraph(nodes))


PROPERTY = {'density': density, 'cut_ratio': cut_ratio, 'coreness': coreness, 'cc': n_components}


# ---- subgraph construction -----------------------------------------------------------------------------------

def _bfs_subgraphs(G, n_subgraphs, n_nodes, n_cc, rng):
    subs = []
    for _ in range(n_subgraphs):
        cur = []
        for start in _pick(rng, G.nodes, n_cc):
            reached = [start] + [v for _, v in nx.bfs_edges(G, start, depth_limit=BFS_MAX_DEPTH)]
            cur.extend(reached[:n_nodes])
        subs.append(cur)
    return subs


def _plant(G, comp, ids):
    """The component's edges merged onto the base nodes ``ids`` (nx.compose: same ids are the same node)."""
    mapping = {old: new for old, new in zip(comp.nodes, ids)}
    try:
        nx.relabel_nodes(comp, mapping, copy=False)            # in place, as the reference does (prepare_dataset.py:266,503)
    except nx.NetworkXUnfeasible:
        # drawn ids that permute the component's own labels in a cycle cannot be relabelled in place: the reference's
        # run ends here with this exception; a copy has the same edges
        comp = nx.relabel_nodes(comp, mapping, copy=True)
    return nx.compose(G, comp).copy()


def _planted_subgraphs(G, cfg, seed, rng):
    subs = []
    k = cfg['n_subgraph_nodes']
    for _ in range(cfg['n_subgraphs']):
        cur = []
        for _c in range(cfg['n_components']):
            comp = _component(cfg['generator'], k, cfg['m'], cfg['p'], cfg['q'], seed, rng)
            ids = _pick(rng, G.nodes, k)
            G = _plant(G, comp, ids)
            cur.extend(ids)
        subs.append(cur)
    return G, subs


def _coreness_subgraphs(G, cfg, seed, rng):
    core = nx.core_number(G)                                   # once, on the base graph
    by_core = {}
    for v, c in core.items():
        by_core.setdefault(c, []).append(v)
    k = cfg['n_subgraph_nodes']
    subs = []
    for c in by_core:                                          # core numbers in order of first appearance
        pool = by_core[c]
        for _ in range(cfg['n_subgraphs']):
            cur = []
            for _c in range(cfg['n_components']):
                if len(pool) < k:
                    break
                comp = _component(cfg['generator'], k, cfg['m'], cfg['p'], cfg['q'], seed, rng)
                ids = _pick(rng, pool, k)
                G = _plant(G, comp, ids)
                cur.extend(ids)
                pool = list(set(pool).difference(set(ids)))
                by_core[c] = pool
            if cur:
                subs.append(cur)
    return G, subs


def _staple(G, root, cfg, p, seed, rng):
    """A generated component joined to G by ONE edge root -- (random component node); -> (G, new ids, that node)."""
    k = cfg['n_subgraph_nodes']
    comp = _component(cfg['generator'], k, cfg['m'], p, cfg['q'], seed, rng)
    first = len(G.nodes)
    ids = list(range(first, first + k))
    joined = nx.disjoint_union(G, comp)
    anchor = _pick(rng, ids, 1)[0]
    joined.add_edge(root, anchor)
    return joined.copy(), ids, anchor


def _stapled_subgraphs(G, cfg, seed, rng):
    """prepare_dataset.py:160-225 (n_connected_components = None: the count is drawn per subgraph)."""
    diameter = nx.diameter(G)
    hops_range = [int(diameter * f) for f in K_HOPS_RANGE]
    base_nodes = G.nodes                                       # the BASE graph's view: stapled ids never join it
    k = cfg['n_subgraph_nodes']
    fixed_cc = cfg['n_components']
    kept = []
    n_cc = fixed_cc
    for _ in range(cfg['n_subgraphs']):
        hops = _pick(rng, hops_range, 1)[0]
        p = BA_P_RANGE[hops_range.index(hops)]
        root = _pick(rng, base_nodes, 1)[0]
        seen = [root]
        G, ids, anchor = _staple(G, root, cfg, p, seed, rng)
        cur = list(ids)
        seen.extend(ids)
        anchors = [anchor]
        reach = nx.single_source_shortest_path_length(G, root, cutoff=hops)
        cand = [v for v in reach
                if all(nx.shortest_path_length(G, a, v) == hops for a in anchors) and v not in seen]
        if not cand:
            far = max(reach.values())
            cand = [v for v, d in reach.items() if d == far]
        if fixed_cc is None:
            n_cc = _pick(rng, CC_RANGE, 1)[0]
        for _c in range(n_cc - 1):
            root2 = _pick(rng, cand, 1)[0]
            seen.append(root2)
            G, ids, anchor = _staple(G, root2, cfg, p, seed, rng)
            cur.extend(ids)
            seen.extend(ids)
            anchors.append(anchor)
        if len(cur) >= k * n_cc:
            got = nx.number_connected_components(G.subgraph(cur))
            if (fixed_cc is None and got in CC_RANGE) or (fixed_cc is not None and got > 1):
                kept.append(cur)
    out = []
    for s in kept:                                             # counted again on the final graph
        got = nx.number_connected_components(G.subgraph(s))
        if (fixed_cc is None and got in CC_RANGE) or (fixed_cc is not None and got > 1):
            out.append(s)
    return G, out


# ---- graph edits towards a property value (prepare_dataset.py:552-616) ----------------------------------------

def _edit_density(G, subs, rng):
    for s in subs:
        view = G.subgraph(s)                                   # a live view: it follows the edits
        target = _pick(rng, DENSITY_RANGE, 1)[0]
        for _ in range(MAX_TRIES):
            cur = nx.density(view)
            if abs(cur - target) < DENSITY_EPSILON:
                break
            if cur > target:
                G.remove_edge(*_pick(rng, view.edges, 1)[0])
            else:
                G.add_edge(*_pick(rng, view.nodes, 2))


def _edit_cut_ratio(G, subs, rng):
    for s in subs:
        view = G.subgraph(s)
        target = _pick(rng, CUT_RATIO_RANGE, 1)[0]
        for _ in range(MAX_TRIES):
            cur = cut_ratio(G, s)
            if abs(cur - target) < CUT_RATIO_EPSILON:
                break
            rest = set(G.nodes).difference(set(view.nodes))
            boundary = list(nx.edge_boundary(G, view.nodes, rest))
            if cur > target:
                G.remove_edge(*_pick(rng, boundary, 1)[0])
            else:
                inside = _pick(rng, view.nodes, 1)[0]
                outside = _pick(rng, rest, 1)[0]
                G.add_edge(inside, outside)


# ---- labels and split ------------------------------------------------------------------------------------------

def equal_count_bins(values, n_bins):
    """prepare_dataset.py:712-728: cut points at the 1/n, 2/n, ... order statistics (last dropped)."""
    v = sorted(values)
    idx = (len(v) / float(n_bins)) * np.arange(1, n_bins + 1)
    cuts = np.unique(np.array([v[int(b) - 1] for b in idx]))
    return np.delete(cuts, len(cuts) - 1)


def letters(bin_ids):
    """prepare_dataset.py:730-753: bins -> 'A', 'B', ... in the iteration order of the set of bin ids."""
    names = {}
    for i, b in enumerate(set(bin_ids)):
        names[b] = chr(65 + i)
    return [names[b] for b in bin_ids]


def labels_of(G, subs, prop, n_bins):
    values = [PROPERTY[prop](G, s) for s in subs]
    if prop == 'cc':
        ids = np.digitize(values, bins=[1, 5])                 # one component vs. several
    elif prop == 'density':
        ids = np.digitize(values, bins=equal_count_bins(values, len(DENSITY_RANGE)))
    elif prop == 'cut_ratio':
        ids = np.digitize(values, bins=equal_count_bins(values, len(CUT_RATIO_RANGE)))
    else:
        ids = np.digitize(values, bins=equal_count_bins(values, n_bins))
    return letters(ids), values


def split_mask(n, rng):
    """prepare_dataset.py:756-778: 0 train (80 %), 1 val, 2 test -- three draws from shrinking index sets."""
    idx = set(range(n))
    train = list(_pick(rng, idx, int(len(idx) * 0.8)))
    idx = idx.difference(set(train))
    val = list(_pick(rng, idx, len(idx) // 2))
    idx = idx.difference(set(val))
    test = list(_pick(rng, idx, len(idx)))
    tr, va, te = set(train), set(val), set(test)
    return [0 if i in tr else (1 if i in va else 2) for i in range(n) if i in tr or i in va or i in te]


# ---- the recipes ------------------------------------------------------------------------------------------------

def generate(prop, seed=42, rng=None, **overrides):
    """-> dict(graph, subgraphs, labels, values, relabelled_subgraphs, identity): the state the reference's
    SyntheticGraph ends with for ``desired_property = prop`` when its global generator was seeded with ``seed`` --
    ``subgraphs`` / ``labels`` exactly as it holds them (ids from before the final relabelling, see the module
    docstring), ``relabelled_subgraphs`` what the relabelling maps them to, ``identity`` whether the two agree.
    ``rng``: the stream to continue (a random.Random), else a fresh one seeded with ``seed``."""
    cfg = dict(RECIPES[prop])
    cfg.update(overrides)
    rng = rng if rng is not None else random.Random(seed)
    if cfg['base'] == 'barabasi_albert':
        G = nx.barabasi_albert_graph(cfg['n'], cfg['m'], seed=seed)
    elif cfg['base'] == 'duplication_divergence_graph':
        G = nx.duplication_divergence_graph(cfg['n'], cfg['p'], seed=seed)
    else:
        raise ValueError('unknown base graph %r' % (cfg['base'],))
    kind = cfg['subgraph_type']
    if kind == 'bfs':
        subs = _bfs_subgraphs(G, cfg['n_subgraphs'], cfg['n_subgraph_nodes'], cfg['n_components'], rng)
    elif kind == 'plant' and prop == 'coreness':
        G, subs = _coreness_subgraphs(G, cfg, seed, rng)
    elif kind == 'plant':
        G, subs = _planted_subgraphs(G, cfg, seed, rng)
    elif kind == 'staple':
        G, subs = _stapled_subgraphs(G, cfg, seed, rng)
    else:
        raise ValueError('unknown subgraph type %r' % (kind,))
    if prop == 'density':
        _edit_density(G, subs, rng)
    elif prop == 'cut_ratio':
        _edit_cut_ratio(G, subs, rng)
    # largest connected component, consecutive ids (prepare_dataset.py:618-639)
    keep = max(nx.connected_components(G), key=len)
    dropped = set(G.nodes).difference(set(keep))
    G = G.subgraph(keep)
    mapping = {old: new for old, new in zip(list(G.nodes), range(len(G.nodes)))}
    G = nx.relabel_nodes(G, mapping)
    relabelled = [[mapping[v] for v in s if v not in dropped] for s in subs]
    identity = all(old == new for old, new in mapping.items()) and not dropped
    if not nx.is_connected(G):
        G = G.subgraph(max(nx.connected_components(G), key=len))
    labels, values = labels_of(G, subs, prop, cfg['n_bins'])
    if prop == 'cc' and len(Counter(labels)) != 2:
        raise AssertionError('component recipe: both label classes must occur (prepare_dataset.py:704)')
    return dict(graph=G, subgraphs=subs, labels=labels, values=values, relabelled_subgraphs=relabelled,
                identity=identity, rng=rng, n_bins=cfg['n_bins'])


def write_dataset(out_dir, prop, seed=42, embed_dim=32, embedding_type='gin', repair_ids=True, **overrides):
    """Generates and writes edge_list.txt, subgraphs.pth (prepare_dataset.py:781-799,822) and
    <type>_embeddings.pth under out_dir.  ``repair_ids``: when the final relabelling moved node ids, write the
    relabelled subgraphs with labels recomputed on them instead of the reference's stale lists (module docstring)."""
    out = Path(out_dir)
    (out / 'similarities').mkdir(parents=True, exist_ok=True)
    st = generate(prop, seed, **overrides)
    G, subs, labels, values = st['graph'], st['subgraphs'], st['labels'], st['values']
    repaired = False
    if repair_ids and not st['identity']:
        subs = st['relabelled_subgraphs']
        labels, values = labels_of(G, subs, prop, st['n_bins'])
        repaired = True
    nx.write_edgelist(G, str(out / 'edge_list.txt'), data=False)
    mask = split_mask(len(labels), st['rng'])                  # the stream goes on: the reference draws the split last
    names = {0: 'train', 1: 'val', 2: 'test'}
    with open(out / 'subgraphs.pth', 'w') as f:
        for s, lab, sp in zip(subs, labels, mask):
            if len(s) == 0:
                continue
            f.write('\t'.join(['-'.join(str(v) for v in s), str(lab), names[sp], '\n']))
    g = torch.Generator().manual_seed(seed + 3)
    torch.save(torch.randn(G.number_of_nodes(), embed_dim, generator=g), out / ('%s_embeddings.pth' % embedding_type))
    return out, dict(n_nodes=G.number_of_nodes(), n_edges=G.number_of_edges(), n_subgraphs=len(subs), labels=labels,
                     values=values, ids_repaired=repaired)


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
