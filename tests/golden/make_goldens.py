#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Needs /root/reference (read-only) -- it does not exist on the GPU box, so only the small
``tests/golden/*.npz`` files this script writes are committed and travel.  The reference's
own functions are executed unmodified; what is substituted:

  * import stand-ins for packages the image lacks (tests/golden/_standins/README.md);
  * the three global RNG streams, replaced by the counter-based draw tape of
    oracle/tape.py (hard part 2) through module-level shims
    (``np.random.choice``, ``random.uniform``, ``torch.randn`` as seen by
    anchor_patch_samplers.py only);
  * ``nx.adjacency_matrix`` (as seen by subgraph_utils.py) wrapped to return a scipy
    ``spmatrix`` as networkx 2.4 did, so ``.todense()`` yields ``np.matrix`` (su:136-143);
  * stored N-anchor tensors made ``.contiguous()`` (torch>=2 stride behaviour; values
    unchanged).

A fixture is data: inputs and expected outputs of each stage boundary (g1..g12 of
SURVEY.md section 8c).  No reference source text is stored.
"""
import os
import sys
import json
import shutil
import tempfile
import itertools
from pathlib import Path

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path('/root/reference')
sys.path[:0] = [str(HERE / '_standins'), str(REF / 'SubGNN'), str(REF), str(REPO)]

import numpy as np          # noqa: E402
import random as pyrandom   # noqa: E402
import networkx as nx       # noqa: E402
import scipy.sparse         # noqa: E402
import torch                # noqa: E402

from oracle import tape as T   # noqa: E402

import config as refconfig                 # noqa: E402  (reference config.py)
import anchor_patch_samplers as aps        # noqa: E402
import subgraph_utils as su                # noqa: E402
import gamma as refgamma                   # noqa: E402
import subgraph_mpn as refmpn              # noqa: E402
import SubGNN as S                         # noqa: E402

SEED = 20260101


# ---------------------------------------------------------------------------------------
# RNG shims driven by the tape
# ---------------------------------------------------------------------------------------

class Ctx:
    def __init__(self):
        self.seed = SEED
        self.mode = None
        self.stream = 0
        self.item = 0
        self.j = 0
        self.walk_items = None
        self.layer = 0
        self.rand_matrix = None
        self.rand_slots = 0
        self.rand_slot = 0
        self.rand_stream = 0
        self.record = {}


CTX = Ctx()


class _NpRandomShim:
    def choice(self, a, size=None, replace=True):
        seq = list(a) if not isinstance(a, np.ndarray) else a
        n = len(seq)
        arr = np.asarray(seq)
        m = CTX.mode
        if m == 'walk':
            k = 1 if size is None else int(size)
            out = []
            for _ in range(k):
                out.append(arr[T.choice_index(CTX.seed, CTX.stream, CTX.item, CTX.j, n)])
                CTX.j += 1
            return out[0] if size is None else np.array(out)
        if m == 'struct_start':
            st = T.stream_id(T.STREAM_STRUCT_START)
            out = np.array([arr[T.choice_index(CTX.seed, st, 0, j, n)] for j in range(int(size))])
            CTX.mode = 'walk'
            return out
        if m == 's_pick':
            st = T.stream_id(T.STREAM_S_PICK, 0, CTX.layer)
            CTX.layer += 1
            return np.array([arr[T.choice_index(CTX.seed, st, 0, j, n)] for j in range(int(size))])
        if m == 'pos':
            return np.array([arr[T.choice_index(CTX.seed, CTX.stream, CTX.item, j, n)] for j in range(int(size))])
        raise RuntimeError('np.random.choice called outside a tape context: %r' % (m,))


class _NpShim:
    random = _NpRandomShim()

    def __getattr__(self, k):
        return getattr(np, k)


class _PyRandomShim:
    def uniform(self, a, b):
        assert (a, b) == (0, 1) and CTX.mode == 'walk'
        u = T.uniform01(CTX.seed, CTX.stream, CTX.item, CTX.j)
        CTX.j += 1
        return u

    def __getattr__(self, k):
        return getattr(pyrandom, k)


class _TorchShim:
    def randn(self, shape):
        """One 'randn' matrix per anchor slot (aps:177,189).  Only the argmax after the reference
        zeroes the PAD columns matters, so the matrix holds +1 at the column of the entry the tape's
        neighbourhood-anchor law picks for (row, slot) and -1 elsewhere (all -1 = every variate
        negative = PAD wins on a padded row)."""
        ids = CTX.rand_matrix
        assert tuple(shape) == tuple(ids.shape)
        R, L = ids.shape
        z = -np.ones((R, L), dtype=np.float64)
        for r in range(R):
            cols = np.nonzero(ids[r])[0]
            order = cols[np.argsort(ids[r][cols], kind='stable')]      # columns by ascending id
            k = T.nanchor_pick(CTX.seed, CTX.rand_stream, r * CTX.rand_slots + CTX.rand_slot, len(cols), len(cols) < L)
            if k >= 0:
                z[r, order[k]] = 1.0
        CTX.rand_slot += 1
        return torch.from_numpy(z)

    def __getattr__(self, k):
        return getattr(torch, k)


class _NxShim:
    def adjacency_matrix(self, G, *a, **k):
        return scipy.sparse.csr_matrix(nx.adjacency_matrix(G, *a, **k))

    def __getattr__(self, k):
        return getattr(nx, k)


aps.np = _NpShim()
aps.random = _PyRandomShim()
aps.torch = _TorchShim()
su.nx = _NxShim()

_orig = {k: getattr(aps, k) for k in (
    'triangular_random_walk', 'sample_structure_anchor_patches', 'perform_random_walks',
    'sample_neighborhood_anchor_patch', 'sample_position_anchor_patches',
    'init_anchors_neighborhood', 'init_anchors_pos_int', 'init_anchors_pos_ext',
    'init_anchors_structure')}
_orig_get_border_nodes = su.get_border_nodes


def w_triangular_random_walk(hparams, G, sub, walk_len, in_border, all_valid, inside):
    CTX.item = next(CTX.walk_items)
    CTX.j = 0
    CTX.mode = 'walk'
    if 'views' in CTX.record and inside and sub is not G:
        CTX.record['views'].setdefault(id(sub), [int(v) for v in sub.nodes()])
    return _orig['triangular_random_walk'](hparams, G, sub, walk_len, in_border, all_valid, inside)


def w_get_border_nodes(graph, subgraph):
    b, non = _orig_get_border_nodes(graph, subgraph)
    if 'in_border' in CTX.record:
        CTX.record['in_border'].append(np.asarray(b).reshape(-1).astype(np.int64))
        CTX.record['views_bor'].append(np.array([int(v) for v in subgraph.nodes()], dtype=np.int64))
    return b, non


def w_sample_structure_anchor_patches(hparams, G, device, mse):
    n = mse * hparams['n_anchor_patches_structure'] * hparams['n_layers']
    CTX.mode = 'struct_start'
    CTX.stream = T.stream_id(T.STREAM_STRUCT_PATCH)
    CTX.walk_items = iter(range(n))
    return _orig['sample_structure_anchor_patches'](hparams, G, device, mse)


def w_perform_random_walks(hparams, G, ids, inside):
    W = hparams['n_triangular_walks']
    nonpad = [p for p in range(ids.shape[0]) if int((ids[p] != 0).sum()) > 0]
    CTX.walk_items = iter([p * W + w for p in nonpad for w in range(W)])
    CTX.stream = T.stream_id(T.STREAM_WALK_INT if inside else T.STREAM_WALK_BOR)
    CTX.mode = 'walk'
    CTX.record['in_border'] = []
    CTX.record['views_bor'] = []
    CTX.record['views'] = {}
    CTX.record['view_patch_order'] = []
    # record the induced-subgraph node-view order per patch (inside walks use it, aps:70)
    out = _orig['perform_random_walks'](hparams, G, ids, inside)
    key = 'int' if inside else 'bor'
    CTX.record['walk_in_border_' + key] = CTX.record.pop('in_border')
    CTX.record['walk_views_bor_' + key] = CTX.record.pop('views_bor')
    CTX.record.pop('views')
    return out


_split_of_tensor = {}
_layer_count = {}


def w_init_anchors_neighborhood(split, hparams, G, device, tr, va, te, trb, vab, teb):
    _split_of_tensor.clear()
    _layer_count.clear()
    for name, t in (('train', tr), ('val', va), ('test', te)):
        if t is not None:
            _split_of_tensor[id(t)] = name
    return _orig['init_anchors_neighborhood'](split, hparams, G, device, tr, va, te, trb, vab, teb)


def w_sample_neighborhood_anchor_patch(hparams, G, cc_ids, border_set, sample_inside=True):
    split = _split_of_tensor[id(cc_ids)]
    layer = _layer_count.get((split, sample_inside), 0)
    _layer_count[(split, sample_inside)] = layer + 1
    mat = cc_ids if sample_inside else border_set
    CTX.rand_matrix = mat.reshape(mat.shape[0] * mat.shape[1], -1).numpy()
    CTX.rand_slots = hparams['n_anchor_patches_N_in'] if sample_inside else hparams['n_anchor_patches_N_out']
    CTX.rand_slot = 0
    CTX.rand_stream = T.stream_id(T.STREAM_N_INT if sample_inside else T.STREAM_N_BOR, split, layer)
    return _orig['sample_neighborhood_anchor_patch'](hparams, G, cc_ids, border_set, sample_inside)


_pos_schedule = None


def w_init_anchors_pos_int(split, hparams, G, device, tr, va, te):
    global _pos_schedule
    names = {'all': ['train', 'val', 'test'], 'train_val': ['train', 'val'], 'test': ['test']}[split]
    data = {'train': tr, 'val': va, 'test': te}
    _pos_schedule = iter([(nm, l, s) for nm in names for l in range(hparams['n_layers']) for s in range(len(data[nm]))])
    return _orig['init_anchors_pos_int'](split, hparams, G, device, tr, va, te)


def w_init_anchors_pos_ext(hparams, G, device):
    CTX.layer = 0
    return _orig['init_anchors_pos_ext'](hparams, G, device)


def w_sample_position_anchor_patches(hparams, G, subgraph=None):
    CTX.mode = 'pos'
    if not subgraph:
        CTX.stream = T.stream_id(T.STREAM_P_EXT, 0, CTX.layer)
        CTX.layer += 1
        CTX.item = 0
    else:
        nm, l, s = next(_pos_schedule)
        CTX.stream = T.stream_id(T.STREAM_P_INT, nm, l)
        CTX.item = s
    return _orig['sample_position_anchor_patches'](hparams, G, subgraph)


def w_init_anchors_structure(hparams, sa, irw, brw):
    CTX.mode = 's_pick'
    CTX.layer = 0
    return _orig['init_anchors_structure'](hparams, sa, irw, brw)


_wrappers = {
    'triangular_random_walk': w_triangular_random_walk,
    'sample_structure_anchor_patches': w_sample_structure_anchor_patches,
    'perform_random_walks': w_perform_random_walks,
    'sample_neighborhood_anchor_patch': w_sample_neighborhood_anchor_patch,
    'sample_position_anchor_patches': w_sample_position_anchor_patches,
    'init_anchors_neighborhood': w_init_anchors_neighborhood,
    'init_anchors_pos_int': w_init_anchors_pos_int,
    'init_anchors_pos_ext': w_init_anchors_pos_ext,
    'init_anchors_structure': w_init_anchors_structure,
}
for _k, _w in _wrappers.items():
    setattr(aps, _k, _w)
    setattr(S, _k, _w)          # SubGNN.py:55 ``from anchor_patch_samplers import *``
su.get_border_nodes = w_get_border_nodes


# ---------------------------------------------------------------------------------------
# tiny datasets in the on-disk formats
# ---------------------------------------------------------------------------------------

def write_dataset(root, name, edges, subgraphs, labels, splits, D, with_ego, rng):
    d = Path(root) / name
    (d / 'similarities').mkdir(parents=True, exist_ok=True)
    with open(d / 'edge_list.txt', 'w') as f:
        for u, v in edges:
            f.write('%d %d\n' % (u, v))
    with open(d / 'subgraphs.pth', 'w') as f:
        for nodes, lab, sp in zip(subgraphs, labels, splits):
            f.write('-'.join(str(n) for n in nodes) + '\t' + str(lab) + '\t' + sp + '\t\n')
    n = 1 + max(max(e) for e in edges)
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    emb = torch.randn(n, D, generator=g)
    torch.save(emb, d / 'gin_embeddings.pth')
    G0 = nx.Graph()
    G0.add_nodes_from(range(n))
    G0.add_edges_from(edges)
    apsp = np.zeros((n, n), dtype=np.float64)
    for s, dd in nx.all_pairs_shortest_path_length(G0):
        for t, l in dd.items():
            apsp[s, t] = l
    np.save(d / 'shortest_path_matrix.npy', apsp)
    with open(d / 'degree_sequence.txt', 'w') as f:
        json.dump({str(v): G0.degree(v) for v in range(n)}, f)
    if with_ego:
        with open(d / 'ego_graphs.txt', 'w') as f:
            json.dump({str(v): [int(w) for w in G0.neighbors(v)] for v in range(n)}, f)
    return d, emb.numpy(), apsp


def make_tiny(rng):
    """BA(60,3) with shuffled, randomly oriented edge list (unsorted node order) and
    multi-component subgraphs."""
    G = nx.barabasi_albert_graph(60, 3, seed=11)
    extra = [(1, 7), (7, 9), (1, 9), (20, 21), (21, 22), (20, 22), (33, 2), (33, 17)]
    G.add_edges_from(extra)
    edges = list(G.edges())
    perm = rng.permutation(len(edges))
    edges = [edges[i] if rng.random() < 0.5 else edges[i][::-1] for i in perm]
    subgraphs, labels, splits = [], [], []
    nodes = list(G.nodes())
    for i in range(18):
        ncc = int(rng.integers(1, 4))
        chosen = []
        for _ in range(ncc):
            start = int(rng.choice(nodes))
            size = int(rng.integers(2, 7))
            comp = list(itertools.islice(nx.bfs_tree(G, start).nodes(), size))
            chosen.extend(comp)
        chosen = list(dict.fromkeys(chosen))
        rng.shuffle(chosen)
        subgraphs.append(chosen)
        labels.append(int(rng.integers(0, 3)))
        splits.append('train' if i < 10 else ('val' if i < 14 else 'test'))
    labels[0], labels[1], labels[2] = 0, 1, 2
    return edges, subgraphs, labels, splits


def make_density(rng):
    """DENSITY-style: BA(200,5), sorted edge list, single-component BFS subgraphs."""
    G = nx.barabasi_albert_graph(200, 5, seed=42)
    edges = sorted(G.edges())
    subgraphs, labels, splits = [], [], []
    for i in range(30):
        start = int(rng.integers(0, 200))
        comp = list(itertools.islice(nx.bfs_tree(G, start, depth_limit=3).nodes(), 12))
        subgraphs.append(comp)
        labels.append(i % 3)
        splits.append('train' if i < 20 else ('val' if i < 25 else 'test'))
    return edges, subgraphs, labels, splits


BASE_HP = {
    "use_neighborhood": True, "use_structure": True, "use_position": True, "seed": 0,
    "node_embed_size": 8, "structure_patch_type": "triangular_random_walk",
    "lstm_aggregator": "last", "n_processes": 2, "resample_anchor_patches": False,
    "freeze_node_embeds": False, "use_mpn_projection": True, "compute_similarities": True,
    "sample_walk_len": 12, "n_triangular_walks": 3, "random_walk_len": 6, "rw_beta": 0.65,
    "batch_size": 6, "learning_rate": 1e-3, "n_layers": 2, "neigh_sample_border_size": 2,
    "n_anchor_patches_pos_out": 7, "n_anchor_patches_pos_in": 5, "n_anchor_patches_N_in": 4,
    "n_anchor_patches_N_out": 6, "n_anchor_patches_structure": 5, "linear_hidden_dim_1": 16,
    "linear_hidden_dim_2": 8, "lin_dropout": 0.0, "lstm_dropout": 0.0, "lstm_n_layers": 1,
    "cc_aggregator": "sum", "trainable_cc": False, "max_sim_epochs": 2, "embedding_type": "gin",
}


def paths_for(name):
    return dict(graph_path=name + '/edge_list.txt', subgraph_path=name + '/subgraphs.pth',
                embedding_path=name + '/gin_embeddings.pth', similarities_path=name + '/similarities/',
                shortest_paths_path=name + '/shortest_path_matrix.npy',
                degree_dict_path=name + '/degree_sequence.txt', ego_graph_path=name + '/ego_graphs.txt')


def ragged_pad(lists, fill=-1):
    L = max((len(x) for x in lists), default=0)
    out = np.full((len(lists), max(L, 1)), fill, dtype=np.int64)
    for i, x in enumerate(lists):
        out[i, :len(x)] = x
    return out


def t2n(t):
    return t.detach().cpu().numpy()


def build_model(name, hp, seed=0):
    torch.manual_seed(seed)
    hp = dict(hp)
    m = S.SubGNN(hp, **paths_for(name))
    return m


def stage_goldens(root, name, out):
    """g1..g9, g12 for one dataset directory (with the ego dict present or not as written)."""
    hp = dict(BASE_HP)
    model = build_model(name, hp)
    G = model.networkx_graph
    # g1
    order = [int(v) for v in G.nodes()]
    out['g1_node_order'] = np.array(order, dtype=np.int64)
    m = max(order)
    rowptr = np.zeros(m + 2, dtype=np.int64)
    cols = []
    for v in range(1, m + 1):
        nb = [int(w) for w in G.neighbors(v)] if v in G else []
        rowptr[v + 1] = rowptr[v] + len(nb)
        cols.extend(nb)
    out['g1_rowptr'] = rowptr
    out['g1_col'] = np.array(cols, dtype=np.int32)
    with open(Path(root) / name / 'edge_list.txt') as f:
        out['edge_list'] = np.array([[int(x) for x in l.split()] for l in f if l.strip()], dtype=np.int64)
    out['embeddings'] = t2n(model.node_embeddings.weight)
    out['apsp'] = np.load(Path(root) / name / 'shortest_path_matrix.npy')
    for sp in ('train', 'val', 'test'):
        out['subgraphs_' + sp] = ragged_pad(getattr(model, sp + '_sub_G'), 0)
        out['labels_' + sp] = t2n(getattr(model, sp + '_sub_G_label')).reshape(-1)

    model.prepare_data()
    for sp in ('train', 'val'):
        out['g2_cc_ids_' + sp] = t2n(getattr(model, sp + '_cc_ids'))
        out['g3_border_' + sp] = t2n(getattr(model, sp + '_N_border'))
        out['g4_np_sim_' + sp] = t2n(getattr(model, sp + '_neigh_pos_similarities'))
        out['g7_int_struc_sim_' + sp] = t2n(getattr(model, sp + '_int_struc_similarities'))
        out['g7_bor_struc_sim_' + sp] = t2n(getattr(model, sp + '_bor_struc_similarities'))
    # border sets for the other k as well (direct call, both k)
    ego = None
    ep = Path(root) / name / 'ego_graphs.txt'
    if ep.exists():
        ego = {int(k): v for k, v in json.load(open(ep)).items()}
    for k in (1, 2, 3):
        rows = []
        cc = model.train_cc_ids
        for s in range(cc.shape[0]):
            for c in range(cc.shape[1]):
                rows.append(sorted(int(v) for v in su.get_component_border_neighborhood_set(G, cc[s, c], k, ego)))
        out['g3_border_k%d_train' % k] = ragged_pad(rows, -1)
    # g5
    out['g5_structure_anchors'] = t2n(model.structure_anchors)
    out['g5_int_walks'] = t2n(model.int_structure_anchor_random_walks)
    out['g5_bor_walks'] = t2n(model.bor_structure_anchor_random_walks)
    out['g5_in_border'] = ragged_pad(CTX.record['walk_in_border_bor'], 0)
    out['g5_views_bor'] = ragged_pad(CTX.record['walk_views_bor_bor'], 0)
    # node-view order used by the inside walks: recompute it exactly as aps:138 does
    views = []
    for p in range(model.structure_anchors.shape[0]):
        ids = model.structure_anchors[p]
        ids = ids[ids != 0]
        views.append([int(v) for v in G.subgraph(ids.numpy()).nodes()])
    out['g5_views_int'] = ragged_pad(views, 0)
    # g6 degree sequences (unsorted is not observable in the reference; store the sorted ones)
    degd = {int(k): v for k, v in json.load(open(Path(root) / name / 'degree_sequence.txt')).items()}
    for internal in (True, False):
        key = 'int' if internal else 'ext'
        rows = [refgamma.get_degree_sequence(G, model.structure_anchors[a], degd, internal=internal)
                for a in range(model.structure_anchors.shape[0])]
        out['g6_anchor_deg_' + key] = ragged_pad(rows, -1)
        cc = model.train_cc_ids.view(-1, model.train_cc_ids.shape[-1])
        rows = [refgamma.get_degree_sequence(G, cc[r], degd, internal=internal) for r in range(cc.shape[0])]
        out['g6_cc_deg_' + key + '_train'] = ragged_pad(rows, -1)
        rows = [refgamma.get_degree_sequence(G, cc[r], None, internal=internal) for r in range(cc.shape[0])]
        out['g6_cc_deg_nodict_' + key + '_train'] = ragged_pad(rows, -1)
    # g8
    for sp in ('train', 'val'):
        for l in range(hp['n_layers']):
            model.anchors_neigh_int[sp][l] = model.anchors_neigh_int[sp][l].contiguous()
            model.anchors_neigh_border[sp][l] = model.anchors_neigh_border[sp][l].contiguous()
            out['g8_N_int_%s_%d' % (sp, l)] = t2n(model.anchors_neigh_int[sp][l])
            out['g8_N_bor_%s_%d' % (sp, l)] = t2n(model.anchors_neigh_border[sp][l])
            out['g8_P_int_%s_%d' % (sp, l)] = t2n(model.anchors_pos_int[sp][l])
    for l in range(hp['n_layers']):
        out['g8_P_ext_%d' % l] = t2n(model.anchors_pos_ext[l])
        patches, idx, irw, brw = model.anchors_structure[l]
        out['g8_S_idx_%d' % l] = np.array([int(i) for i in idx], dtype=np.int64)
        out['g8_S_patches_%d' % l] = t2n(patches)
        out['g8_S_int_rw_%d' % l] = t2n(irw)
        out['g8_S_bor_rw_%d' % l] = t2n(brw)
    # g12 + g9 on one batch
    ds = S.SubgraphDataset(model.train_sub_G, model.train_sub_G_label, model.train_cc_ids, model.train_N_border,
                           model.train_neigh_pos_similarities, model.train_int_struc_similarities,
                           model.train_bor_struc_similarities, model.multilabel, model.multilabel_binarizer)
    idxs = [7, 2, 5, 0, 9, 3]
    batch = model._pad_collate([ds[i] for i in idxs])
    out['g12_idx'] = np.array(idxs, dtype=np.int64)
    for k in ('subgraph_ids', 'cc_ids', 'N_border', 'NP_sim', 'I_S_sim', 'B_S_sim', 'subgraph_idx', 'label'):
        out['g12_' + k] = t2n(batch[k])
    cc_ids = batch['cc_ids']
    mask = (cc_ids != 0)[:, :, 0]
    out['state_dict_keys'] = np.array(sorted(model.state_dict().keys()))
    for k, v in model.state_dict().items():
        out['sd/' + k] = t2n(v)
    with torch.no_grad():
        for ch in ('neighborhood', 'position', 'structure'):
            for inside in (True, False):
                ap, am, ae = aps.get_anchor_patches('train', model.hparams, G, model.node_embeddings, batch['subgraph_idx'],
                                                    cc_ids, mask, model.lstm, model.anchors_neigh_int, model.anchors_neigh_border,
                                                    model.anchors_pos_int, model.anchors_pos_ext, model.anchors_structure, 1, ch, inside, model.device)
                tag = 'g9_%s_%s_' % (ch[0].upper(), 'in' if inside else 'out')
                out[tag + 'patches'] = t2n(ap)
                out[tag + 'mask'] = t2n(am)
                out[tag + 'embeds'] = t2n(ae)
    return model, batch


def mpn_goldens(model, batch, out):
    """g10: SG_MPN.forward outputs and grads for N-style (per-CC anchors, dense NP slab) and
    S-style (shared anchors + index list) inputs, incl. padded CC rows and PAD anchors."""
    G = model.networkx_graph
    cc_ids = batch['cc_ids']
    mask = (cc_ids != 0)[:, :, 0]
    for ch, sims, tag in (('neighborhood', batch['NP_sim'], 'N'), ('position', batch['NP_sim'], 'P'),
                          ('structure', batch['B_S_sim'], 'S')):
        for inside in (True, False):
            if ch == 'structure':
                sims = batch['I_S_sim'] if inside else batch['B_S_sim']
            layer = 1
            mp = {'neighborhood': model.neighborhood_mpns, 'position': model.position_mpns,
                  'structure': model.structure_mpns}[ch][layer]['internal' if inside else 'border']
            ap, am, ae = aps.get_anchor_patches('train', model.hparams, G, model.node_embeddings, batch['subgraph_idx'],
                                                cc_ids, mask, model.lstm, model.anchors_neigh_int, model.anchors_neigh_border,
                                                model.anchors_pos_int, model.anchors_pos_ext, model.anchors_structure, layer, ch, inside, model.device)
            ae = ae.detach().clone().requires_grad_(True)
            g = torch.Generator().manual_seed(5)
            cc_embeds = torch.randn(cc_ids.shape[0], cc_ids.shape[1], ae.shape[-1], generator=g).requires_grad_(True)
            idx = model.anchors_structure[layer][1] if ch == 'structure' else None
            for p in mp.parameters():
                p.grad = None
            o1, o2 = mp(G, sims, cc_ids, cc_embeds, mask, ap, ae, am, idx)
            g1 = torch.randn(o1.shape, generator=g)
            g2 = torch.randn(o2.shape, generator=g)
            ((o1 * g1).sum() + (o2 * g2).sum()).backward()
            t = 'g10_%s_%s_' % (tag, 'in' if inside else 'out')
            out[t + 'sims'] = t2n(sims)
            out[t + 'cc_embeds'] = t2n(cc_embeds)
            out[t + 'patches'] = t2n(ap)
            out[t + 'mask'] = t2n(am)
            out[t + 'anchor_embeds'] = t2n(ae)
            if idx is not None:
                out[t + 'sim_index'] = np.array([int(i) for i in idx], dtype=np.int64)
            out[t + 'out_cc'] = t2n(o1)
            out[t + 'out_pos'] = t2n(o2)
            out[t + 'gout_cc'] = t2n(g1)
            out[t + 'gout_pos'] = t2n(g2)
            out[t + 'grad_cc_embeds'] = t2n(cc_embeds.grad)
            out[t + 'grad_anchor_embeds'] = t2n(ae.grad)
            out[t + 'W'] = t2n(mp.linear.weight)
            out[t + 'b'] = t2n(mp.linear.bias)
            out[t + 'wp'] = t2n(mp.linear_position.weight)
            out[t + 'bp'] = t2n(mp.linear_position.bias)
            out[t + 'grad_W'] = t2n(mp.linear.weight.grad)
            out[t + 'grad_b'] = t2n(mp.linear.bias.grad)
            out[t + 'grad_wp'] = t2n(mp.linear_position.weight.grad)
            out[t + 'grad_bp'] = t2n(mp.linear_position.bias.grad)


def forward_goldens(root, name, out, variants):
    """g11: full forward logits + loss + grads through the reference LightningModule."""
    for vname, over in variants.items():
        hp = dict(BASE_HP)
        hp.update(over)
        model = build_model(name, hp, seed=3)
        model.prepare_data()
        for sp in ('train', 'val'):
            for l in range(hp['n_layers']):
                model.anchors_neigh_int[sp][l] = model.anchors_neigh_int[sp][l].contiguous()
                model.anchors_neigh_border[sp][l] = model.anchors_neigh_border[sp][l].contiguous()
        if hp['trainable_cc']:
            # give the trainable CC embeddings non-trivial values
            g = torch.Generator().manual_seed(9)
            with torch.no_grad():
                for nm in ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B'):
                    p = getattr(model, 'train_%s_cc_embed' % nm)
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        ds = S.SubgraphDataset(model.train_sub_G, model.train_sub_G_label, model.train_cc_ids, model.train_N_border,
                               model.train_neigh_pos_similarities, model.train_int_struc_similarities,
                               model.train_bor_struc_similarities, model.multilabel, model.multilabel_binarizer)
        idxs = [7, 2, 5, 0, 9, 3]
        batch = model._pad_collate([ds[i] for i in idxs])
        model.train()
        model.zero_grad()
        res = model.training_step(batch, 0)
        logits = model.forward('train', model.train_N_I_cc_embed, model.train_N_B_cc_embed, model.train_S_I_cc_embed,
                               model.train_S_B_cc_embed, model.train_P_I_cc_embed, model.train_P_B_cc_embed,
                               batch['subgraph_ids'], batch['cc_ids'], batch['subgraph_idx'], batch['NP_sim'],
                               batch['I_S_sim'], batch['B_S_sim'])
        loss = res['loss']
        model.backward(None, loss, None, 0)
        t = 'g11_%s/' % vname
        out[t + 'hparams'] = np.array(json.dumps({k: v for k, v in model.hparams.items()}))
        out[t + 'idx'] = np.array(idxs, dtype=np.int64)
        out[t + 'logits'] = t2n(logits)
        out[t + 'loss'] = t2n(loss)
        for k, v in model.state_dict().items():
            out[t + 'sd/' + k] = t2n(v)
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[t + 'grad/' + k] = t2n(p.grad)
        if hp['trainable_cc']:
            for nm in ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B'):
                p = getattr(model, 'train_%s_cc_embed' % nm)
                out[t + 'cc_param/' + nm] = t2n(p)
                if p.grad is not None:
                    out[t + 'cc_grad/' + nm] = t2n(p.grad)
        # prepared state the product module is fed with (layered parity: same stage inputs)
        for sp in ('train',):
            out[t + 'cc_ids_' + sp] = t2n(getattr(model, sp + '_cc_ids'))
            out[t + 'np_sim_' + sp] = t2n(getattr(model, sp + '_neigh_pos_similarities'))
            out[t + 'int_sim_' + sp] = t2n(getattr(model, sp + '_int_struc_similarities'))
            out[t + 'bor_sim_' + sp] = t2n(getattr(model, sp + '_bor_struc_similarities'))
            out[t + 'border_' + sp] = t2n(getattr(model, sp + '_N_border'))
            for l in range(hp['n_layers']):
                out[t + 'N_int_%s_%d' % (sp, l)] = t2n(model.anchors_neigh_int[sp][l])
                out[t + 'N_bor_%s_%d' % (sp, l)] = t2n(model.anchors_neigh_border[sp][l])
                out[t + 'P_int_%s_%d' % (sp, l)] = t2n(model.anchors_pos_int[sp][l])
        for l in range(hp['n_layers']):
            out[t + 'P_ext_%d' % l] = t2n(model.anchors_pos_ext[l])
            patches, idx, irw, brw = model.anchors_structure[l]
            out[t + 'S_idx_%d' % l] = np.array([int(i) for i in idx], dtype=np.int64)
            out[t + 'S_patches_%d' % l] = t2n(patches)
            out[t + 'S_int_rw_%d' % l] = t2n(irw)
            out[t + 'S_bor_rw_%d' % l] = t2n(brw)
        out[t + 'structure_anchors'] = t2n(model.structure_anchors)


def main():
    rng = np.random.default_rng(123)
    root = tempfile.mkdtemp(prefix='subgnn_golden_')
    refconfig.PROJECT_ROOT = Path(root)
    try:
        for name, maker, ego, D in (('tiny', make_tiny, False, 8), ('tiny_ego', make_tiny, True, 8),
                                    ('density', make_density, True, 8)):
            rng2 = np.random.default_rng(77 if name.startswith('tiny') else 78)
            edges, subgraphs, labels, splits = maker(rng2)
            write_dataset(root, name, edges, subgraphs, labels, splits, D, ego, rng2)
            out = {'seed': np.array(SEED), 'base_hparams': np.array(json.dumps(BASE_HP)),
                   'has_ego': np.array(ego)}
            model, batch = stage_goldens(root, name, out)
            if name == 'tiny':
                mpn_goldens(model, batch, out)
                forward_goldens(root, name, out, {
                    'sum': {},
                    'max_trainable': {'cc_aggregator': 'max', 'trainable_cc': True, 'lstm_n_layers': 2},
                    'bn': {'batch_norm': True, 'n_layers': 1},
                    'sumlstm_norm': {'lstm_aggregator': 'sum', 'norm_pos_struc_embed': True, 'n_layers': 1},
                    'ff_attn': {'ff_attn': True, 'n_layers': 1},
                })
            np.savez_compressed(HERE / (name + '.npz'), **out)
            print(name, 'written:', len(out), 'arrays,', (HERE / (name + '.npz')).stat().st_size // 1024, 'KiB')
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
