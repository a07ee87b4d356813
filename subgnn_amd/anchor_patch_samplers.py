"""Anchor-patch sampling and retrieval (mirrors reference SubGNN/anchor_patch_samplers.py).

Function names, argument meaning and returned containers follow the reference; the graph
argument is a ``DeviceGraph`` (CSR in HBM) instead of a networkx graph, and every random
decision reads the counter-based draw tape (tape.py) keyed by ``hparams['seed']`` instead of
the global numpy / random / torch streams -- which is what lets walks and anchor draws run as
parallel HIP kernels and still be reproducible draw for draw.
"""
from collections import defaultdict

import torch

from . import ops, tape
from .config import PAD_VALUE


def _seed(hparams):
    return int(hparams.get('seed', 0)) & tape.MASK64


# ---------------------------------------------------------------------------------------
# triangular random walks (aps:20-158, 210-243)
# ---------------------------------------------------------------------------------------

VIEW_PAIRWISE_MAX = 1 << 26      # P * L * L entries of the pairwise comparison; above it the sort-based form


def patch_node_views(anchor_patch_ids):
    """Node view of every patch's induced subgraph: unique ids in first-occurrence order
    (the reference iterates a networkx subgraph view whose order is CPython-set order).
    No host round trip: an entry is kept when no earlier entry of its row has the same id (rows are
    walks of a few dozen steps: the L x L comparison per row is a handful of launches; very long
    rows take the sort-based form), and the kept entries are packed by ops.Ragged.from_mask."""
    P, L = anchor_patch_ids.shape
    dev = anchor_patch_ids.device
    ids = anchor_patch_ids
    if P * L * L <= VIEW_PAIRWISE_MAX and ids.is_cuda and ids.dtype == torch.int64 and 0 < P <= ops.pack_fused_limits()[0] \
            and 0 < P * L <= ops.pack_fused_limits()[1]:
        return ops.Ragged.from_first_occurrence(ids)                              # ONE launch: first occurrences found and packed
    if P * L * L <= VIEW_PAIRWISE_MAX and ids.is_cuda:
        keep = ops.first_occurrence_mask(ids)                                     # one launch: entry i against the i entries before it
    elif P * L * L <= VIEW_PAIRWISE_MAX:
        earlier = torch.ones(L, L, dtype=torch.bool, device=dev).tril(-1)          # [i, j]: j < i
        dup = ((ids.unsqueeze(2) == ids.unsqueeze(1)) & earlier.unsqueeze(0)).any(dim=2)
        keep = ~dup & (ids != PAD_VALUE)
    else:
        rows = torch.arange(P, device=dev).unsqueeze(1).expand(P, L)
        key = (rows * (1 << 32) + ids).reshape(-1)
        skey, sidx = torch.sort(key, stable=True)
        first = torch.ones_like(skey, dtype=torch.bool)
        first[1:] = skey[1:] != skey[:-1]
        keep = torch.zeros(P * L, dtype=torch.bool, device=dev)
        keep[sidx] = first
        keep = keep.view(P, L) & (ids != PAD_VALUE)
    return ops.Ragged.from_mask(ids, keep)


def in_border_sets(graph, views):
    """per patch: the view nodes with an edge leaving the patch (su.get_border_nodes)."""
    flags = ops.patch_in_border(graph, views)
    if flags.is_cuda:
        return ops.filter_sets(views, flags)
    L = views._max_len
    if L is None:
        L = views.max_len
    padded = views.to_padded(width=L, fill=PAD_VALUE)
    fpad = ops.Ragged(views.ptr, flags.to(torch.int32), max_len=L).to_padded(width=L, fill=0)
    return ops.Ragged.from_mask(padded, fpad != 0)


def perform_random_walks(hparams, networkx_graph, anchor_patch_ids, inside, views=None, in_border=None, first_patch=0):
    """aps:118-158 -> (n_patches, n_triangular_walks, random_walk_len) int64, PAD filled.
    ``first_patch``: the given patches are rows first_patch ... of a longer patch list (a rank's share of the shared
    patches under strong scaling): their walks read the tape items of the whole list's walk numbers."""
    g = networkx_graph
    ids = anchor_patch_ids.to(g.device)
    P = ids.shape[0]
    W, T = hparams['n_triangular_walks'], hparams['random_walk_len']
    views = views if views is not None else patch_node_views(ids)
    if inside:
        out = ops.triangular_walks(g, 1, P * W, T, hparams['rw_beta'], _seed(hparams),
                                   tape.stream_id(tape.STREAM_WALK_INT), patches=views, walks_per_patch=W,
                                   item_base=first_patch * W)
    else:
        in_border = in_border if in_border is not None else in_border_sets(g, views)
        out = ops.triangular_walks(g, 2, P * W, T, hparams['rw_beta'], _seed(hparams),
                                   tape.stream_id(tape.STREAM_WALK_BOR), patches=views, in_border=in_border,
                                   walks_per_patch=W, item_base=first_patch * W)
    return out.view(P, W, T)


def perform_random_walks_both(hparams, networkx_graph, anchor_patch_ids, views=None, in_border=None, first_patch=0):
    """perform_random_walks(..., inside=True) and (..., inside=False) (aps:118-158) over the same patches in one launch ->
    (internal walks, border walks), each (n_patches, n_triangular_walks, random_walk_len) int64."""
    g = networkx_graph
    ids = anchor_patch_ids.to(g.device)
    P = ids.shape[0]
    W, T = hparams['n_triangular_walks'], hparams['random_walk_len']
    views = views if views is not None else patch_node_views(ids)
    in_border = in_border if in_border is not None else in_border_sets(g, views)
    out = ops.triangular_walks_both(g, P * W, T, hparams['rw_beta'], _seed(hparams), tape.stream_id(tape.STREAM_WALK_INT),
                                    tape.stream_id(tape.STREAM_WALK_BOR), views, in_border, W, item_base=first_patch * W)
    return out[0].view(P, W, T), out[1].view(P, W, T)


def sample_structure_anchor_patches(hparams, networkx_graph, device, max_sim_epochs, trim=True, share=None):
    """aps:210-243 -> (n sampled patches, max patch length) int64 (trailing all-PAD columns
    trimmed, as padding to the longest patch does in the reference; ``trim=False`` keeps the walks' full width -- the
    longest patch is a value on the device, and reading it makes the host wait for everything queued on the stream:
    the per-pass path does without, its consumers strip PAD entries themselves).
    'triangular_random_walk': every patch is a walk of sample_walk_len steps over the whole graph.
    'ego_graph' (aps:226-228): patch i = the nodes within structure_anchor_patch_radius hops of the i-th
    start node (the starts are one np.random.choice over the graph's nodes: tape item 0, draw i), centre
    included, listed in the base graph's node order like the node view of nx.ego_graph -- the k-hop BFS
    kernel (sgnn_khop_border) from singleton sets."""
    g = networkx_graph
    n = max_sim_epochs * hparams['n_anchor_patches_structure'] * hparams['n_layers']
    kind = hparams['structure_patch_type']
    if kind == 'triangular_random_walk' and share is not None:
        # share = (first walk, one past the last): this call draws a rank's share of the n walks (same tape items)
        lo, hi = share
        return ops.triangular_walks(g, 0, hi - lo, hparams['sample_walk_len'], hparams['rw_beta'], _seed(hparams),
                                    tape.stream_id(tape.STREAM_STRUCT_PATCH), item_base=lo).contiguous()
    if kind == 'triangular_random_walk':
        out = ops.triangular_walks(g, 0, n, hparams['sample_walk_len'], hparams['rw_beta'], _seed(hparams),
                                   tape.stream_id(tape.STREAM_STRUCT_PATCH))
    elif kind == 'ego_graph':
        pool = ops.Ragged(_span_ptr(g.n_nodes, g.device), g.node_order)
        starts = ops.choice_ragged(pool, n, _seed(hparams), tape.stream_id(tape.STREAM_STRUCT_START))[0]
        singles = ops.Ragged(torch.arange(n + 1, dtype=torch.int64, device=g.device), starts.to(torch.int32).contiguous(), max_len=1)
        hood = ops.khop_border(g, singles, int(hparams['structure_anchor_patch_radius']))
        # node-view order: by position in the graph's node order (ids -> position + 1, sorted per patch, back to ids)
        L = int(hood.lengths.max().item()) + 1 if n > 0 else 1
        width = max(L, 1)
        pos = torch.zeros((n, width), dtype=torch.int64, device=g.device)
        pos[:, 0] = g.node_pos[starts].long() + 1
        if width > 1:
            hp_ = ops.Ragged(hood.ptr, (g.node_pos[hood.nodes.long()] + 1).to(torch.int32), max_len=width - 1).to_padded(width - 1)
            pos[:, 1:] = hp_
        big = torch.iinfo(torch.int64).max
        srt = torch.sort(torch.where(pos == 0, torch.full_like(pos, big), pos), dim=1).values
        out = torch.where(srt == big, torch.zeros_like(srt), g.node_order[(srt - 1).clamp(min=0, max=g.n_nodes - 1)].long())
    else:
        raise NotImplementedError("structure_patch_type %r" % kind)
    if not trim and kind == 'triangular_random_walk':
        return out.contiguous()
    longest = int((out != PAD_VALUE).sum(1).max().item()) if n > 0 else 0
    return out[:, :max(longest, 1)].contiguous()


# ---------------------------------------------------------------------------------------
# sampling (aps:163-208)
# ---------------------------------------------------------------------------------------

def sample_neighborhood_anchor_patch(hparams, networkx_graph, cc_ids, border_set, sample_inside=True, split='train',
                                     layer=0, epoch=0):
    """aps:163-198 -> (S, C, n_anchor_patches_N_in | _N_out) int64."""
    mat = cc_ids if sample_inside else border_set
    S, C, L = mat.shape
    A = hparams['n_anchor_patches_N_in'] if sample_inside else hparams['n_anchor_patches_N_out']
    kind = tape.STREAM_N_INT if sample_inside else tape.STREAM_N_BOR
    out = ops.sample_anchors_padded(mat.reshape(S * C, L).contiguous(), A, _seed(hparams),
                                    tape.stream_id(kind, split, layer, epoch))
    return out.view(S, C, A)


def sample_position_anchor_patches(hparams, networkx_graph, subgraph=None, split='train', layer=0, item=0, epoch=0):
    """aps:200-208 for ONE draw list (python list out, like the reference): without ``subgraph`` the
    n_anchor_patches_pos_out shared border anchors drawn from all graph nodes (aps:206), with it the
    n_anchor_patches_pos_in internal anchors drawn from that subgraph's node list (aps:208).
    ``item`` is the subgraph's number within its split -- the tape item the batched form
    (init_anchors_pos_int) uses for the same subgraph, so both forms give the same draws."""
    g = networkx_graph
    if not subgraph:
        r = ops.Ragged(_span_ptr(g.n_nodes, g.device), g.node_order)
        return ops.choice_ragged(r, hparams['n_anchor_patches_pos_out'], _seed(hparams),
                                 tape.stream_id(tape.STREAM_P_EXT, 0, layer, epoch))[0].tolist()
    r = ops.Ragged.from_lists([[int(v) for v in subgraph]], g.device)
    return ops.choice_ragged(r, hparams['n_anchor_patches_pos_in'], _seed(hparams),
                             tape.stream_id(tape.STREAM_P_INT, split, layer, epoch), item_base=int(item))[0].tolist()


# ---------------------------------------------------------------------------------------
# initialisation (aps:248-328)
# ---------------------------------------------------------------------------------------

_SPLITS = {'all': ['train', 'val', 'test'], 'train_val': ['train', 'val'], 'test': ['test']}


def init_anchors_neighborhood(split, hparams, networkx_graph, device, train_cc_ids, val_cc_ids, test_cc_ids,
                              train_N_border, val_N_border, test_N_border, epoch=0):
    data = {'train': (train_cc_ids, train_N_border), 'val': (val_cc_ids, val_N_border),
            'test': (test_cc_ids, test_N_border)}
    anchors_int_neigh, anchors_border_neigh = defaultdict(dict), defaultdict(dict)
    for name in _SPLITS[split]:
        cc, bs = data[name]
        for n in range(hparams['n_layers']):
            anchors_int_neigh[name][n] = sample_neighborhood_anchor_patch(hparams, networkx_graph, cc, bs, True, name, n, epoch)
            anchors_border_neigh[name][n] = sample_neighborhood_anchor_patch(hparams, networkx_graph, cc, bs, False, name, n, epoch)
    return anchors_int_neigh, anchors_border_neigh


def init_anchors_pos_int(split, hparams, networkx_graph, device, train_sub_G, val_sub_G, test_sub_G, epoch=0):
    """(S, n_anchor_patches_pos_in) per split and layer; ``*_sub_G`` are lists of node lists."""
    data = {'train': train_sub_G, 'val': val_sub_G, 'test': test_sub_G}
    anchors = defaultdict(dict)
    for name in _SPLITS[split]:
        subs = data[name] if isinstance(data[name], ops.Ragged) else ops.Ragged.from_lists(data[name], networkx_graph.device)
        for n in range(hparams['n_layers']):
            anchors[name][n] = ops.choice_ragged(subs, hparams['n_anchor_patches_pos_in'], _seed(hparams),
                                                 tape.stream_id(tape.STREAM_P_INT, name, n, epoch))
    return anchors


_SPAN_PTR = {}


def _span_ptr(n, device):
    """[0, n] on the device: the row pointer of a one-set pool.  Uploaded once per (n, device) -- a
    host->device copy of a fresh tensor blocks the host until the stream has drained."""
    key = (int(n), str(device))
    if key not in _SPAN_PTR:
        _SPAN_PTR[key] = torch.tensor([0, int(n)], dtype=torch.int64, device=device)
    return _SPAN_PTR[key]


def init_anchors_pos_ext(hparams, networkx_graph, device, epoch=0):
    g = networkx_graph
    order = ops.Ragged(_span_ptr(g.n_nodes, g.device), g.node_order)
    return {n: ops.choice_ragged(order, hparams['n_anchor_patches_pos_out'], _seed(hparams),
                                 tape.stream_id(tape.STREAM_P_EXT, 0, n, epoch))[0]
            for n in range(hparams['n_layers'])}


def init_anchors_structure(hparams, structure_anchors, int_structure_anchor_rw, bor_structure_anchor_rw,
                           indices_on_device=False, epoch=0):
    """aps:300-328.  The second entry of every layer's tuple is the list of picked patch numbers, as in
    the reference; ``indices_on_device`` keeps it as the device tensor it was drawn into (the per-pass
    path: no device->host->device round trip for a value only ever used as a column index)."""
    dev = structure_anchors.device
    P = structure_anchors.shape[0]
    pool = ops.Ragged(_span_ptr(P, dev), torch.arange(P, dtype=torch.int32, device=dev))
    out = {}
    for n in range(hparams['n_layers']):
        idx = ops.choice_ragged(pool, hparams['n_anchor_patches_structure'], _seed(hparams),
                                tape.stream_id(tape.STREAM_S_PICK, 0, n, epoch))[0]
        out[n] = (structure_anchors[idx, :], idx if indices_on_device else [int(i) for i in idx.tolist()],
                  int_structure_anchor_rw[idx, :, :], bor_structure_anchor_rw[idx, :, :])
    return out


# ---------------------------------------------------------------------------------------
# retrieval (aps:333-433): the reference-shaped (materialising) path
# ---------------------------------------------------------------------------------------

def embed_anchor_patch(node_matrix, anchor_patch_ids, device):
    return node_matrix(anchor_patch_ids.to(device)), (anchor_patch_ids != PAD_VALUE).bool()


def aggregate_structure_anchor_patch(hparams, networkx_graph, lstm, node_matrix, anchor_patch_ids, all_patch_walks,
                                     inside, device, table=None):
    """aps:413-433: walks (A, W, T) -> LSTM over each walk's embeddings -> sum over W -> (A, D)."""
    n = anchor_patch_ids.shape[0]
    walks = all_patch_walks.to(device)
    E = node_matrix.weight if table is None else table
    if hparams.get('fused_forward', True) and hasattr(lstm, 'forward_walks') and walks.is_cuda and walks.dtype == torch.int64:
        # lookup, LSTM, last step, Linear and the sum over a patch's walks as one chain of this library's launches
        ids = walks.reshape(n * hparams['n_triangular_walks'], hparams['random_walk_len'])
        pre = getattr(all_patch_walks, '_sgnn_sorted', None)
        if pre is not None:
            ids._sgnn_sorted = pre
        X = lstm.forward_walks(E, ids, hparams['n_triangular_walks'])
        if X is not None:
            return X
    walk_embeds = ops.gather_rows(E, walks)
    x = walk_embeds.view(n * hparams['n_triangular_walks'], hparams['random_walk_len'], hparams['node_embed_size'])
    h = lstm(x).view(n, hparams['n_triangular_walks'], -1)
    return torch.sum(h, dim=1)


def get_anchor_patches(dataset_type, hparams, networkx_graph, node_matrix, subgraph_idx, cc_ids, cc_embed_mask, lstm,
                       anchors_neigh_int, anchors_neigh_border, anchors_pos_int, anchors_pos_ext, anchors_structure,
                       layer_num, channel, inside, device=None):
    """aps:333-399 -> (anchor_patches (B,C,A,Lp), anchor_mask, anchor_embeds (B,C,A,D))."""
    B, C, _ = cc_ids.shape
    dev = cc_ids.device
    sidx = subgraph_idx.view(-1)
    if channel == 'neighborhood':
        src = anchors_neigh_int if inside else anchors_neigh_border
        patches = src[dataset_type][layer_num].to(dev)[sidx]
        embeds, mask = embed_anchor_patch(node_matrix, patches, dev)
        return patches.unsqueeze(-1), mask.unsqueeze(-1), embeds
    if channel == 'position':
        if inside:
            patches = anchors_pos_int[dataset_type][layer_num].to(dev)[sidx].unsqueeze(1).repeat(1, C, 1)
        else:
            patches = anchors_pos_ext[layer_num].to(dev).view(1, 1, -1).repeat(B, C, 1)
        patches[~cc_embed_mask] = PAD_VALUE
        embeds, mask = embed_anchor_patch(node_matrix, patches, dev)
        return patches.unsqueeze(-1), mask.unsqueeze(-1), embeds
    if channel == 'structure':
        patches, indices, int_rw, bor_rw = anchors_structure[layer_num]
        emb = aggregate_structure_anchor_patch(hparams, networkx_graph, lstm, node_matrix, patches,
                                               (int_rw if inside else bor_rw).to(dev), inside, dev)
        patches = patches.to(dev).unsqueeze(0).unsqueeze(0).repeat(B, C, 1, 1)
        patches[~cc_embed_mask] = PAD_VALUE
        mask = (patches != PAD_VALUE).bool()
        embeds = emb.unsqueeze(0).unsqueeze(0).repeat(B, C, 1, 1) * cc_embed_mask.view(B, C, 1, 1).to(emb.dtype)
        return patches, mask, embeds
    raise Exception('An invalid channel has been entered.')
