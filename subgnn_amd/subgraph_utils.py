"""Host helpers + set kernels (mirrors the hot subset of reference SubGNN/subgraph_utils.py)."""
import numpy as np
import torch

from . import ops
from .config import PAD_VALUE


def read_subgraphs(sub_f):
    """Parse ``subgraphs.pth`` (text: ``n1-n2-...\\tlabel[-label]\\tsplit\\t``; su:24-92).
    Labels are numbered in order of first appearance; val/test are swapped when val is the
    smaller split (su:89-90)."""
    label_ids = {}
    split_nodes = {'train': [], 'val': [], 'test': []}
    split_labels = {'train': [], 'val': [], 'test': []}
    multilabel = False
    with open(sub_f) as fin:
        for line in fin:
            cols = line.split('\t')
            nodes = [int(n) for n in cols[0].split('-') if n != '']
            if not nodes:
                continue
            labs = cols[1].split('-')
            multilabel = multilabel or len(labs) > 1
            for lab in labs:
                label_ids.setdefault(lab, len(label_ids))
            sp = cols[2].strip()
            if sp in split_nodes:
                split_nodes[sp].append(nodes)
                split_labels[sp].append([label_ids[lab] for lab in labs])
    if not multilabel:
        for sp in split_labels:
            split_labels[sp] = torch.tensor(split_labels[sp]).long().squeeze()
    if len(split_nodes['val']) < len(split_nodes['test']):
        return (split_nodes['train'], split_labels['train'], split_nodes['test'], split_labels['test'],
                split_nodes['val'], split_labels['val'])
    return (split_nodes['train'], split_labels['train'], split_nodes['val'], split_labels['val'],
            split_nodes['test'], split_labels['test'])


def get_border_nodes(graph, patch_nodes):
    """su.get_border_nodes for one patch (list of unique node ids in view order): the patch
    nodes that have an edge leaving the patch (with the reference's id-1 / node-order
    indexing, su:136-143).  Returns a python list in patch order."""
    r = ops.Ragged.from_lists([list(patch_nodes)], graph.device)
    flags = ops.patch_in_border(graph, r)[:len(patch_nodes)].cpu().numpy()
    return [v for v, f in zip(patch_nodes, flags) if f]


def get_component_border_neighborhood_set(graph, component, k, ego_graph_dict=None):
    """su.get_component_border_neighborhood_set for one component -> python set.
    ``ego_graph_dict`` not None selects the ``ego_graphs.txt`` semantics (1 hop, ids shifted by
    -1, su:168-174)."""
    comp = [int(v) for v in component if int(v) != PAD_VALUE]
    r = ops.Ragged.from_lists([comp], graph.device)
    b = ops.khop_border(graph, r, k, ego_dict_mode=ego_graph_dict is not None)
    return set(b.to_lists()[0])


def border_sets(graph, cc_ids, k, ego_dict_mode=False):
    """SubGNN.initialize_border_sets (S.py:673-700) for a whole (S, C, L) component tensor ->
    padded (S, C, Lb) int64 on the device, every row sorted ascending (canonical order)."""
    S, C, L = cc_ids.shape
    sets = ops.Ragged.from_padded(cc_ids.reshape(S * C, L))
    b = ops.sort_ragged(ops.khop_border(graph, sets, k, ego_dict_mode=ego_dict_mode))
    # id 0 can be a legitimate entry in ego-dict mode; it is indistinguishable from PAD in the
    # padded matrix, exactly as in the reference
    return b.to_padded().view(S, C, -1)


def masked_sum(vector, mask, dim=1, keepdim=False):
    """su.masked_sum (su:213-237).  (B, C, H) with a (B, C[,1]) mask over dim 1 runs the HIP
    kernel; other layouts are rejected (the hot path has only this one)."""
    if vector.dim() != 3 or dim != 1 or keepdim:
        raise NotImplementedError('masked_sum: only (B, C, H) over dim=1')
    m = mask.reshape(vector.shape[0], vector.shape[1])
    return ops.masked_sum(vector, m)


def weighted_sum(matrix, attention):
    """su.weighted_sum for the (B, C, H) x (B, C) case used with ff_attn (S.py:301)."""
    return torch.bmm(attention.unsqueeze(1), matrix).squeeze(1)


def calc_f1(logits, labels, avg_type='macro', multilabel_binarizer=None):
    from sklearn.metrics import f1_score
    if multilabel_binarizer is not None:
        pred = torch.sigmoid(logits) > 0.5
    else:
        pred = torch.argmax(logits, dim=-1)
    return torch.tensor([f1_score(labels.cpu().detach(), pred.cpu().detach(), average=avg_type)])


def calc_accuracy(logits, labels, multilabel_binarizer=None):
    """su.calc_accuracy (su:108-124): sklearn accuracy_score semantics (exact-match for multilabel),
    evaluated on the device -- training_step calls this every batch and a host round trip there
    would serialise the step."""
    if multilabel_binarizer is not None:
        pred = torch.sigmoid(logits) > 0.5
        acc = (pred == labels.bool()).all(dim=-1).float().mean()
    else:
        acc = (torch.argmax(logits, 1) == labels).float().mean()
    return acc.reshape(1)


def trim_zero_columns(x):
    """_pad_collate's trimming of all-PAD columns (S.py:1098-1099,1109-1110)."""
    B, C, L = x.shape
    flat = x.reshape(B * C, L)
    keep = flat.abs().sum(dim=0) != 0
    return flat[:, keep].reshape(B, C, -1)


def components_from_labels(sub_ptr, sub_nodes, labels, max_sub_len=0, dims_reduce=None, dims=None):
    """cc labels (smallest position per component) -> padded (S, C, L) int64 component tensor
    in canonical order: components by their first node's position, nodes in subgraph order
    (duplicates dropped) -- sgnn_cc_compact (one statistics launch, one write launch)."""
    return ops.cc_compact(sub_ptr, sub_nodes, labels, max_sub_len, dims_reduce, dims)
