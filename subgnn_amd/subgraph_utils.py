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


def f1_from_predictions(y_true, y_pred, average):
    """sklearn.metrics.f1_score(y_true, y_pred, average='micro' | 'macro') for integer class vectors or 0/1 indicator matrices
    (numpy arrays), from the confusion counts: F1 of a class = 2 TP / (2 TP + FP + FN), 0 where that is 0 / 0 (sklearn's
    zero_division default, minus its warning); macro = mean over the classes present in y_true or y_pred (all columns of an
    indicator matrix), micro = the same ratio over the pooled counts.  The reference calls sklearn once per validation batch
    (su:90-105): ~0.5 ms of argument checking per call -- and 10-20 ms whenever a class of a small batch has no prediction and
    the warning is formatted.  tests/test_host_logic.py checks the values against sklearn's."""
    import numpy as np
    yt, yp = np.asarray(y_true), np.asarray(y_pred)
    if yt.ndim == 2:                                     # multilabel indicator: one binary problem per column
        yt, yp = yt.astype(bool), yp.astype(bool)
        tp = (yt & yp).sum(0).astype(np.float64)
        fp = (~yt & yp).sum(0).astype(np.float64)
        fn = (yt & ~yp).sum(0).astype(np.float64)
    else:
        yt, yp = yt.reshape(-1), yp.reshape(-1)
        classes = np.union1d(yt, yp)
        tp = np.array([np.sum((yt == c) & (yp == c)) for c in classes], dtype=np.float64)
        fp = np.array([np.sum((yt != c) & (yp == c)) for c in classes], dtype=np.float64)
        fn = np.array([np.sum((yt == c) & (yp != c)) for c in classes], dtype=np.float64)
    if average == 'micro':
        tp, fp, fn = tp.sum(keepdims=True), fp.sum(keepdims=True), fn.sum(keepdims=True)
    elif average != 'macro':
        raise ValueError('f1_from_predictions: average must be micro or macro')
    den = 2 * tp + fp + fn
    f = np.where(den > 0, 2 * tp / np.where(den > 0, den, 1.0), 0.0)
    return float(f.mean()) if f.size else 0.0


def _binary_auroc(y_true, y_score):
    """Area under the ROC curve of one binary problem the way sklearn builds it: scores sorted descending (stable), the
    cumulative true / false positive counts at the LAST entry of every run of equal scores, a (0, 0) point in front, trapezoids.
    (sklearn drops collinear points first, roc_curve's drop_intermediate: the area is the same.)  nan when y_true holds one
    class only: what the sklearn of this image returns (with a warning; the reference's pinned 0.20.2 raised ValueError there
    and the reference's epoch end, which does not catch it, stopped the run)."""
    import numpy as np
    yt = np.asarray(y_true).reshape(-1)
    ys = np.asarray(y_score, dtype=np.float64).reshape(-1)
    classes = np.unique(yt)
    if classes.size != 2:
        return float('nan')
    pos = yt == classes[1]
    order = np.argsort(-ys, kind='mergesort')
    ys, pos = ys[order], pos[order]
    last = np.r_[np.nonzero(np.diff(ys))[0], ys.size - 1]               # last index of every run of equal scores
    tps = np.cumsum(pos, dtype=np.float64)[last]
    fps = 1.0 + last - tps
    tpr = np.r_[0.0, tps] / tps[-1]
    fpr = np.r_[0.0, fps] / fps[-1]
    return float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) / 2.0))


def roc_auc(y_true, y_score, multi_class=None):
    """sklearn.metrics.roc_auc_score as the reference's epoch-end metrics call it (S.py:408-444): binary (y_score (n,)),
    one-vs-rest macro average over the classes (integer y_true, y_score (n, K) class probabilities, multi_class='ovr') and
    multilabel macro average (indicator y_true (n, K)).  Seven sklearn calls per validation epoch -- one per class and one for the
    average -- were ~4 ms of argument validation on 200 subgraphs (a third of a PPI-BP stand-in epoch's host time); the values
    are checked against sklearn's in tests/test_host_logic.py.  Raises ValueError where sklearn does (a class absent from the
    split, a class count that differs from the score columns)."""
    import numpy as np
    yt, ys = np.asarray(y_true), np.asarray(y_score, dtype=np.float64)
    if ys.ndim == 1:
        return _binary_auroc(yt, ys)
    if yt.ndim == 2:                                     # multilabel indicator
        return float(np.mean([_binary_auroc(yt[:, c], ys[:, c]) for c in range(ys.shape[1])]))
    if multi_class != 'ovr':
        raise ValueError("roc_auc: multi-class scores need multi_class='ovr'")
    classes = np.unique(yt)
    if classes.size != ys.shape[1]:
        raise ValueError("Number of classes in y_true not equal to the number of columns in 'y_score'")
    if not np.allclose(1.0, ys.sum(axis=1)):
        raise ValueError('Target scores need to be probabilities for multiclass roc_auc, i.e. they should sum up to 1.0 over classes')
    return float(np.mean([_binary_auroc(yt == c, ys[:, i]) for i, c in enumerate(classes)]))


def calc_f1(logits, labels, avg_type='macro', multilabel_binarizer=None):
    """su.calc_f1 (su:90-105): sklearn's f1_score of the arg-max (multilabel: sigmoid > 0.5) predictions."""
    if multilabel_binarizer is not None:
        pred = torch.sigmoid(logits) > 0.5
    else:
        pred = torch.argmax(logits, dim=-1)
    if avg_type in ('micro', 'macro'):
        return torch.tensor([f1_from_predictions(labels.detach().cpu().numpy(), pred.detach().cpu().numpy(), avg_type)])
    from sklearn.metrics import f1_score
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
