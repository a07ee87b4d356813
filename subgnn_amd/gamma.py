"""Structure-channel similarity gamma_S on the GPU (mirrors reference SubGNN/gamma.py).

``get_degree_sequence`` / ``calc_dtw`` keep the reference names and argument meaning for one
patch; the batched forms underneath are what the hot path calls (one launch for all sets,
one launch for all (component, anchor) pairs)."""
import torch

from . import ops
from .config import PAD_VALUE


def degree_sequences(graph, padded_ids, internal=True, use_degree_dict=True):
    """Batched gamma.get_degree_sequence (gamma.py:21-49): padded (rows, L) int64 ids ->
    (ptr int64[rows+1], values int32[total]) with every row's sequence sorted ascending."""
    sets = padded_ids if isinstance(padded_ids, ops.Ragged) else ops.Ragged.from_padded(padded_ids)
    di, de = ops.degree_sequence(graph, sets, sort=True, use_degree_dict=use_degree_dict,
                                 want_external=not internal)
    return sets, (di if internal else de)


def get_degree_sequence(graph, nodes, degree_dict=None, internal=True):
    """gamma.get_degree_sequence(graph, nodes, degree_dict, internal) for ONE padded id vector;
    returns the python list the reference returns.  ``degree_dict`` None -> degrees from the
    graph (gamma.py:41-45); anything else -> the graph's loaded degree table."""
    ids = torch.as_tensor(nodes, dtype=torch.int64, device=graph.device).view(1, -1)
    sets, vals = degree_sequences(graph, ids, internal, use_degree_dict=degree_dict is not None)
    n = int(sets.ptr[-1].item())
    return vals[:n].cpu().tolist()


def dtw_similarity_matrix(cc_sets, cc_seq, anchor_sets, anchor_seq, tie_order=None):
    """1/(1+fastdtw(cc, anchor, radius=1, dist=calc_dist)) for all pairs (gamma.py:51-59,
    SubGNN.py:811-822) -> (n_cc_rows, n_anchors) float32, empty (padded) rows = PAD."""
    return ops.dtw_similarity(cc_sets.ptr, cc_seq, max(cc_sets.max_len, 1),
                              anchor_sets.ptr, anchor_seq, max(anchor_sets.max_len, 1), tie_order)


def calc_dtw(graph_device, component_degree, patch_degree, tie_order=None):
    """gamma.calc_dtw for one pair of python lists (convenience; the hot path is batched)."""
    dev = graph_device
    x = ops.Ragged.from_lists([list(component_degree)], dev)
    y = ops.Ragged.from_lists([list(patch_degree)], dev)
    if len(component_degree) == 0:
        return 1.0          # fastdtw of an empty series costs 0 (DTW)
    out = ops.dtw_similarity(x.ptr, x.nodes, max(len(component_degree), 1), y.ptr, y.nodes, max(len(patch_degree), 1),
                             tie_order)
    return float(out[0, 0].item())


assert PAD_VALUE == 0
