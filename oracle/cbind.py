"""ctypes binding of oracle/_build/liboracle_c.so (test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, '_build', 'liboracle_c.so')
        subprocess.check_call(['make', '-s', '-C', _HERE])             # (a no-op when the library is newer than its source)
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def ragged(lists, dtype=np.int32):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.zeros(max(int(ptr[-1]), 1), dtype=dtype)
    for i, l in enumerate(lists):
        flat[ptr[i]:ptr[i + 1]] = l
    return ptr, flat


def degree_sequence(rowptr, col, full_degree, set_ptr, set_nodes, sorted_=True):
    n = int(set_ptr[-1])
    oi = np.zeros(max(n, 1), dtype=np.int32)
    oe = np.zeros(max(n, 1), dtype=np.int32)
    lib().oc_degree_sequence(_p(rowptr), _p(col), _p(full_degree), _p(set_ptr), _p(set_nodes),
                             ctypes.c_int64(len(set_ptr) - 1), ctypes.c_int64(len(rowptr) - 2),
                             ctypes.c_int(1 if sorted_ else 0), _p(oi), _p(oe))
    return oi[:n], oe[:n]


def fastdtw_sim(x_ptr, x_val, y_ptr, y_val, tie_order=None):
    if tie_order is None:
        from .fastdtw_restate import DEFAULT_TIE_ORDER as tie_order
    nx, ny = len(x_ptr) - 1, len(y_ptr) - 1
    out = np.zeros((nx, ny), dtype=np.float32)
    lib().oc_fastdtw_sim(_p(x_ptr), _p(x_val), ctypes.c_int64(nx), _p(y_ptr), _p(y_val), ctypes.c_int64(ny),
                         ctypes.c_int(tie_order), _p(out))
    return out


def sp_similarity(apsp, set_ptr, set_nodes):
    n = len(set_ptr) - 1
    out = np.zeros((n, apsp.shape[1]), dtype=np.float32)
    lib().oc_sp_similarity(_p(np.ascontiguousarray(apsp)), ctypes.c_int64(apsp.shape[1]), _p(set_ptr), _p(set_nodes),
                           ctypes.c_int64(n), _p(out))
    return out


def bfs_min_hops_to_sets(rowptr, col, sources, set_ptr, set_nodes):
    """(n_sets, n_sources) float32: min over a set's members of the hop count from each source, 0 where a member is
    unreachable (oc_bfs_min_hops_to_sets; one BFS per source, sources shared among the host's cores)."""
    n_sets, n_src = len(set_ptr) - 1, len(sources)
    out = np.zeros((n_sets, n_src), dtype=np.float32)
    src = np.ascontiguousarray(sources, dtype=np.int32)
    lib().oc_bfs_min_hops_to_sets(_p(rowptr), _p(col), ctypes.c_int64(len(rowptr) - 2), _p(src), ctypes.c_int64(n_src),
                                  _p(set_ptr), _p(set_nodes), ctypes.c_int64(n_sets), _p(out))
    return out
