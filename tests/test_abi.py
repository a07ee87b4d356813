"""CPU: the C-ABI library builds, loads, and exports every symbol include/subgnn_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _declared():
    txt = open(os.path.join(REPO, 'include', 'subgnn_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(sgnn_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from subgnn_amd import build, _lib
    build.build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names        # the ctypes table mirrors the header
    assert lib.sgnn_abi_version() == 11


def test_workspace_queries_run_on_the_host():
    from subgnn_amd import _lib
    lib = _lib.load()
    assert lib.sgnn_khop_border_workspace_bytes(1000, 10, 0) == 10 * (((1000 + 32) // 32) * 4 + 1001 * 4) + 16
    assert lib.sgnn_khop_border_workspace_bytes(1000, 10, 1) == 10 * 1001 * 4 + 16
    assert lib.sgnn_dtw_workspace_bytes(100, 20, 10, 50) > 0
    assert lib.sgnn_bfs_hops_workspace_bytes(1000, 70, 16) == 3 * 1001 * 2 * 8 + 18 * 8 + 18 * 4 + 8 + 2 * ((1000 + 32) // 32) * 4 + ((1000 + 4) // 4) * 4 + 8


def test_argument_errors_are_reported_not_thrown():
    from subgnn_amd import _lib
    lib = _lib.load()
    assert lib.sgnn_degree_sequence(None, None, 0, None, None, None, None, 0, 1, 1, None, None, None, None) == -1
    assert lib.sgnn_mpn_fwd(None, None, None, None) == -1


def test_cpu_tensors_are_rejected():
    import torch
    from subgnn_amd import ops, _lib
    with pytest.raises(_lib.SubgnnHipError):
        ops.Ragged(torch.zeros(2, dtype=torch.int64), torch.zeros(1, dtype=torch.int32))


def test_fused_packing_limits_come_from_the_library():
    """the Python gate of the one-launch packings must not drift from the library's (device-dependent) limits."""
    from subgnn_amd import ops, _lib
    lib = _lib.load()
    rows, entries = ops.pack_fused_limits()
    assert rows == lib.sgnn_pack_fused_max_rows() and entries == lib.sgnn_pack_fused_max_entries()
    assert rows == 8192 and 1 <= entries <= 24576           # (without a device: the compile-time figures, sized for gfx950's LDS)
