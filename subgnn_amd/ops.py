"""Torch-facing wrappers of the C ABI in include/subgnn_hip.h.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every operator
below runs a hand-written HIP kernel of libsubgnn_hip.so on the current stream.  Inputs
must already live on the GPU; nothing here falls back to a CPU implementation.
"""
import ctypes
import functools
import threading

import numpy as np
import torch

from . import _lib
from ._lib import MpnArgs, check

PAD = 0


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_CUR_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """The current HIP stream's handle.  Through torch's raw accessor: ``torch.cuda.current_stream()`` builds a Stream object by
    way of five Python calls (9 us each time: a quarter of a millisecond of a 130-launch preparation that the host, not the
    device, bounds at shard size)."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.SubgnnHipError('%s must be a CUDA/HIP tensor (there is no CPU path)' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)


# ---------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------

_TRI = {}


class Ragged:
    """CSR-style node sets on the device: ptr int64[n+1], nodes int32[total]."""

    def __init__(self, ptr, nodes, max_len=None):
        _req(ptr, torch.int64, 'ptr')
        _req(nodes, torch.int32, 'nodes')
        self.ptr, self.nodes = ptr, nodes
        self.n = ptr.numel() - 1
        self._max_len = max_len

    @property
    def max_len(self):
        if self._max_len is None:
            self._max_len = int((self.ptr[1:] - self.ptr[:-1]).max().item()) if self.n > 0 else 0
        return self._max_len

    @property
    def lengths(self):
        return self.ptr[1:] - self.ptr[:-1]

    @property
    def total(self):
        return self.nodes.numel() if self.n == 0 else int(self.ptr[-1].item())

    @staticmethod
    def from_padded(ids):
        """(rows, L) padded int64 ids -> ragged, PAD stripped, order kept (gamma.py:27,
        aps:131, S.py:769 all strip PAD this way)."""
        assert ids.dim() == 2
        return Ragged.from_mask(ids, None)

    @staticmethod
    def from_first_occurrence(ids):
        """The node view of every row: its non-PAD ids without repeats, first occurrences kept in order (aps:131-138)."""
        return Ragged.from_mask(ids, None, first_occurrence=True)

    @staticmethod
    def from_mask(ids, mask, first_occurrence=False):
        """The masked entries of every row of a padded (rows, L) matrix, packed, order kept -- without
        a host round trip: the packed position of an entry is its row's offset plus its rank among the
        row's kept entries, dropped entries go to one spare slot behind the data (the node array is an
        arena of rows*L + 1 entries; ``ptr`` says what is live)."""
        assert ids.dim() == 2 and (mask is None or mask.shape == ids.shape)
        n, L = ids.shape
        dev = ids.device
        ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if n == 0 or L == 0:
            return Ragged(ptr, torch.zeros(1, dtype=torch.int32, device=dev), max_len=L)
        if ids.is_cuda and ids.dtype == torch.int64:
            # counts per row, prefix sum, packed write (sgnn_pack_rows_count / _write): mask None = strip PAD
            lib = _lib.load()
            ids = ids.contiguous()
            fused_rows, fused_entries = pack_fused_limits()
            if n <= fused_rows and n * L <= fused_entries and (mask is None or mask.dtype == torch.uint8 or first_occurrence):
                # a few hundred rows (the structure patches): one launch does count, scan, write and the arena's zero tail
                m8 = None if (mask is None or first_occurrence) else mask.contiguous()
                ptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
                nodes = torch.empty(n * L + 1, dtype=torch.int32, device=dev)
                check(lib.sgnn_pack_rows_fused(_ptr(ids), _ptr(m8), 2 if first_occurrence else (0 if m8 is None else 1), n, L,
                                               _ptr(ptr), _ptr(nodes), _stream()), 'sgnn_pack_rows_fused')
                return Ragged(ptr, nodes, max_len=L)
            if first_occurrence:
                mask = first_occurrence_mask(ids)
            m8 = None if mask is None else mask.to(torch.uint8).contiguous()
            counts = torch.empty(n, dtype=torch.int64, device=dev)
            check(lib.sgnn_pack_rows_count(_ptr(ids), _ptr(m8), n, L, _ptr(counts), _stream()), 'sgnn_pack_rows_count')
            torch.cumsum(counts, 0, out=ptr[1:])
            nodes = torch.zeros(n * L + 1, dtype=torch.int32, device=dev)      # an arena: ptr says what is live
            check(lib.sgnn_pack_rows_write(_ptr(ids), _ptr(m8), n, L, _ptr(ptr), _ptr(nodes), _stream()), 'sgnn_pack_rows_write')
            return Ragged(ptr, nodes, max_len=L)
        if first_occurrence:
            Lq = ids.shape[1]
            earlier = torch.ones(Lq, Lq, dtype=torch.bool, device=dev).tril(-1)
            mask = ~((ids.unsqueeze(2) == ids.unsqueeze(1)) & earlier.unsqueeze(0)).any(dim=2) & (ids != PAD)
        if mask is None:
            mask = ids != PAD
        # rank of an entry among its row's kept entries.  torch's scan along the innermost dimension of
        # a (50k, 20) matrix takes 0.15 ms; for the short rows of this path the same numbers come out of
        # one small GEMM with a triangular matrix (counts < 2^24 are exact in fp32)
        if L <= 256:
            tri = _TRI.get((L, dev))
            if tri is None:
                tri = _TRI[(L, dev)] = torch.ones(L, L, dtype=torch.float32, device=dev).triu()
            rank = (mask.to(torch.float32) @ tri).to(torch.int64)
        else:
            rank = torch.cumsum(mask, dim=1)
        torch.cumsum(rank[:, -1], 0, out=ptr[1:])
        dst = torch.where(mask, ptr[:-1].view(-1, 1) + rank - 1, n * L)
        nodes = torch.zeros(n * L + 1, dtype=torch.int32, device=dev)
        nodes.scatter_(0, dst.reshape(-1), ids.reshape(-1).to(torch.int32))
        return Ragged(ptr, nodes, max_len=L)

    @staticmethod
    def from_lists(lists, device):
        """One concatenated array + offsets, built without a Python loop over the lists (50k subgraph lists: the loop was
        ~150 ms of the cold first pass), uploaded with two copies."""
        import itertools
        n = len(lists)
        lens = np.fromiter(map(len, lists), dtype=np.int64, count=n)
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=ptr[1:])
        total = int(ptr[-1])
        if total:
            try:
                flat = np.fromiter(itertools.chain.from_iterable(lists), dtype=np.int32, count=total)
            except (TypeError, ValueError):                 # lists of arrays / tensors: element-wise iteration does not apply
                flat = np.concatenate([np.asarray(l, dtype=np.int32).reshape(-1) for l in lists])
        else:
            flat = np.zeros(1, dtype=np.int32)
        ml = int(lens.max()) if n else 0
        return Ragged(torch.from_numpy(ptr).to(device), torch.from_numpy(flat).to(device), max_len=ml)

    def to_padded(self, width=None, fill=PAD, dtype=torch.int64):
        """(n, width) matrix, rows left-aligned, ``fill`` behind them.  A gather (entry (i, j) reads
        nodes[ptr[i] + j] where j < len_i): no data-dependent sizes, so no host round trip when
        ``width`` is given."""
        width = self.max_len if width is None else width
        dev = self.ptr.device
        if self.n == 0 or width <= 0 or self.nodes.numel() == 0:
            return torch.full((self.n, max(width, 0)), fill, dtype=dtype, device=dev)
        j = torch.arange(width, device=dev).view(1, -1)
        idx = (self.ptr[:-1].view(-1, 1) + j).clamp_(max=self.nodes.numel() - 1)
        vals = self.nodes[idx].to(dtype)
        return torch.where(j < self.lengths.view(-1, 1), vals, torch.full_like(vals, fill))

    def to_lists(self):
        p = self.ptr.cpu().numpy()
        v = self.nodes.cpu().numpy()
        return [v[p[i]:p[i + 1]].tolist() for i in range(self.n)]


class DeviceGraph:
    """The base graph resident in HBM: CSR by node id in networkx neighbour order (walks) and
    sorted order (adjacency tests), node order / position (G.nodes() semantics), degrees."""

    def __init__(self, rowptr, col, node_order, device, full_degree=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        self.max_id = len(rowptr) - 2
        self.nnz = int(rowptr[-1])
        col_sorted = col.copy()
        # sort every row ascending (vectorised: sort by (row, value))
        rows = np.repeat(np.arange(self.max_id + 1, dtype=np.int64), np.diff(rowptr))
        order = np.lexsort((col_sorted, rows))
        col_sorted = col_sorted[order]
        node_order = np.ascontiguousarray(node_order, dtype=np.int32)
        node_pos = np.zeros(self.max_id + 1, dtype=np.int32)
        node_pos[node_order] = np.arange(len(node_order), dtype=np.int32)
        self.n_nodes = len(node_order)
        self.device = device
        self.rowptr = torch.from_numpy(rowptr).to(device)
        self.col = torch.from_numpy(col if len(col) else np.zeros(1, np.int32)).to(device)
        self.col_sorted = torch.from_numpy(col_sorted if len(col_sorted) else np.zeros(1, np.int32)).to(device)
        self.node_order = torch.from_numpy(node_order).to(device)
        self.node_pos = torch.from_numpy(node_pos).to(device)
        self.full_degree = None
        if full_degree is not None:
            self.full_degree = torch.from_numpy(np.ascontiguousarray(full_degree, dtype=np.int32)).to(device)
        # self-loop entries per row (0 or 1 on a simple graph): spares the degree-sequence kernel a
        # compare per streamed neighbour
        n_self = np.bincount(rows[col[:len(rows)] == rows], minlength=self.max_id + 1) if len(rows) else \
            np.zeros(self.max_id + 1, dtype=np.int64)
        self.self_loops = torch.from_numpy(np.minimum(n_self, 255).astype(np.uint8)).to(device)
        # no id twice in a row: what lets the degree-sequence kernel search long lists instead of streaming
        # them (a repeated entry would be counted once instead of twice)
        rs = rows[order] if len(rows) else rows
        self.simple_rows = not bool(len(col_sorted) > 1 and np.any((col_sorted[1:] == col_sorted[:-1]) & (rs[1:] == rs[:-1])))


    @classmethod
    def from_device_csr(cls, rowptr, col):
        """A graph that is already a device-resident CSR with ascending rows and node order 1..n (the
        synthetic generators): nothing is staged through the host, so graphs far beyond the host-side
        constructor's reach (a 1 GB ``col``) can be built.  ``col`` serves as ``col_sorted`` too."""
        _req(rowptr, torch.int64, 'rowptr')
        _req(col, torch.int32, 'col')
        self = cls.__new__(cls)
        dev = rowptr.device
        self.max_id = rowptr.numel() - 2
        self.nnz = int(rowptr[-1].item())
        self.n_nodes = self.max_id
        self.device = dev
        self.rowptr, self.col, self.col_sorted = rowptr, col, col
        self.node_order = torch.arange(1, self.max_id + 1, dtype=torch.int32, device=dev)
        self.node_pos = torch.zeros(self.max_id + 1, dtype=torch.int32, device=dev)
        self.node_pos[1:] = torch.arange(self.max_id, dtype=torch.int32, device=dev)
        self.full_degree = None
        deg = rowptr[1:] - rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(self.max_id + 1, dtype=torch.int32, device=dev), deg)
        n_self = torch.zeros(self.max_id + 1, dtype=torch.int64, device=dev)
        n_self.index_add_(0, rows.long(), (col[:self.nnz] == rows).to(torch.int64))
        self.self_loops = n_self.clamp_(max=255).to(torch.uint8)
        same = (col[1:self.nnz] == col[:self.nnz - 1]) & (rows[1:] == rows[:-1]) if self.nnz > 1 else None
        asc = bool(((col[1:self.nnz] >= col[:self.nnz - 1]) | (rows[1:] != rows[:-1])).all()) if self.nnz > 1 else True
        if not asc:
            raise ValueError('from_device_csr needs ascending rows')
        self.simple_rows = not bool(same.any()) if same is not None else True
        return self

    HUB_BITMAP_BYTES = 1 << 30     # budget of the long lists' membership bitmaps (the benchmark graph: 1M nodes x 380 bits = 48 MB)

    def hub_tables(self):
        """Membership bitmaps of the long neighbour lists, built on first use and kept: -> (hub_index int32 (max_id + 1,),
        hub_bits int32 (max_id + 1, W), W) or None (no list of >= sgnn_degree_sequence_search_threshold() entries, rows with
        repeated ids, or more than HUB_BITMAP_BYTES of bitmaps: the kernel then searches / streams those lists as before).
        hub_index numbers the H long lists; bit hub_index[v] of ROW x says that node id x is in v's list (W = ceil(H / 32)
        words per node: by node, so that one member's lookups against all the hubs of its set share a line): what lets
        sgnn_degree_sequence_hub_bitmaps ask "is this member a neighbour of that hub" with one load instead of a binary search
        of the hub's list.  Device-side, no host round trip beyond the hub count; a property of the graph."""
        cached = self.__dict__.get('_hub_tables', False)
        if cached is not False:
            return cached
        out = None
        if getattr(self, 'simple_rows', False) and self.nnz > 0:
            thr = int(_lib.load().sgnn_degree_sequence_search_threshold())
            deg = self.rowptr[1:] - self.rowptr[:-1]
            hubs = torch.nonzero(deg >= thr).view(-1)
            H = int(hubs.numel())
            W, rows_n = (H + 31) // 32, self.max_id + 1                                 # one row of W words per NODE id
            if 0 < H and rows_n * W * 4 <= self.HUB_BITMAP_BYTES:
                dev = self.device
                hub_index = torch.full((self.max_id + 1,), -1, dtype=torch.int32, device=dev)
                hub_index[hubs] = torch.arange(H, dtype=torch.int32, device=dev)
                hdeg = deg[hubs]
                owner = torch.repeat_interleave(torch.arange(H, dtype=torch.int64, device=dev), hdeg)
                start = torch.cumsum(hdeg, 0) - hdeg                                    # first entry of every hub in ``owner``
                pos = torch.arange(owner.numel(), dtype=torch.int64, device=dev) - start[owner] + self.rowptr[hubs][owner]
                x = self.col_sorted[pos].to(torch.int64)
                words = torch.zeros(rows_n * W, dtype=torch.int64, device=dev)
                # (no id twice in a row -- simple_rows -- so the bits of a word are distinct powers of two: their sum is their OR)
                words.index_add_(0, x * W + (owner >> 5), torch.ones_like(owner) << (owner & 31))
                out = (hub_index, words.to(torch.int32).view(rows_n, W).contiguous(), W)
        self.__dict__['_hub_tables'] = out
        return out

    def node_records(self):
        """One 16-byte record per node id for the degree-sequence kernel (with hub_tables): int32 (max_id + 1, 4) = {row start,
        degree, hub number (0xffffff: none) | self-loop entries << 24, full degree (the degree dict's, else degree + self
        loops)} -- one load per set member instead of a line each out of rowptr, hub_index, self_loops and full_degree.  None
        without hub tables.  Built on first use and kept."""
        cached = self.__dict__.get('_node_records', False)
        if cached is not False:
            return cached
        out = None
        hub = self.hub_tables()
        if hub is not None:
            deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int32)
            sl = self.self_loops.to(torch.int32)
            z = ((sl.to(torch.int64) << 24) | torch.where(hub[0] >= 0, hub[0], torch.full_like(hub[0], 0xffffff)).to(torch.int64)).to(torch.int32)
            full = self.full_degree if self.full_degree is not None else deg + sl
            out = torch.stack([self.rowptr[:-1].to(torch.int32), deg, z, full.to(torch.int32)], 1).contiguous()
        self.__dict__['_node_records'] = out
        return out


_WARM = set()


def warm_up(device=None):
    """Pay the one-time start-up costs of the path NOW (a model's constructor calls this) instead of inside the first pass:
    the code objects of libsubgnn_hip.so (sgnn_warm_up: one empty launch per translation unit), the BLAS libraries' handles and
    first kernels (the head's and the LSTM's GEMM shapes), and the handful of torch kernels the preparation uses.  Once per
    device and process; ~0.3-0.5 s the first time.  Nothing here computes a result anybody reads, and nothing here moves a
    random stream: the body runs inside ``torch.random.fork_rng`` (the CPU generator and this device's Philox offset are put
    back), so the first model of a process draws the dropout masks every later same-seed model draws."""
    if not torch.cuda.is_available():
        return 0.0
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != 'cuda' or str(dev) in _WARM:
        return 0.0
    import time
    t0 = time.perf_counter()
    _WARM.add(str(dev))
    with torch.cuda.device(dev), torch.random.fork_rng(devices=[dev]):
        check(_lib.load().sgnn_warm_up(_stream()), 'sgnn_warm_up')
        # torch kernels of the preparation (sorts, scans, gathers, scatters, reductions) on a few elements each
        i64 = torch.arange(8, device=dev)
        i32 = i64.to(torch.int32)
        f32 = torch.rand(8, 8, device=dev, requires_grad=True)
        torch.cumsum(i64, 0); torch.sort(i64); torch.argsort(i32); i64.index_select(0, i64 % 4); i64.max(); (i64 != 0).sum()
        torch.zeros(9, dtype=torch.int64, device=dev).scatter_(0, i64, i64); torch.where(i64 > 3, i64, i64 - 1)
        torch.unique(i64 % 3, return_inverse=True); torch.repeat_interleave(i64, 2); torch.stack((i64.sum(), i64.sum()))
        # the BLAS libraries (rocBLAS / hipBLASLt): handle creation + the GEMM forms the float half uses, forward and backward
        w = torch.rand(8, 8, device=dev, requires_grad=True)
        b = torch.rand(8, device=dev, requires_grad=True)
        y = torch.nn.functional.dropout(torch.relu(torch.nn.functional.linear(f32, w, b)), 0.1, True)
        y = y + torch.addmm(b, f32, w.t()) + torch.bmm(f32.view(2, 4, 8), w.view(2, 8, 4).transpose(1, 2).contiguous().transpose(1, 2)).reshape(8, 4).sum() \
            + (f32 @ b).view(-1, 1) + torch.cat([f32, w], 0)[:8]
        y.sum().backward()
        torch._foreach_norm([w.grad, b.grad])
        torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


# ---------------------------------------------------------------------------------------
# integer half
# ---------------------------------------------------------------------------------------

def first_occurrence_mask(ids):
    """(rows, L) int64 -> uint8 mask: non-PAD entries that no earlier entry of their row repeats (sgnn_first_occurrence_mask)."""
    _req(ids, torch.int64, 'ids')
    ids = ids.contiguous()
    keep = torch.empty(ids.shape, dtype=torch.uint8, device=ids.device)
    check(_lib.load().sgnn_first_occurrence_mask(_ptr(ids), ids.shape[0], ids.shape[1], _ptr(keep), _stream()),
          'sgnn_first_occurrence_mask')
    return keep


def filter_sets(sets, flags):
    """The entries of every set whose flag (uint8, aligned with sets.nodes) is set, order kept -> Ragged (sgnn_filter_sets)."""
    lib = _lib.load()
    _req(flags, torch.uint8, 'flags')
    dev = sets.ptr.device
    counts = torch.empty(sets.n, dtype=torch.int64, device=dev)
    ptr = torch.zeros(sets.n + 1, dtype=torch.int64, device=dev)
    if sets.n == 0:
        return Ragged(ptr, torch.zeros(1, dtype=torch.int32, device=dev), max_len=0)
    fused_rows, fused_entries = pack_fused_limits()
    if sets.n <= fused_rows and sets.nodes.numel() <= fused_entries:
        arena = max(sets.nodes.numel(), 1)
        ptr = torch.empty(sets.n + 1, dtype=torch.int64, device=dev)
        nodes = torch.empty(arena, dtype=torch.int32, device=dev)
        check(lib.sgnn_filter_sets_fused(_ptr(sets.ptr), _ptr(sets.nodes), _ptr(flags), sets.n, arena, _ptr(ptr), _ptr(nodes), _stream()),
              'sgnn_filter_sets_fused')
        return Ragged(ptr, nodes, max_len=sets._max_len)
    check(lib.sgnn_filter_sets(_ptr(sets.ptr), _ptr(sets.nodes), _ptr(flags), sets.n, _ptr(counts), None, None, _stream()),
          'sgnn_filter_sets')
    torch.cumsum(counts, 0, out=ptr[1:])
    nodes = torch.zeros(max(sets.nodes.numel(), 1), dtype=torch.int32, device=dev)       # an arena: ptr says what is live
    check(lib.sgnn_filter_sets(_ptr(sets.ptr), _ptr(sets.nodes), _ptr(flags), sets.n, None, _ptr(ptr), _ptr(nodes), _stream()),
          'sgnn_filter_sets')
    return Ragged(ptr, nodes, max_len=sets._max_len)


def heaviest_first(g, sets):
    """Dispatch order for set kernels whose cost is the members' total degree: heaviest sets first."""
    # (over the whole node array -- entries behind ptr[-1] are an arena's zero tail, PAD = node 0 of degree 0 -- so that no size
    # is read back: the first pass of a split waited here for everything queued before it, 43 ms on the driver's box)
    tot = sets.nodes.numel()
    deg = (g.rowptr[1:] - g.rowptr[:-1])[sets.nodes.long().clamp_(0, g.max_id)]
    csum = torch.zeros(tot + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(deg, 0, out=csum[1:])
    work = csum[sets.ptr[1:].clamp(max=tot)] - csum[sets.ptr[:-1].clamp(max=tot)]
    # descending by work = ascending by (cap - work) under the library's own stable radix sort (sgnn_sort_edges_by_key) -- the
    # first use of torch.argsort loaded another library's sort (0.2 s of the cold first pass).  Only an ORDER of dispatch:
    # works beyond the cap tie at the front.
    cap = (1 << 31) - 1
    key = (cap - work.clamp(max=cap)).to(torch.int32).contiguous()
    return sort_edges_by_key(key, cap)[1]


def degree_sequence(g, sets, sort=True, use_degree_dict=True, want_external=True, use_self_loop_table=True, order=None,
                    search_long_lists=True, hub_bitmaps=True):
    """gamma.get_degree_sequence for every set at once -> (internal, external) int32 flat
    tensors aligned with ``sets.nodes`` (each set's slice sorted ascending if ``sort``).
    ``search_long_lists``: hand the kernel the row-sorted CSR as well, so that hub lists are searched
    for the set's members instead of streamed (same results; off = stream everything).
    ``hub_bitmaps``: ... and the graph's membership bitmaps of those lists (DeviceGraph.hub_tables), so that they are not
    even searched: one bit per (member, hub list).  Same results; off = search."""
    lib = _lib.load()
    n_tot = sets.nodes.numel()
    out_i = torch.empty(n_tot, dtype=torch.int32, device=g.device)
    out_e = torch.empty(n_tot, dtype=torch.int32, device=g.device) if want_external else None
    fd = g.full_degree if use_degree_dict else None
    sl = g.self_loops if use_self_loop_table else None
    hub = g.hub_tables() if (search_long_lists and hub_bitmaps and getattr(g, 'simple_rows', False)) else None
    if hub is not None:
        # (the per-node records stand in for rowptr / hub_index / self_loops / full_degree when the self-loop table is wanted; their
        # full degree is the graph's degree dict where it has one, degree + self loops otherwise: what fd = None gives)
        rec = g.node_records() if use_self_loop_table else None
        check(lib.sgnn_degree_sequence_hub_bitmaps(_ptr(g.rowptr), _ptr(g.col), _ptr(g.col_sorted), g.nnz, _ptr(fd), _ptr(sl),
                                                   _ptr(hub[0]), _ptr(hub[1]), hub[2], _ptr(rec), 1 if use_degree_dict else 0,
                                                   _ptr(sets.ptr), _ptr(sets.nodes), sets.n, max(sets.max_len, 1),
                                                   1 if sort else 0, _ptr(out_i), _ptr(out_e), _ptr(order), _stream()),
              'sgnn_degree_sequence_hub_bitmaps')
    elif search_long_lists and getattr(g, 'simple_rows', False):
        check(lib.sgnn_degree_sequence_sorted_rows(_ptr(g.rowptr), _ptr(g.col), _ptr(g.col_sorted), g.nnz, _ptr(fd), _ptr(sl),
                                                   _ptr(sets.ptr), _ptr(sets.nodes), sets.n, max(sets.max_len, 1),
                                                   1 if sort else 0, _ptr(out_i), _ptr(out_e), _ptr(order), _stream()),
              'sgnn_degree_sequence_sorted_rows')
    else:
        check(lib.sgnn_degree_sequence(_ptr(g.rowptr), _ptr(g.col), g.nnz, _ptr(fd), _ptr(sl), _ptr(sets.ptr),
                                       _ptr(sets.nodes), sets.n, max(sets.max_len, 1), 1 if sort else 0, _ptr(out_i),
                                       _ptr(out_e), _ptr(order), _stream()), 'sgnn_degree_sequence')
    if sets.max_len > CC_LDS_MAX:
        # sets beyond the kernels' LDS tables: degrees from the workspace-backed kernel, each such slice ordered by a
        # device sort (they are few: one sort per set)
        total = int(sets.total)
        ws, wsb = _huge_ws(lib, total, g.device, 'sgnn_degree_sequence_huge_workspace_bytes')
        check(lib.sgnn_degree_sequence_huge(_ptr(g.rowptr), _ptr(g.col), g.nnz, _ptr(fd), _ptr(sets.ptr), _ptr(sets.nodes),
                                            sets.n, total, _ptr(out_i), _ptr(out_e), _ptr(ws), wsb, _stream()),
              'sgnn_degree_sequence_huge')
        if sort:
            ptr = sets.ptr.tolist()
            for s in (sets.lengths > CC_LDS_MAX).nonzero().view(-1).tolist():
                a, b = ptr[s], ptr[s + 1]
                out_i[a:b] = torch.sort(out_i[a:b]).values
                if out_e is not None:
                    out_e[a:b] = torch.sort(out_e[a:b]).values
    return out_i, out_e


_PACK_FUSED = {}


def pack_fused_limits():
    """(rows, entries) one fused packing launch serves on the current device -- asked of the library
    (sgnn_pack_fused_max_rows / _max_entries: the entries follow the device's LDS), once per device."""
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    lim = _PACK_FUSED.get(dev)
    if lim is None:
        lib = _lib.load()
        lim = _PACK_FUSED[dev] = (int(lib.sgnn_pack_fused_max_rows()), int(lib.sgnn_pack_fused_max_entries()))
    return lim


CC_LDS_MAX = 2048          # csrc/graph_sets.hip CC_MAX / PB_MAX: sets up to here keep their tables in LDS


def _huge_ws(lib, total, device, query='sgnn_cc_huge_workspace_bytes'):
    wsb = getattr(lib, query)(int(total))
    return torch.empty(wsb // 4 + 1, dtype=torch.int32, device=device), wsb


def cc_labels(g, subs):
    """Component label (smallest position in the component) per subgraph position.  Subgraphs of more than 2048 nodes
    take the workspace-backed kernel (sgnn_cc_labels_huge: tables in HBM instead of LDS, work ~ the members' degrees)."""
    lib = _lib.load()
    out = torch.empty(subs.nodes.numel(), dtype=torch.int32, device=g.device)
    check(lib.sgnn_cc_labels(_ptr(g.rowptr), _ptr(g.col_sorted), g.nnz, _ptr(subs.ptr), _ptr(subs.nodes), subs.n,
                             int(subs.max_len), _ptr(out), _stream()), 'sgnn_cc_labels')
    if subs.max_len > CC_LDS_MAX:
        total = int(subs.total)
        ws, wsb = _huge_ws(lib, total, g.device)
        check(lib.sgnn_cc_labels_huge(_ptr(g.rowptr), _ptr(g.col_sorted), g.nnz, _ptr(subs.ptr), _ptr(subs.nodes), subs.n, total,
                                      _ptr(out), _ptr(ws), wsb, _stream()), 'sgnn_cc_labels_huge')
    return out


def cc_compact(sub_ptr, sub_nodes, labels, max_sub_len=0, dims_reduce=None, dims=None):
    """cc labels -> the padded (S, C, L) int64 component tensor of initialize_cc_ids, canonical order
    (components by their first node's position, nodes in subgraph order, duplicates dropped).
    ``dims_reduce``: callable applied to the device tensor [C, L] before it is read -- under data
    parallelism the padded shape is a property of ALL ranks' subgraphs (dist.all_reduce_max_).
    ``dims``: the padded (C, L) of THESE subgraphs when the caller already knows it (it is a property of the subgraph
    lists: hotpath keeps it per split) -- no statistics launch, no host round trip."""
    lib = _lib.load()
    _req(labels, torch.int32, 'labels')
    S = sub_ptr.numel() - 1
    dev = sub_ptr.device
    huge = max_sub_len <= 0 or max_sub_len > CC_LDS_MAX      # subgraphs beyond the LDS tables: the workspace-backed kernels too
    if huge and max_sub_len <= 0:
        huge = S > 0 and int((sub_ptr[1:] - sub_ptr[:-1]).max().item()) > CC_LDS_MAX
    if huge:
        total = int(sub_ptr[-1].item())
        ws, wsb = _huge_ws(lib, total, dev)
    if dims is not None:
        C, L = int(dims[0]), int(dims[1])
    else:
        stats = torch.zeros((2, max(S, 1)), dtype=torch.int32, device=dev)
        check(lib.sgnn_cc_compact_stats(_ptr(sub_ptr), _ptr(sub_nodes), _ptr(labels), S, int(max_sub_len), _ptr(stats[0]),
                                        _ptr(stats[1]), _stream()), 'sgnn_cc_compact_stats')
        if huge:
            check(lib.sgnn_cc_compact_huge(_ptr(sub_ptr), _ptr(sub_nodes), _ptr(labels), S, total, 0, 0, 0, _ptr(stats[0]),
                                           _ptr(stats[1]), None, _ptr(ws), wsb, _stream()), 'sgnn_cc_compact_huge')
        d = stats.amax(dim=1)
        if dims_reduce is not None:
            d = dims_reduce(d)
        C, L = (max(int(v), 1) for v in d.tolist())                       # the one host round trip
    out = torch.zeros((S, C, L), dtype=torch.int64, device=dev)
    check(lib.sgnn_cc_compact(_ptr(sub_ptr), _ptr(sub_nodes), _ptr(labels), S, int(max_sub_len), C, L, _ptr(out),
                              _stream()), 'sgnn_cc_compact')
    if huge:
        check(lib.sgnn_cc_compact_huge(_ptr(sub_ptr), _ptr(sub_nodes), _ptr(labels), S, total, 1, C, L, None, None, _ptr(out),
                                       _ptr(ws), wsb, _stream()), 'sgnn_cc_compact_huge')
    return out


_KHOP_WS = {}


def _khop_ws(lib, g, n_sets, lds):
    """Workspace of the border BFS, kept across calls with the same layout (the kernel leaves the
    bitmaps zeroed, so the buffer is zero-filled once when it is created; the LDS variant needs no
    initialisation).  The layout -- per-workgroup bitmaps first, then the queues -- depends on the
    graph size and the workgroup count, and a queue region is not zero: a call with another
    layout gets a fresh buffer."""
    ws_bytes = lib.sgnn_khop_border_workspace_bytes(g.max_id, n_sets, 1 if lds else 0)
    key = (str(g.device), bool(lds))
    layout = (int(g.max_id), int(ws_bytes))
    hit = _KHOP_WS.get(key)
    if hit is None or hit[0] != layout:
        hit = (layout, torch.zeros(ws_bytes // 4 + 1, dtype=torch.int32, device=g.device))
        _KHOP_WS[key] = hit
    return hit[1], ws_bytes


def khop_border(g, sets, k, ego_dict_mode=False, want_hops=False, bitmap_in_lds=None):
    """k-hop border of every set -> Ragged (discovery order) [+ uint8 hop level per entry]."""
    lib = _lib.load()
    lds = bool(lib.sgnn_khop_border_bitmap_fits_lds(g.max_id)) if bitmap_in_lds is None else bool(bitmap_in_lds)
    ws, ws_bytes = _khop_ws(lib, g, sets.n, lds)
    counts = torch.zeros(sets.n, dtype=torch.int64, device=g.device)
    check(lib.sgnn_khop_border(_ptr(g.rowptr), _ptr(g.col), g.nnz, g.max_id, _ptr(sets.ptr), _ptr(sets.nodes), sets.n,
                               k, 1 if ego_dict_mode else 0, _ptr(counts), None, None, None, _ptr(ws), ws_bytes,
                               1 if lds else 0, _stream()), 'sgnn_khop_border(count)')
    ptr = torch.zeros(sets.n + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(counts, 0, out=ptr[1:])
    total = int(ptr[-1].item())
    nodes = torch.zeros(max(total, 1), dtype=torch.int32, device=g.device)
    hops = torch.zeros(max(total, 1), dtype=torch.uint8, device=g.device) if want_hops else None
    check(lib.sgnn_khop_border(_ptr(g.rowptr), _ptr(g.col), g.nnz, g.max_id, _ptr(sets.ptr), _ptr(sets.nodes), sets.n,
                               k, 1 if ego_dict_mode else 0, None, _ptr(ptr), _ptr(nodes), _ptr(hops), _ptr(ws),
                               ws_bytes, 1 if lds else 0, _stream()), 'sgnn_khop_border(fill)')
    r = Ragged(ptr, nodes)
    return (r, hops) if want_hops else r


def khop_border_one_pass(g, sets, bitmap_in_lds=None):
    """1-hop borders written by the BFS itself into slices of one arena sized by the bound
    sum of the members' degrees (no count pass).  Returns (arena int32, slice offsets int64[n+1],
    counts int64[n]): set s = arena[off[s] : off[s] + counts[s]], discovery order."""
    lib = _lib.load()
    lds = bool(lib.sgnn_khop_border_bitmap_fits_lds(g.max_id)) if bitmap_in_lds is None else bool(bitmap_in_lds)
    ws, ws_bytes = _khop_ws(lib, g, sets.n, lds)
    counts = torch.zeros(sets.n, dtype=torch.int64, device=g.device)
    tot = int(sets.ptr[-1].item())
    deg = (g.rowptr[1:] - g.rowptr[:-1])[sets.nodes[:tot].long()]
    csum = torch.zeros(tot + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(deg, 0, out=csum[1:])
    bound = csum[sets.ptr[1:]] - csum[sets.ptr[:-1]]                      # capacity of each slice
    off = torch.zeros(sets.n + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(bound, 0, out=off[1:])
    arena = torch.empty(max(int(off[-1].item()), 1), dtype=torch.int32, device=g.device)
    check(lib.sgnn_khop_border_arena(_ptr(g.rowptr), _ptr(g.col), g.nnz, g.max_id, _ptr(sets.ptr), _ptr(sets.nodes),
                                     sets.n, 1, _ptr(off), _ptr(arena), _ptr(counts), _ptr(ws), ws_bytes,
                                     1 if lds else 0, _stream()), 'sgnn_khop_border_arena')
    return arena, off, counts


def khop_border_sample(g, sets, k, n_slots, seed, stream_id, bitmap_in_lds=None, order=None, item_base=0,
                       count_reduce=None, width=None):
    """k-hop border BFS + neighbourhood-border anchor draw without a padded border matrix.  Returns
    anchors (n_sets, n_slots) int64 with the reference's PAD rule applied, their hop levels as
    float32 similarities (0 on PAD) and the border sizes.  One fused kernel: the draw is a rank
    query on the BFS's visited bitmap (the border is never materialised or sorted).
    ``bitmap_in_lds``: None = LDS (id ranges beyond the LDS bitmap are processed in slices when k = 1, else
    the bitmap moves to the workspace); False = workspace; an int > 1 = LDS bytes the bitmap may take."""
    lib = _lib.load()
    mode = 1 if bitmap_in_lds is None else (int(bitmap_in_lds) if not isinstance(bitmap_in_lds, bool) else (1 if bitmap_in_lds else 0))
    ws_bytes = lib.sgnn_khop_border_sample_workspace_bytes(g.max_id, sets.n, k, 1, mode)
    if ws_bytes <= 16:
        ws = _KHOP_WS.get(('k1', str(g.device)))
        if ws is None:
            ws = _KHOP_WS[('k1', str(g.device))] = torch.zeros(4, dtype=torch.int32, device=g.device)
    else:
        ws, ws_bytes = _khop_ws(lib, g, sets.n, mode != 0 and bool(lib.sgnn_khop_border_bitmap_fits_lds(g.max_id)))
    counts = torch.zeros(sets.n, dtype=torch.int64, device=g.device)
    anchor = torch.empty((sets.n, n_slots), dtype=torch.int64, device=g.device)
    hop = torch.empty((sets.n, n_slots), dtype=torch.uint8, device=g.device)
    allneg = torch.empty((sets.n, n_slots), dtype=torch.uint8, device=g.device)
    check(lib.sgnn_khop_border_sample(_ptr(g.rowptr), _ptr(g.col), _ptr(g.col_sorted), g.nnz, g.max_id, _ptr(sets.ptr),
                                      _ptr(sets.nodes), sets.n, k, n_slots, seed, stream_id, int(item_base), _ptr(anchor),
                                      _ptr(hop), _ptr(allneg), _ptr(counts), _ptr(order), _ptr(ws), ws_bytes, mode, _stream()),
          'sgnn_khop_border_sample')
    # aps:190: padded columns hold 0, so PAD wins when every real variate is negative and the
    # padded row (width = the largest border) has at least one PAD column
    # (``count_reduce``: applied to the device scalar holding the largest border when these sets are one shard
    # of the matrix -- the padded width is a property of ALL rows: dist.all_reduce_max_)
    # ``width``: that maximum as a device scalar when the caller has kept it (border sizes depend on the sets and the
    # graph only, not on the draw: hotpath keeps it per split after the first pass -- no reduction launch, and under
    # data parallelism no collective, in later passes)
    if width is None:
        width = counts.max().view(1)
        if count_reduce is not None:
            width = count_reduce(width)
    sims = torch.empty((sets.n, n_slots), dtype=torch.float32, device=g.device)
    check(lib.sgnn_khop_sample_finish(_ptr(anchor), _ptr(hop), _ptr(allneg), _ptr(counts), _ptr(width.to(torch.int64)), sets.n, n_slots,
                                      _ptr(sims), _stream()), 'sgnn_khop_sample_finish')
    return anchor, sims, counts


SORT_SETS_MAX = 1024


def sort_ragged(r, extra=None):
    """Canonical (ascending) order inside every set.  Sets known to be small (components, border
    sets of one subgraph: ``max_len`` given by the producer, <= 1024) take the per-set rank sort
    (sgnn_sort_sets, one launch, no host round trip); anything else one device-wide sort of
    (set, id) keys."""
    if r._max_len is not None and r._max_len <= SORT_SETS_MAX and r.nodes.numel() > 0 and r.n > 0:
        lib = _lib.load()
        _req(r.ptr, torch.int64, 'ptr')
        _req(r.nodes, torch.int32, 'nodes')
        out = torch.zeros_like(r.nodes)
        pos = torch.zeros_like(r.nodes) if extra is not None else None
        check(lib.sgnn_sort_sets(_ptr(r.ptr), _ptr(r.nodes), r.n, r._max_len, _ptr(out), _ptr(pos), _stream()),
              'sgnn_sort_sets')
        res = Ragged(r.ptr, out, r._max_len)
        if extra is not None:
            tot = int(r.ptr[-1].item())
            return res, extra[pos[:tot].long()].contiguous()
        return res
    tot = int(r.ptr[-1].item())
    if tot == 0:
        return (r, extra) if extra is not None else r
    rows = torch.repeat_interleave(torch.arange(r.n, device=r.ptr.device), r.lengths)
    key = rows * (1 << 32) + r.nodes[:tot].to(torch.int64)
    key, order = torch.sort(key)
    nodes = (key & 0xFFFFFFFF).to(torch.int32)
    out = Ragged(r.ptr, nodes.contiguous(), r._max_len)
    if extra is not None:
        return out, extra[:tot][order].contiguous()
    return out


def canonical_rows(ids):
    """Padded id rows in the canonical form of the neighbourhood-anchor law: non-PAD entries
    ascending, PADs last."""
    big = torch.iinfo(torch.int64).max
    srt = torch.sort(torch.where(ids == 0, torch.full_like(ids, big), ids), dim=1).values
    return torch.where(srt == big, torch.zeros_like(srt), srt).contiguous()


def sample_anchors_padded(ids, n_slots, seed, stream_id, canonical=False):
    """sample_neighborhood_anchor_patch on a padded (rows, L) id matrix -> (rows, n_slots).
    ``canonical``: the rows are already ascending with PADs last."""
    lib = _lib.load()
    _req(ids, torch.int64, 'ids')
    rows, L = ids.shape
    if not canonical and rows * L > 0:
        ids = canonical_rows(ids)
    out = torch.empty((rows, n_slots), dtype=torch.int64, device=ids.device)
    check(lib.sgnn_sample_anchors_padded(_ptr(ids), rows, L, n_slots, seed, stream_id, _ptr(out), _stream()),
          'sgnn_sample_anchors_padded')
    return out


def sample_anchors_ragged(sets, n_slots, seed, stream_id, row_has_pad=None, canonical=False, item_base=0):
    """Same draw on ragged sets (``canonical``: every set is already ascending).  ``item_base``: number of
    the first set within the whole matrix when ``sets`` is a shard of its rows."""
    lib = _lib.load()
    _req(row_has_pad, torch.uint8, 'row_has_pad')
    if not canonical:
        sets = sort_ragged(sets)
    out = torch.empty((sets.n, n_slots), dtype=torch.int64, device=sets.ptr.device)
    check(lib.sgnn_sample_anchors_ragged(_ptr(sets.ptr), _ptr(sets.nodes), sets.n, _ptr(row_has_pad), n_slots, seed,
                                         stream_id, int(item_base), _ptr(out), _stream()), 'sgnn_sample_anchors_ragged')
    return out


def choice_ragged(sets, n_draws, seed, stream_id, item_base=0):
    lib = _lib.load()
    out = torch.empty((sets.n, n_draws), dtype=torch.int64, device=sets.ptr.device)
    check(lib.sgnn_choice_ragged(_ptr(sets.ptr), _ptr(sets.nodes), sets.n, n_draws, seed, stream_id, int(item_base),
                                 _ptr(out), _stream()), 'sgnn_choice_ragged')
    return out


def triangular_walks(g, mode, n_items, walk_len, beta, seed, stream_id, patches=None, in_border=None,
                     walks_per_patch=1, kernel=0, item_base=0):
    """mode 0 'graph' / 1 'inside' / 2 'border' -> (n_items, walk_len) int64, PAD filled.
    kernel: 0 = pick by graph size, 1 = the wavefront-per-walk kernel (same walks).
    item_base: global number of this call's first walk (a share of a larger launch: same draws)."""
    lib = _lib.load()
    out = torch.empty((n_items, walk_len), dtype=torch.int64, device=g.device)
    check(lib.sgnn_triangular_walks(_ptr(g.rowptr), _ptr(g.col), _ptr(g.col_sorted), g.nnz, _ptr(g.node_order),
                                    g.n_nodes, _ptr(patches.ptr) if patches else None,
                                    _ptr(patches.nodes) if patches else None,
                                    _ptr(in_border.ptr) if in_border else None,
                                    _ptr(in_border.nodes) if in_border else None,
                                    mode, n_items, walks_per_patch, walk_len, float(beta), seed, stream_id, int(item_base),
                                    g.max_id, int(kernel), _ptr(out), _stream()), 'sgnn_triangular_walks')
    return out


def triangular_walks_both(g, n_items, walk_len, beta, seed, stream_id_int, stream_id_bor, patches, in_border, walks_per_patch,
                          item_base=0):
    """The internal and the border walks over the same patches in ONE launch -> (2, n_items, walk_len) int64, [0] internal,
    [1] border: what triangular_walks(mode 1) and triangular_walks(mode 2) return.  Falls back to those two calls where the
    graph's id bitmap does not fit LDS."""
    lib = _lib.load()
    out = torch.empty((2, n_items, walk_len), dtype=torch.int64, device=g.device)
    rc = lib.sgnn_triangular_walks_both(_ptr(g.rowptr), _ptr(g.col), _ptr(g.col_sorted), g.nnz, _ptr(patches.ptr), _ptr(patches.nodes),
                                        _ptr(in_border.ptr), _ptr(in_border.nodes), n_items, walks_per_patch, walk_len, float(beta),
                                        seed, stream_id_int, stream_id_bor, int(item_base), g.max_id, _ptr(out), _stream())
    if rc == -2:                                                        # SGNN_ERR_SET_TOO_LARGE: no LDS bitmap for this graph
        out[0] = triangular_walks(g, 1, n_items, walk_len, beta, seed, stream_id_int, patches=patches,
                                  walks_per_patch=walks_per_patch, item_base=item_base)
        out[1] = triangular_walks(g, 2, n_items, walk_len, beta, seed, stream_id_bor, patches=patches, in_border=in_border,
                                  walks_per_patch=walks_per_patch, item_base=item_base)
        return out
    check(rc, 'sgnn_triangular_walks_both')
    return out


def patch_in_border(g, patches):
    """uint8 flag per patch node: is it an in-border node (su.get_border_nodes semantics)."""
    lib = _lib.load()
    out = torch.zeros(patches.nodes.numel(), dtype=torch.uint8, device=g.device)
    check(lib.sgnn_patch_in_border(_ptr(g.rowptr), _ptr(g.col), g.nnz, _ptr(g.node_order), _ptr(g.node_pos),
                                   g.n_nodes, _ptr(patches.ptr), _ptr(patches.nodes), patches.n, _ptr(out), _stream()),
          'sgnn_patch_in_border')
    if patches.max_len > CC_LDS_MAX:                          # patches beyond the LDS table: membership table in HBM
        total = int(patches.total)
        ws, wsb = _huge_ws(lib, total, g.device, 'sgnn_patch_in_border_huge_workspace_bytes')
        check(lib.sgnn_patch_in_border_huge(_ptr(g.rowptr), _ptr(g.col), g.nnz, _ptr(g.node_order), _ptr(g.node_pos),
                                            g.n_nodes, _ptr(patches.ptr), _ptr(patches.nodes), patches.n, total, _ptr(out),
                                            _ptr(ws), wsb, _stream()), 'sgnn_patch_in_border_huge')
    return out


def sp_similarity_dense(apsp, sets):
    lib = _lib.load()
    _req(apsp, torch.float64, 'apsp')
    n_cols = apsp.shape[1]
    out = torch.empty((sets.n, n_cols), dtype=torch.float32, device=apsp.device)
    check(lib.sgnn_sp_similarity_dense(_ptr(apsp), n_cols, _ptr(sets.ptr), _ptr(sets.nodes), sets.n, _ptr(out),
                                       _stream()), 'sgnn_sp_similarity_dense')
    return out


def probe_stream_copy(src, dst, bytes_per_lane):
    """Measurement aid: copy ``src`` to ``dst`` (same byte size) with 4 or 16 bytes per lane."""
    nbytes = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == nbytes and src.is_cuda and dst.is_cuda
    check(_lib.load().sgnn_probe_stream_copy(_ptr(src), _ptr(dst), nbytes, int(bytes_per_lane), _stream()),
          'sgnn_probe_stream_copy')


def bfs_hops(g, sources, max_hops=64, node_major=False, pull_alpha=-1):
    """uint8 hop counts (255 = not reached) by multi-source BFS: (n_sources, max_id+1), or
    (max_id+1, n_sources) when ``node_major`` (coalesced for min_hops_to_sets).
    pull_alpha: direction switch -- pull once frontier edges * alpha > all edges (-1 = default 32, 0 = always push);
    results do not depend on it."""
    lib = _lib.load()
    _req(sources, torch.int32, 'sources')
    ns = sources.numel()
    shape = (g.max_id + 1, ns) if node_major else (ns, g.max_id + 1)
    dist = torch.empty(shape, dtype=torch.uint8, device=g.device)
    wsb = lib.sgnn_bfs_hops_workspace_bytes(g.max_id, ns, max_hops)
    ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=g.device)
    check(lib.sgnn_bfs_hops(_ptr(g.rowptr), _ptr(g.col), g.nnz, g.max_id, _ptr(sources), ns, max_hops,
                            1 if node_major else 0, int(pull_alpha), _ptr(dist), _ptr(ws), wsb, _stream()), 'sgnn_bfs_hops')
    return dist


def bfs_min_hops_to_sets(g, sources, sets, max_hops=64, want_status=False, pull_alpha=-1, push_levels=-1):
    """min over the members of every set of the hop distance from every source -> (n_sets, n_sources)
    float32, 0 for unreachable pairs; one multi-source BFS, no (sources x nodes) hop table.
    ``want_status``: also an int32[4] device tensor -- [0] the last level that found anything, [1] whether level
    ``max_hops`` itself still did (too few levels enqueued: the result may be incomplete), [2] the first level that pulled.
    ``push_levels``: levels that may still push (each costs a second launch); beyond them every level pulls.  -1 = all;
    results do not depend on it."""
    lib = _lib.load()
    _req(sources, torch.int32, 'sources')
    ns = sources.numel()
    out = torch.empty((sets.n, ns), dtype=torch.float32, device=g.device)
    status = torch.zeros(4, dtype=torch.int32, device=g.device) if want_status else None
    wsb = lib.sgnn_bfs_min_hops_workspace_bytes(g.max_id, ns, max_hops, sets.n)
    ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=g.device)
    check(lib.sgnn_bfs_min_hops_to_sets(_ptr(g.rowptr), _ptr(g.col), g.nnz, g.max_id, _ptr(sources), ns, max_hops,
                                        int(pull_alpha), int(push_levels), _ptr(sets.ptr), _ptr(sets.nodes), sets.n, _ptr(out), _ptr(status), _ptr(ws), wsb,
                                        _stream()), 'sgnn_bfs_min_hops_to_sets')
    return (out, status) if want_status else out


def min_hops_to_sets(dist, sets, node_major=False):
    lib = _lib.load()
    _req(dist, torch.uint8, 'dist')
    if node_major:
        n_ids, ns = dist.shape
    else:
        ns, n_ids = dist.shape
    out = torch.empty((sets.n, ns), dtype=torch.float32, device=dist.device)
    check(lib.sgnn_min_hops_to_sets(_ptr(dist), ns, n_ids - 1, 1 if node_major else 0, _ptr(sets.ptr), _ptr(sets.nodes),
                                    sets.n, _ptr(out), _stream()), 'sgnn_min_hops_to_sets')
    return out


_HASH_COEF = {}


def _row_representatives(rows):
    """For every row of an (n, w) integer matrix the index of one representative row with the same
    content, without a host round trip: a 64-bit row hash is sorted, runs of equal hashes are the
    groups, the first row of a run (in sorted order) represents it.  Exactness does not rest on the
    hash: a row that differs from its representative (a collision) represents itself."""
    n, w = rows.shape
    dev = rows.device
    coef = _HASH_COEF.get((w, dev))
    if coef is None:                                  # uploaded once per width (a blocking host->device copy)
        g = torch.Generator(device='cpu').manual_seed(0x5DEECE66D)
        coef = _HASH_COEF[(w, dev)] = (torch.randint(-(1 << 62), 1 << 62, (w,), generator=g, dtype=torch.int64) | 1).to(dev)
    h = (rows.to(torch.int64) * coef).sum(dim=1)                          # wraps modulo 2^64
    hs, perm = torch.sort(h)
    pos = torch.arange(n, device=dev)
    first = torch.ones(n, dtype=torch.bool, device=dev)
    if n > 1:
        first[1:] = hs[1:] != hs[:-1]
    # run start (sorted order) of every entry: the run number is a prefix sum of the run-start flags; the
    # start position of run g is scattered from its first entry (the others write to a spare slot) --
    # torch.cummax over one long row is a serial scan (0.13 ms for 50k)
    gid = torch.cumsum(first, 0) - 1
    run_start = torch.empty(n + 1, dtype=torch.int64, device=dev)
    run_start.scatter_(0, torch.where(first, gid, n), pos)
    start = run_start[gid]
    rep = torch.empty(n, dtype=torch.int64, device=dev)
    rep[perm] = perm[start]
    same = (rows.index_select(0, rep) == rows).all(dim=1)
    return torch.where(same, rep, pos)


def _unique_rows(rows):
    """torch.unique(rows, dim=0, return_inverse=True) up to the order of the unique rows (host
    round trip for the count; the DTW path uses _row_representatives and stays on the device)."""
    rep = _row_representatives(rows)
    is_rep = rep == torch.arange(rows.shape[0], device=rows.device)
    gid = torch.cumsum(is_rep, 0) - 1
    return rows[is_rep], gid[rep]


def distinct_row_fraction(x_ptr, x_val, max_x):
    """Share of the x rows that are distinct (one host round trip): what decides whether grouping
    repeated rows before the DTW pays.  A property of the split's components and the graph -- callers
    that run the same rows every pass (hotpath.prepare_sparse) ask once and keep the answer."""
    n = x_ptr.numel() - 1
    if n <= 1024 or max_x > 64:
        return 1.0
    rows = Ragged(x_ptr, x_val, max_len=max_x).to_padded(width=max_x, fill=-1, dtype=torch.int32)
    rep = _row_representatives(rows)
    return float((rep == torch.arange(n, device=rows.device)).sum().item()) / n


def distinct_rows_async(x_ptr, x_val, max_x):
    """distinct_row_fraction without the wait: the count of distinct rows travels to pinned host memory behind the launches
    queued here -> (pinned int64 (1,), event, n rows) or None where the answer is known (few rows / long rows: no grouping).
    ``distinct_rows_ready`` reads it once the copy has landed."""
    n = x_ptr.numel() - 1
    if n <= 1024 or max_x > 64:
        return None
    rows = Ragged(x_ptr, x_val, max_len=max_x).to_padded(width=max_x, fill=-1, dtype=torch.int32)
    rep = _row_representatives(rows)
    cnt = (rep == torch.arange(n, device=rows.device)).sum().view(1)
    host = torch.empty(1, dtype=torch.int64).pin_memory()
    host.copy_(cnt, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev, n


def distinct_rows_ready(pending, wait=False):
    """-> fraction of distinct rows, or None while the copy of ``distinct_rows_async`` is still in flight (``wait``: block)."""
    if pending is None:
        return 1.0
    host, ev, n = pending
    if wait:
        ev.synchronize()
    elif not ev.query():
        return None
    return float(int(host[0])) / n


def dtw_similarity(x_ptr, x_val, max_x, y_ptr, y_val, max_y, tie_order=None, order_rows=True, dedupe=True, order=None,
                   _live=None, x_prep=None, kernel=0):
    """1/(1+fastdtw) for all (x row, y row) pairs -> (n_x, n_y) float32; empty x rows -> PAD.
    ``dedupe``: identical x rows (sorted degree sequences of small components repeat a lot: 50k BFS
    components of the benchmark have 2.7k distinct internal sequences) are computed once and the
    result rows gathered back.  ``order_rows``: process the x rows sorted by (length, coarse series) so
    that the lanes of a wavefront work on similar series.  Neither changes any value.
    ``x_prep``: a dict the caller keeps for THESE x rows (the degree sequences of a split's components are the same
    every pass): the grouping of repeated rows and the processing order are computed on the first call and reused;
    the series the kernel reads are always this call's ``x_val``.
    ``kernel``: 0 = pick by size, 1 = the general (workspace-resident) kernel; same values.
    ``tie_order``: fastdtw's predecessor rule (0 / 1 / 2); None = config.DTW_TIE_ORDER, the product's default."""
    if tie_order is None:
        from .config import DTW_TIE_ORDER as tie_order
    tie_order = int(tie_order)
    if dedupe and x_ptr.numel() - 1 > 1024 and max_x <= 64:
        kept = x_prep.get('dedupe') if x_prep is not None else None
        if kept is not None and kept[1].numel() == x_ptr.numel() - 1:
            # the grouping (which row stands for which, where the kept entries go) is reused; the VALUES the kernel reads
            # are this call's: scattered again from x_val
            uptr, rep, dst, live = kept
            # where every ENTRY of x_val goes (kept rows: their slot; entries of rows that repeat an earlier row: the spare
            # slot): the padded form's map restricted to the real entries, made once -- two launches per pass (zeros, scatter)
            # instead of the eight of padding the rows again
            dst_e = x_prep.get('dedupe_entry_dst')
            if dst_e is None or dst_e.numel() != x_val.numel():
                nrow = x_ptr.numel() - 1
                j = torch.arange(max_x, device=x_ptr.device).view(1, -1)
                real = j < (x_ptr[1:] - x_ptr[:-1]).view(-1, 1)
                dst_e = dst.view(nrow, max_x)[real].contiguous()                 # row-major = the order of x_val
                if dst_e.numel() < x_val.numel():                               # (x_val may carry an arena tail behind the last row)
                    dst_e = torch.cat([dst_e, dst_e.new_full((x_val.numel() - dst_e.numel(),), x_val.numel())])
                x_prep['dedupe_entry_dst'] = dst_e
            uval = torch.zeros(x_val.numel() + 1, dtype=torch.int32, device=x_ptr.device)
            uval.scatter_(0, dst_e, x_val)
            out_u = dtw_similarity(uptr, uval, max_x, y_ptr, y_val, max_y, tie_order, order_rows, dedupe=False, order=order,
                                   _live=live, x_prep=x_prep.setdefault('grouped', {}), kernel=kernel)
            return out_u.index_select(0, rep)
        # no host round trip: every row keeps its slot, the rows that repeat an earlier one are given
        # length 0 (their pairs exit at once -- sorted by length they fill whole wavefronts) and read
        # their representative's result row afterwards
        n = x_ptr.numel() - 1
        rows = Ragged(x_ptr, x_val, max_len=max_x).to_padded(width=max_x, fill=-1, dtype=torch.int32)
        rep = _row_representatives(rows)
        pos = torch.arange(n, device=x_ptr.device)
        lens = torch.where(rep == pos, x_ptr[1:] - x_ptr[:-1], torch.zeros_like(pos))
        uptr = torch.zeros(n + 1, dtype=torch.int64, device=x_ptr.device)
        torch.cumsum(lens, 0, out=uptr[1:])
        j = torch.arange(max_x, device=x_ptr.device).view(1, -1)
        dst = torch.where(j < lens.view(-1, 1), uptr[:-1].view(-1, 1) + j, x_val.numel())   # dropped entries -> spare slot
        uval = torch.zeros(x_val.numel() + 1, dtype=torch.int32, device=x_ptr.device)
        uval.scatter_(0, dst.reshape(-1), rows.reshape(-1))
        live = None
        if order is None and order_rows:
            # the length-first processing order puts the emptied rows in front: hand the kernel the live
            # range (a device-side pair, no round trip) so that it deals its lanes over the live rows only
            n_live = (lens > 0).sum()
            live = torch.stack((n - n_live, n_live))
        if x_prep is not None:
            x_prep['dedupe'] = (uptr, rep, dst.reshape(-1), live)
            # (the per-entry map of the kept path, made here as well: a mask index is a host round trip, and a caller that records
            # its passes into a hipGraph has ONE eager pass left once the grouping is decided -- hotpath._settle_dtw_grouping)
            real = j < (x_ptr[1:] - x_ptr[:-1]).view(-1, 1)
            dst_e = dst[real].contiguous()
            if dst_e.numel() < x_val.numel():
                dst_e = torch.cat([dst_e, dst_e.new_full((x_val.numel() - dst_e.numel(),), x_val.numel())])
            x_prep['dedupe_entry_dst'] = dst_e
        out_u = dtw_similarity(uptr, uval, max_x, y_ptr, y_val, max_y, tie_order, order_rows, dedupe=False, order=order,
                               _live=live, x_prep=x_prep.setdefault('grouped', {}) if x_prep is not None else None, kernel=kernel)
        return out_u.index_select(0, rep)
    lib = _lib.load()
    for t, nm in ((x_ptr, 'x_ptr'), (y_ptr, 'y_ptr')):
        _req(t, torch.int64, nm)
    for t, nm in ((x_val, 'x_val'), (y_val, 'y_val')):
        _req(t, torch.int32, nm)
    nx, ny = x_ptr.numel() - 1, y_ptr.numel() - 1
    out = (torch.empty if _live is None else torch.zeros)((nx, ny), dtype=torch.float32, device=x_ptr.device)
    if order is not None:                                   # caller's processing order (tuning)
        order = order.to(torch.int32).contiguous()
    elif order_rows and nx > 64:
        # (length, then the series sampled at its start, thirds and end): rows are sorted degree
        # sequences, four quantiles place a series' shape well enough that the lanes of a wavefront
        # sweep similar windows -- one int64 key per row (sgnn_dtw_order_keys), no host round trip.
        # Measured on the benchmark's external side (round 1 kernel): unordered 8.9 ms, (length, median, sum) 8.1 ms,
        # (length, four quantiles) 7.5 ms, full lexicographic order 7.9 ms; round 2: the coarse-series key, see the kernel.
        order = x_prep.get('order') if x_prep is not None else None
        if order is None or order.numel() != nx:
            key = torch.empty(nx, dtype=torch.int64, device=x_ptr.device)
            check(lib.sgnn_dtw_order_keys(_ptr(x_ptr), _ptr(x_val), nx, _ptr(key), _stream()), 'sgnn_dtw_order_keys')
            order = torch.argsort(key).to(torch.int32).contiguous()
            if x_prep is not None:
                x_prep['order'] = order
    wsb = lib.sgnn_dtw_workspace_bytes(nx, max_x, ny, max_y)
    ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=x_ptr.device)
    if _live is not None and order is not None:
        check(lib.sgnn_dtw_similarity_live(_ptr(x_ptr), _ptr(x_val), nx, max_x, _ptr(y_ptr), _ptr(y_val), ny, max_y,
                                           tie_order, int(kernel), _ptr(order), _ptr(_live), _ptr(out), _ptr(ws), wsb, _stream()),
              'sgnn_dtw_similarity_live')
        return out
    check(lib.sgnn_dtw_similarity(_ptr(x_ptr), _ptr(x_val), nx, max_x, _ptr(y_ptr), _ptr(y_val), ny, max_y, tie_order,
                                  int(kernel), _ptr(order), _ptr(out), _ptr(ws), wsb, _stream()), 'sgnn_dtw_similarity')
    return out


# ---------------------------------------------------------------------------------------
# float half (autograd Functions)
# ---------------------------------------------------------------------------------------

DETERMINISTIC = True       # table gradients by sorted segmented sums (bit-reproducible); False: float atomics.
#                            The default for calls made outside a model's forward; a model states its own choice for the
#                            duration of ITS forward with ``deterministic(flag)``, every op records the choice in its autograd
#                            context and the backward follows the record -- two models with different settings do not
#                            disturb each other.
_DET_SCOPE = None


class deterministic:
    """``with ops.deterministic(flag):`` -- the backward form (sorted sums / float atomics) of the ops whose FORWARD runs
    inside the block."""

    def __init__(self, flag):
        self.flag = bool(flag)

    def __enter__(self):
        global _DET_SCOPE
        self.old, _DET_SCOPE = _DET_SCOPE, self.flag

    def __exit__(self, *exc):
        global _DET_SCOPE
        _DET_SCOPE = self.old


def _det_now():
    return DETERMINISTIC if _DET_SCOPE is None else _DET_SCOPE


def sort_edges_by_key(keys, max_key):
    """Stable sort of edge numbers by int32 target key in [0, max_key] (sgnn_sort_edges_by_key: radix sort over the
    bits max_key has) -> (key_sorted, order), int32 each.  What scatter_add_rows consumes; an order that does not
    change between passes can be computed once and handed to it as ``presorted``."""
    lib = _lib.load()
    _req(keys, torch.int32, 'keys')
    E = keys.numel()
    sk, order = torch.empty_like(keys), torch.empty_like(keys)
    if E:
        wsb = lib.sgnn_sort_edges_by_key_workspace_bytes(E, int(max_key))
        if wsb < 0:
            raise RuntimeError('sgnn_sort_edges_by_key_workspace_bytes failed')
        ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=keys.device)
        check(lib.sgnn_sort_edges_by_key(_ptr(keys), E, int(max_key), _ptr(sk), _ptr(order), _ptr(ws), wsb, _stream()),
              'sgnn_sort_edges_by_key')
    return sk, order


SCATTER_TOGETHER_BELOW = 1 << 16       # edges: shorter lists into a tapped table wait for the step's other lists (one sort, one scatter)


def scatter_add_rows(table, keys, G=None, edge_row=None, edges_per_row=1, c1=None, c2=None, v=None, arg=None,
                     presorted=None, together=None):
    """table[keys[e], :] += c1[e] * G[row(e), :] + c2[e] * v   without atomics (sgnn_scatter_add_rows_sorted):
    the edges are sorted stably by target row and every row is summed by one owner in that order, so the
    result is bit-reproducible.  keys int32 (E), 0 = no contribution; row(e) = edge_row[e] or e // edges_per_row.
    presorted: sort_edges_by_key(keys, ...) of these very keys, when the caller keeps it.
    together: the _GradAcc whose buffer ``table`` is -- a short list without a kept order then only joins the accumulator's
    pending lists, which are sorted and scattered as ONE list when the table's gradient is handed over (_GradAcc.flush)."""
    lib = _lib.load()
    E = keys.numel()
    if E == 0:
        return table
    _req(keys, torch.int32, 'keys')
    _req(table, torch.float32, 'table')
    for t, nm in ((G, 'G'), (c1, 'c1'), (c2, 'c2'), (v, 'v')):
        _req(t, torch.float32, nm)
    _req(edge_row, torch.int32, 'edge_row')
    _req(arg, torch.int32, 'arg')
    D = table.shape[1]
    if together is not None and presorted is None and arg is None and E < SCATTER_TOGETHER_BELOW and D <= 256 \
            and (G is None or (G.dim() == 2 and G.shape[1] == D and G.is_contiguous())):
        together.jobs.append((keys.reshape(-1), G, edge_row, int(edges_per_row), c1, c2, v))
        return table
    sk, order = presorted if presorted is not None else sort_edges_by_key(keys.reshape(-1), table.shape[0] - 1)
    if sk.numel() != E or order.numel() != E:
        raise ValueError('presorted order does not belong to these keys')
    wsb = lib.sgnn_scatter_add_rows_workspace_bytes(E, D)
    ws = torch.empty(wsb // 4 + 1, dtype=torch.int32, device=table.device)
    check(lib.sgnn_scatter_add_rows_sorted(_ptr(order), _ptr(sk), E, _ptr(edge_row), int(edges_per_row), _ptr(G), D, _ptr(c1),
                                           _ptr(c2), _ptr(v), _ptr(arg), _ptr(table), _ptr(ws), wsb, _stream()),
          'sgnn_scatter_add_rows_sorted')
    return table


class _GradAcc:
    """One dense gradient buffer for an embedding table, shared by every op that reads the table in
    a forward pass.  Without it each consumer returns its own zero-filled (N+1, D) gradient (256 MB
    at N = 1M, D = 64) and autograd adds them pairwise: ~7 fills + 6 adds of the whole table per
    step.  With it the backward kernels atomically add into the same buffer and the table's
    gradient is produced once."""

    def __init__(self, owner=None):
        self.buf = None
        self.owner = owner         # the parameter whose gradient this is (it may hold a buffer an optimizer zeroed)
        self.jobs = []             # short edge lists waiting to be scattered together (scatter_add_rows(together=))
        self.producers = []        # layer bodies whose edge lists (keys, weights) are still to be written: (args, g_z, keys, c1, c2, keep)

    def buffer(self, shape, device):
        if self.buf is None:
            self.buf = take_zeroed(self.owner, shape, device)
        return self.buf

    def flush(self):
        """Scatter the pending lists into the buffer as one sorted list (sgnn_scatter_add_rows_multi)."""
        producers, self.producers = self.producers, []
        lib = _lib.load()
        cap = int(lib.sgnn_mpn_fwd_many_max_bodies()) if producers else 1
        for lo in range(0, len(producers), cap):                 # the waiting bodies' edge lists, eight bodies per launch
            group = producers[lo:lo + cap]
            arr = (MpnArgs * len(group))(*[g[0] for g in group])
            pz, pk, p1, p2 = (_ptr_table([g[i] for g in group]) for i in (1, 2, 3, 4))
            check(lib.sgnn_mpn_bwd_edges_many(len(group), ctypes.cast(arr, ctypes.c_void_p), pz.ctypes.data, pk.ctypes.data,
                                              p1.ctypes.data, p2.ctypes.data, _stream()), 'sgnn_mpn_bwd_edges_many')
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        if self.owner is None:
            raise RuntimeError('pending table-gradient lists without a table')
        table = self.buffer(tuple(self.owner.shape), jobs[0][0].device)
        if len(jobs) == 1:                               # nothing to share a sort with
            keys, G, edge_row, epr, c1, c2, v = jobs[0]
            scatter_add_rows(table, keys, G=G, edge_row=edge_row, edges_per_row=epr, c1=c1, c2=c2, v=v)
            return
        n = len(jobs)
        D = table.shape[1]

        def ptrs(k):
            return np.array([0 if j[k] is None else j[k].data_ptr() for j in jobs], dtype=np.uint64)
        keys_p, G_p, row_p, c1_p, c2_p, v_p = ptrs(0), ptrs(1), ptrs(2), ptrs(4), ptrs(5), ptrs(6)
        n_edges = np.array([j[0].numel() for j in jobs], dtype=np.int64)
        epr = np.array([j[3] for j in jobs], dtype=np.int64)
        total, max_key = int(n_edges.sum()), table.shape[0] - 1
        wsb = lib.sgnn_scatter_add_rows_multi_workspace_bytes(total, D, max_key)
        ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=table.device)
        check(lib.sgnn_scatter_add_rows_multi(n, keys_p.ctypes.data, n_edges.ctypes.data, row_p.ctypes.data, epr.ctypes.data,
                                              G_p.ctypes.data, c1_p.ctypes.data, c2_p.ctypes.data, v_p.ctypes.data, D, max_key,
                                              _ptr(table), _ptr(ws), wsb, _stream()), 'sgnn_scatter_add_rows_multi')


def take_zeroed(owner, shape, device):
    """A zero-filled float32 buffer for ``owner``'s gradient: the one an optimizer handed back already zeroed
    (release_zeroed -- it hangs on the parameter, so it lives and dies with it and no other model can pick it up),
    else a fresh fill."""
    buf = owner.__dict__.pop('_sgnn_zeroed', None) if owner is not None else None
    if buf is not None and tuple(buf.shape) == tuple(shape) and buf.device == torch.device(device):
        return buf
    return torch.zeros(shape, dtype=torch.float32, device=device)


def release_zeroed(owner, buf):
    """``buf`` (the gradient of parameter ``owner``) is all zeros and the optimizer is done with it: the next
    accumulation of that parameter's gradient takes it as is."""
    owner._sgnn_zeroed = buf


def drop_zeroed(owner):
    owner.__dict__.pop('_sgnn_zeroed', None)


class _TableTap(torch.autograd.Function):
    """Identity on the table.  Consumers that find the accumulator on the tapped tensor add their
    gradient into its buffer and return None; autograd still runs this node's backward only after
    all of them (dependencies are counted per edge, not per defined gradient), and it hands the
    buffer to the table's AccumulateGrad."""

    @staticmethod
    def forward(ctx, E, acc):
        ctx.acc = acc
        ctx.set_materialize_grads(False)
        return E.view_as(E)

    @staticmethod
    def backward(ctx, g):
        ctx.acc.flush()                                  # the short lists that waited for each other: one sort, one scatter
        buf, ctx.acc.buf = ctx.acc.buf, None
        if buf is None:
            return g, None
        return (buf if g is None else buf.add_(g)), None


def tap_table(E, half=None):
    """Route the gradients of every fused consumer of ``E`` in this forward pass into one buffer.
    ``half``: an IEEE-half copy of the table (same shape): the fused ops then READ the half copy
    (half the gather bytes, fp32 accumulation) while gradients still flow to the fp32 ``E``."""
    if torch.is_grad_enabled() and E.requires_grad:
        acc = _GradAcc(E)
        t = _TableTap.apply(E, acc)
        t._sgnn_acc = acc
    elif half is not None:
        t = E.detach()                       # a fresh tensor object to carry the attribute
    else:
        return E
    if half is not None:
        _req(half, torch.float16, 'half table')
        t._sgnn_half = half
    return t


class _CCEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, ptr, nodes, aggregator, padded_len, stride, presorted):
        ctx.acc = getattr(E, '_sgnn_acc', None)
        ctx.presorted = presorted
        half = getattr(E, '_sgnn_half', None)
        lib = _lib.load()
        _req(E, torch.float32, 'E')
        n = ptr.numel() - 1
        D = E.shape[1]
        out = torch.empty((n, D), dtype=torch.float32, device=E.device)
        arg = torch.empty((n, D), dtype=torch.int32, device=E.device) if aggregator == 1 else None
        if half is not None:
            check(lib.sgnn_cc_embed_fwd_f16(_ptr(half), D, _ptr(ptr), _ptr(nodes), n, aggregator, padded_len, _ptr(out),
                                            _ptr(arg), _stream()), 'sgnn_cc_embed_fwd_f16')
        else:
            check(lib.sgnn_cc_embed_fwd(_ptr(E), D, _ptr(ptr), _ptr(nodes), n, aggregator, padded_len, _ptr(out),
                                        _ptr(arg), _stream()), 'sgnn_cc_embed_fwd')
        ctx.save_for_backward(ptr, nodes, arg)
        ctx.aggregator, ctx.shape = aggregator, E.shape
        ctx.stride = stride
        ctx.det = _det_now()
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None, None
        lib = _lib.load()
        ptr, nodes, arg = ctx.saved_tensors
        g = g.contiguous()
        gE = ctx.acc.buffer(ctx.shape, g.device) if ctx.acc is not None else \
            torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        if ctx.det and ctx.shape[1] <= 256:
            # every (member, component) pair is an edge member -> component row; sorted by member, one owner per
            # table row (sgnn_scatter_add_rows_sorted).  Max aggregator: a column counts where the member is its argmax.
            n = ptr.numel() - 1
            E = nodes.numel()
            am = arg if ctx.aggregator == 1 else None
            if ctx.stride > 0:                         # fixed-stride sets: row = entry // stride
                scatter_add_rows(gE, nodes, G=g, edges_per_row=ctx.stride, arg=am, presorted=ctx.presorted, together=ctx.acc)
            else:                                      # ragged (the node array may be an arena longer than ptr[-1])
                pos = torch.arange(E, device=g.device)
                rows = (torch.searchsorted(ptr, pos, right=True) - 1).clamp_(min=0, max=max(n - 1, 0)).to(torch.int32)
                keys = torch.where(pos < ptr[-1], nodes, torch.zeros_like(nodes))
                scatter_add_rows(gE, keys.contiguous(), G=g, edge_row=rows.contiguous(), arg=am, together=ctx.acc)
        else:
            check(lib.sgnn_cc_embed_bwd(_ptr(g), ctx.shape[1], _ptr(ptr), _ptr(nodes), ptr.numel() - 1, ctx.aggregator,
                                        _ptr(arg), _ptr(gE), _stream()), 'sgnn_cc_embed_bwd')
        return (None if ctx.acc is not None else gE), None, None, None, None, None, None


def cc_embed(E, sets, aggregator='sum', padded_len=0, stride=0, presorted=None):
    """initialize_cc_embeddings on ragged components -> (n_sets, D).  ``stride`` > 0: the caller promises
    fixed-stride sets (ptr[i] = i * stride, PAD entries included) -- spares the backward a search.
    ``presorted`` (with stride): sort_edges_by_key(sets.nodes, ...) kept by the caller -- the members of a split's
    components do not change between passes, so the backward's sort need not be repeated."""
    return _CCEmbed.apply(E, sets.ptr, sets.nodes, 0 if aggregator == 'sum' else 1, int(padded_len), int(stride),
                          presorted if stride > 0 else None)


SRC_DENSE, SRC_GATHER, SRC_SHARED = 0, 1, 2


def _mpn_args(src, x, ids, id_div, edge_mask, row_mask, sims, sim_col, sims_per_edge, wp, bp, R, A, D):
    a = MpnArgs()
    a.src, a.sims_per_edge = src, 1 if sims_per_edge else 0
    a.R, a.A, a.D = R, A, D
    half = getattr(x, '_sgnn_half', None) if src == SRC_GATHER else None     # fp16-stored twin of the table
    a.x, a.ids, a.id_div = _ptr(half if half is not None else x), _ptr(ids), id_div
    a.x_f16 = 1 if half is not None else 0
    a.edge_mask, a.row_mask = _ptr(edge_mask), _ptr(row_mask)
    a.sims, a.sims_ld, a.sim_col = _ptr(sims), sims.shape[-1], _ptr(sim_col)
    a.wp, a.bp = _ptr(wp), _ptr(bp)
    return a


# Forward launches of layer bodies that wait for each other (ops.mpn(lazy=True)): the bodies of one message-passing layer read the
# layer below only, so their kernels go out as ONE launch when the first consumer needs a result -- ``flush_lazy_mpn`` (called by
# ``update_layers`` and by the layer loop of SubGNN.forward after every layer).  Entries keep their tensors alive.
# The queue belongs to the THREAD that runs the forward (another thread's forward can neither flush nor drop it) and every
# entry remembers the stream it was queued under: launching it under another stream would order it against the wrong work, so
# that raises instead (a forward runs under one stream).
class _LazyQueue(threading.local):
    def __init__(self):
        self.entries = []


_LAZY = _LazyQueue()


def lazy_mpn_pending():
    """Number of layer bodies queued by this thread and not launched yet."""
    return len(_LAZY.entries)


def flush_lazy_mpn():
    """Launch the queued layer bodies (sgnn_mpn_fwd_many, up to 8 per launch)."""
    queue, _LAZY.entries = _LAZY.entries, []
    if not queue:
        return
    now = _stream().value
    if any(q[4] != now for q in queue):
        raise RuntimeError('ops.flush_lazy_mpn: layer bodies were queued under another stream than the one that launches them')
    lib = _lib.load()
    cap = int(lib.sgnn_mpn_fwd_many_max_bodies())
    for lo in range(0, len(queue), cap):
        group = queue[lo:lo + cap]
        if len(group) == 1:
            a, agg, z = group[0][:3]
            check(lib.sgnn_mpn_fwd(ctypes.byref(a), _ptr(agg), _ptr(z), _stream()), 'sgnn_mpn_fwd')
            continue
        arr = (MpnArgs * len(group))(*[g[0] for g in group])
        pa, pz = _ptr_table([g[1] for g in group]), _ptr_table([g[2] for g in group])
        check(lib.sgnn_mpn_fwd_many(len(group), ctypes.cast(arr, ctypes.c_void_p), pa.ctypes.data, pz.ctypes.data, _stream()),
              'sgnn_mpn_fwd_many')


def drop_lazy_mpn():
    """Forget this thread's queued launches (a forward that did not finish)."""
    _LAZY.entries = []


class _MPN(torch.autograd.Function):
    """agg (R,D), z (R,A) = gather-weight-aggregate + read-out; grads for x, wp (bp via z)."""

    @staticmethod
    def forward(ctx, x, wp, bp, sims, ids, edge_mask, row_mask, sim_col, src, id_div, sims_per_edge, R, A, edge_plan=None,
                keep_chunks=False, relu_z=False, lazy=False):
        lib = _lib.load()
        ctx.edge_plan = edge_plan if src == SRC_GATHER else None
        ctx.relu_z = bool(relu_z)
        _req(x, torch.float32, 'x')
        _req(wp, torch.float32, 'wp')
        _req(bp, torch.float32, 'bp')
        _req(sims, torch.float32, 'sims')
        _req(ids, torch.int64, 'ids')
        _req(edge_mask, torch.uint8, 'edge_mask')
        _req(row_mask, torch.uint8, 'row_mask')
        _req(sim_col, torch.int64, 'sim_col')
        D = x.shape[-1]
        z = torch.empty((R, A), dtype=torch.float32, device=x.device)
        if A == 0:
            agg = torch.zeros((R, D), dtype=torch.float32, device=x.device)
        else:
            a = _mpn_args(src, x, ids, id_div, edge_mask, row_mask, sims, sim_col, sims_per_edge, wp, bp, R, A, D)
            if relu_z:
                a.flags = 2                                         # SGNN_MPN_RELU_Z: the read-out leaves the kernel activated
            chunks = lib.sgnn_mpn_fwd_chunks(ctypes.byref(a))       # batch-sized calls split a row's anchors
            agg = torch.empty((chunks, R, D), dtype=torch.float32, device=x.device)
            if lazy and keep_chunks and R > 0:
                # the launch waits for the other bodies of its layer (flush_lazy_mpn); nothing reads agg / z before that
                _LAZY.entries.append((a, agg, z, (x, wp, bp, sims, ids, edge_mask, row_mask, sim_col, getattr(x, '_sgnn_half', None)),
                                      _stream().value))
            else:
                check(lib.sgnn_mpn_fwd(ctypes.byref(a), _ptr(agg), _ptr(z), _stream()), 'sgnn_mpn_fwd')
            if not keep_chunks:
                agg = agg[0] if chunks == 1 else agg.sum(0)          # a fixed order: no atomics
            # (keep_chunks: the (chunks, R, D) partials go to update_layer, which adds them while it loads them)
        ctx.save_for_backward(x, wp, bp, sims, ids, edge_mask, row_mask, sim_col, *((z,) if relu_z else ()))
        ctx.set_materialize_grads(False)              # an unused output (the N channel never reads z) arrives as None
        ctx.meta = (src, id_div, sims_per_edge, R, A, D)
        ctx.acc = getattr(x, '_sgnn_acc', None) if src == SRC_GATHER else None     # x is the tapped table
        ctx.det = _det_now()
        ctx.half = getattr(x, '_sgnn_half', None) if src == SRC_GATHER else None
        return agg, z

    @staticmethod
    def backward(ctx, g_agg, g_z):
        lib = _lib.load()
        x, wp, bp, sims, ids, edge_mask, row_mask, sim_col = ctx.saved_tensors[:8]
        src, id_div, sims_per_edge, R, A, D = ctx.meta
        need_x, need_wp, need_bp = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if g_z is None:
            # nobody read the read-out (the neighbourhood channel's bodies): its weight and bias get NO gradient -- None, as autograd
            # gives an unused parameter, not two zero fills per body that the optimizer then carries through its norm and update
            need_wp = need_bp = False
        # The gradient of the read-out goes through the fused relu.  Where every reader of grad_z is one of the deterministic
        # kernels below, they apply the gate themselves (args.z_act) and produce the read-out bias's gradient too -- a threshold
        # launch and a reduction launch less per layer body; anywhere else the gated gradient is materialised first.
        z_act = ctx.saved_tensors[8] if ctx.relu_z else None
        gather_det = A > 0 and src == SRC_GATHER and ctx.det and D <= 256
        shared_det = A > 0 and src == SRC_SHARED and ctx.det
        planned = gather_det and ctx.edge_plan is not None and ctx.edge_plan['keys'].numel() == R * A
        in_kernel = z_act is not None and g_z is not None and (need_x or need_wp) and ((gather_det and not planned) or shared_det)
        if z_act is not None and g_z is not None and not in_kernel:
            g_z = torch.ops.aten.threshold_backward(g_z.contiguous(), z_act, 0.0)
        z_gate = z_act if in_kernel else None
        if g_agg is not None and g_agg.dim() == 3:
            # the forward handed out its anchor-chunk partials; their consumer adds them, so every chunk receives the same gradient
            if g_agg.shape[0] > 1 and g_agg.stride(0) != 0:
                raise RuntimeError('mpn: the chunk partials of the aggregate were consumed by something other than a sum over chunks')
            g_agg = g_agg[0]
        g_agg = g_agg.contiguous() if g_agg is not None else None
        g_z = g_z.contiguous() if g_z is not None else None
        gx = gwp = gbp = None
        # (the deterministic SHARED backward assigns every element of both gradients, the deterministic GATHER one replaces
        # grad_wp by a column sum: no zero fills for those -- a 4-layer batch-sized step made ~40 of them)
        assigned = A > 0 and src == SRC_SHARED and ctx.det
        if need_x:
            if ctx.acc is not None:
                gx = ctx.acc.buffer(x.shape, x.device)          # the kernel adds into the shared buffer
            else:
                gx = torch.empty_like(x) if (src == SRC_DENSE or assigned) else torch.zeros_like(x)
        if need_wp:
            replaced = A > 0 and src == SRC_GATHER and ctx.det and D <= 256 and g_z is not None
            gwp = None if replaced else (torch.empty if assigned else torch.zeros)(D, dtype=torch.float32, device=x.device)
        if (need_x or need_wp) and A > 0 and src == SRC_GATHER and ctx.det and D <= 256:
            # table gradient by a sorted segmented sum, read-out weight gradient by per-row partials: no atomics
            if ctx.half is not None:
                x._sgnn_half = ctx.half
            a = _mpn_args(src, x, ids, id_div, edge_mask, row_mask, sims, sim_col, sims_per_edge, wp, bp, R, A, D)
            a.z_act = _ptr(z_gate)
            plan = ctx.edge_plan
            if need_x and (g_agg is not None or g_z is not None) and plan is not None and plan['keys'].numel() == R * A:
                # the static half of the edge list (target rows, weights, sorted order) came with the prepared pass
                c2 = (plan['c1'] * g_z.reshape(-1)) if g_z is not None else None
                scatter_add_rows(gx, plan['keys'], G=g_agg, edges_per_row=A, c1=plan['c1'], c2=c2, v=wp if c2 is not None else None,
                                 presorted=plan['sorted'], together=ctx.acc)
            elif need_x and (g_agg is not None or g_z is not None):
                keys = torch.empty(R * A, dtype=torch.int32, device=x.device)
                c1 = torch.empty(R * A, dtype=torch.float32, device=x.device)
                c2 = torch.empty(R * A, dtype=torch.float32, device=x.device) if g_z is not None else None
                waits = ctx.acc is not None and 0 < R * A < SCATTER_TOGETHER_BELOW and (g_agg is None or g_agg.shape[1] == D)
                if waits:
                    # the list joins the step's combined scatter (below): nothing reads it before the table's gradient is handed
                    # over, so its launch waits for the other bodies' too (_GradAcc.flush -> sgnn_mpn_bwd_edges_many)
                    a2 = _mpn_args(src, x, ids, id_div, edge_mask, row_mask, sims, sim_col, sims_per_edge, wp, bp, R, A, D)
                    a2.z_act = _ptr(z_gate)
                    ctx.acc.producers.append((a2, g_z, keys, c1, c2, (x, ids, row_mask, sims, sim_col, z_gate, ctx.half)))
                else:
                    check(lib.sgnn_mpn_bwd_edges(ctypes.byref(a), _ptr(g_z), _ptr(keys), _ptr(c1), _ptr(c2), _stream()),
                          'sgnn_mpn_bwd_edges')
                n_jobs = len(ctx.acc.jobs) if ctx.acc is not None else 0
                scatter_add_rows(gx, keys, G=g_agg, edges_per_row=A, c1=c1, c2=c2, v=wp if c2 is not None else None, together=ctx.acc)
                if waits and len(ctx.acc.jobs) == n_jobs:             # (the scatter did not wait after all: its list is needed now)
                    raise RuntimeError('mpn backward: an edge list whose launch waits was scattered at once')
            if need_wp and g_z is not None:
                ld = D + 1 if need_bp else D            # (column D: the row's share of the read-out bias's gradient)
                partial = torch.empty((R, ld), dtype=torch.float32, device=x.device)
                check(lib.sgnn_mpn_bwd_wp_partial(ctypes.byref(a), _ptr(g_z), _ptr(partial), ld, _stream()),
                      'sgnn_mpn_bwd_wp_partial')
                col = column_sum(partial)
                gwp = col[:D]
                if need_bp:
                    gbp = col[D:].view_as(bp)
        elif (need_x or need_wp) and A > 0 and src == SRC_SHARED and ctx.det:
            # row-tile partials added in tile order (sgnn_mpn_bwd_shared_det): no atomics
            a = _mpn_args(src, x, ids, id_div, edge_mask, row_mask, sims, sim_col, sims_per_edge, wp, bp, R, A, D)
            a.z_act = _ptr(z_gate)
            wsb = lib.sgnn_mpn_bwd_shared_det_workspace_bytes(R, A, D)
            ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=x.device)
            if need_bp and g_z is not None:
                gbp = torch.empty_like(bp)
            check(lib.sgnn_mpn_bwd_shared_det(ctypes.byref(a), _ptr(g_agg), _ptr(g_z), _ptr(gx), _ptr(gwp), _ptr(gbp), _ptr(ws), wsb,
                                              _stream()), 'sgnn_mpn_bwd_shared_det')
        elif (need_x or need_wp) and A > 0:
            if ctx.half is not None:
                x._sgnn_half = ctx.half               # saved tensors come back as new objects
            a = _mpn_args(src, x, ids, id_div, edge_mask, row_mask, sims, sim_col, sims_per_edge, wp, bp, R, A, D)
            partial = None
            if need_wp and src == SRC_DENSE and ctx.det:
                a.flags = 1                           # SGNN_MPN_WP_PARTIAL: per-row partials, summed below in a fixed order
                partial = torch.empty((R, D), dtype=torch.float32, device=x.device)
            check(lib.sgnn_mpn_bwd(ctypes.byref(a), _ptr(g_agg), _ptr(g_z), _ptr(gx), _ptr(partial if partial is not None else gwp),
                                   _stream()), 'sgnn_mpn_bwd')
            if partial is not None:
                gwp = column_sum(partial)
        elif need_x and src == SRC_DENSE:
            gx.zero_()
        if need_wp:
            gwp = gwp.view_as(wp)
        if need_bp and gbp is None:
            if g_z is not None and z_gate is not None:          # gated inside the kernels only: the sum needs the gated values
                g_z = torch.ops.aten.threshold_backward(g_z, z_gate, 0.0)
            gbp = g_z.sum().view_as(bp) if g_z is not None else torch.zeros_like(bp)
        return (None if ctx.acc is not None else gx), gwp, gbp, None, None, None, None, None, None, None, None, None, None, None, None, None, None


def column_sum(t, chunk=512):
    """t.sum(0) for a tall (R, A) matrix.  torch's reduction over the leading dimension of a tall,
    narrow matrix runs on one workgroup per few output columns (50k x 183 -> 192 threads' worth of
    grid, 0.5 ms); summing row blocks first -- (R/chunk, chunk, A) over the middle dimension -- gives
    it R/chunk x A outputs to spread over the chip, and the second, small sum finishes."""
    R = t.shape[0]
    if t.dim() == 2 and t.is_cuda and t.dtype == torch.float32 and R >= 1024 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] \
            and not t.requires_grad:
        # sgnn_column_sum: row-block partials + one wavefront per column, two launches, fixed order
        lib = _lib.load()
        A = t.shape[1]
        out = torch.empty(A, dtype=torch.float32, device=t.device)
        wsb = lib.sgnn_column_sum_workspace_bytes(R, A)
        ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=t.device)
        check(lib.sgnn_column_sum(_ptr(t), t.stride(0), R, A, _ptr(out), _ptr(ws), wsb, _stream()), 'sgnn_column_sum')
        return out
    nb = R // chunk
    if t.dim() != 2 or nb < 8:
        return t.sum(0)
    head = t[:nb * chunk].view(nb, chunk, t.shape[1]).sum(1).sum(0)
    return head + t[nb * chunk:].sum(0) if nb * chunk < R else head


@functools.lru_cache(maxsize=None)
def _block_rows(R, want=1024):
    """Rows per block for a contraction split over row blocks: a divisor of R between want / 2 and 2 want when there is one
    (no remainder block: one GEMM and one add less), else ``want``."""
    for d in sorted(range(want // 2, 2 * want + 1), key=lambda d: abs(d - want)):
        if R % d == 0:
            return d
    return want


def contract_rows(a, b, rows_per_block=None):
    """a^T b for tall a (R, m), b (R, n): an (m x n) result contracted over R >> m, n rows.  The library runs it on
    (m / 64) x (n / 32) workgroups (16 of them for the LSTM's weight gradients: 40 us for 1.1 GFLOP); split over row
    blocks -- one batched GEMM, then a sum in block order -- it fills the chip."""
    R = a.shape[0]
    if R < 2048 or not a.is_cuda:
        return a.t() @ b
    rows_per_block = rows_per_block or _block_rows(R, 1024 if R >= 8192 else 512)
    nb = R // rows_per_block
    k = nb * rows_per_block
    head = torch.bmm(a[:k].reshape(nb, rows_per_block, -1).transpose(1, 2), b[:k].reshape(nb, rows_per_block, -1)).sum(0)
    return head + a[k:].t() @ b[k:] if k < R else head


def contract_rows_batched(a, b, want=512):
    """a[g]^T b[g] for g in range(G): a (G, R, m), b (G, R, n) -> (G, m, n), one ``bmm`` over (group, row block) pairs and a sum
    over the blocks in block order (a fixed order: reproducible)."""
    G, R = a.shape[0], a.shape[1]
    if R < 2 * want or not a.is_cuda:
        return torch.bmm(a.transpose(1, 2), b)
    rows = _block_rows(R, want)
    nb = R // rows
    k = nb * rows
    part = torch.bmm(a[:, :k].reshape(G * nb, rows, -1).transpose(1, 2), b[:, :k].reshape(G * nb, rows, -1))
    out = part.view(G, nb, part.shape[1], part.shape[2]).sum(1)
    if k < R:
        out = out + torch.bmm(a[:, k:].transpose(1, 2), b[:, k:])
    return out


class _ReadoutShared(torch.autograd.Function):
    """z = W * s + b for edge weights W (R, A) (no gradient), per-anchor scores s (A) and a scalar b:
    the read-out of the shared-anchor layer body.  Its backward is two reductions over the rows,
    done with column_sum."""

    @staticmethod
    def forward(ctx, W, s, b):
        ctx.save_for_backward(W)
        ctx.b_shape = b.shape
        return torch.addcmul(b.view(1, 1), W, s.view(1, -1))

    @staticmethod
    def backward(ctx, g):
        (W,) = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise RuntimeError('_ReadoutShared: edge weights carry no gradient')
        gs = column_sum(g * W) if ctx.needs_input_grad[1] else None
        gb = column_sum(g).sum().view(ctx.b_shape) if ctx.needs_input_grad[2] else None
        return None, gs, gb


class _LinearTallSkinny(torch.autograd.Function):
    """y = x W^T + b for a tall x (R >> features).  The library picks a single-pass kernel for the
    weight gradient g^T x -- an (out x in) result contracted over R = 50k rows runs on (out/32) x
    (in/64) workgroups, 4 of the 256 CUs, ~200 us.  Here the contraction is split over row blocks
    (one batched GEMM, then a small sum), which fills the chip (contract_rows)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ W if ctx.needs_input_grad[0] else None
        gW = None
        if ctx.needs_input_grad[1]:
            gW = contract_rows(g, x)
        gb = column_sum(g) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gW, gb


def linear(x, weight, bias):
    """nn.Linear on (R, in); rows >= 8192 take the split-contraction backward."""
    if x.dim() == 2 and x.shape[0] >= 8192 and x.is_cuda:
        return _LinearTallSkinny.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


UPDATE_DIMS = (32, 64, 128)


class _UpdateLayer(torch.autograd.Function):
    """out = relu([x | aggr] W^T + b) (sgnn_update_fwd / sgnn_update_bwd): the message-passing layer's update(),
    without the concatenation and with bias, relu and their backward fused around fp32 MFMA contractions."""

    @staticmethod
    def forward(ctx, x, aggr, W, b):
        lib = _lib.load()
        for t, nm in ((x, 'x'), (aggr, 'aggr'), (W, 'W'), (b, 'b')):
            _req(t, torch.float32, nm)
        R, D = x.shape
        if tuple(aggr.shape[-2:]) != (R, D) or tuple(W.shape) != (D, 2 * D):
            raise ValueError('update layer: x %s, aggr %s, W %s' % (tuple(x.shape), tuple(aggr.shape), tuple(W.shape)))
        out = torch.empty((R, D), dtype=torch.float32, device=x.device)
        ctx.chunks = aggr.shape[0] if aggr.dim() == 3 else 0
        if ctx.chunks > 1:
            # the anchor-chunk partials of ops.mpn, added while they are loaded; their sum is kept for the backward
            total = torch.empty((R, D), dtype=torch.float32, device=x.device)
            check(lib.sgnn_update_fwd_chunks(_ptr(x), _ptr(aggr), ctx.chunks, _ptr(W), _ptr(b), R, D, _ptr(out), _ptr(total), _stream()),
                  'sgnn_update_fwd_chunks')
            aggr = total
        else:
            aggr = aggr[0] if ctx.chunks else aggr
            check(lib.sgnn_update_fwd(_ptr(x), _ptr(aggr), _ptr(W), _ptr(b), R, D, _ptr(out), _stream()), 'sgnn_update_fwd')
        ctx.save_for_backward(x, aggr, W, out)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, aggr, W, out = ctx.saved_tensors
        g = g.contiguous()
        R, D = x.shape
        nx, na, nw, nb = ctx.needs_input_grad
        gx = torch.empty_like(x) if nx else None
        ga = torch.empty_like(aggr) if na else None
        gW = torch.empty_like(W) if nw else None
        gb = torch.empty(D, dtype=torch.float32, device=x.device) if (nb and ctx.has_bias) else None
        wsb = lib.sgnn_update_bwd_workspace_bytes(R, D)
        ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=x.device) if (gW is not None or gb is not None) else None
        check(lib.sgnn_update_bwd(_ptr(g), _ptr(out), _ptr(x), _ptr(aggr), _ptr(W), R, D, _ptr(gx), _ptr(ga), _ptr(gW),
                                  _ptr(gb), _ptr(ws), wsb, _stream()), 'sgnn_update_bwd')
        if ga is not None and ctx.chunks:
            ga = ga.unsqueeze(0).expand(ctx.chunks, R, D)            # d(sum over chunks): the same gradient for every chunk
        return gx, ga, gW, gb


def update_layer(x, aggr, weight, bias):
    """relu(nn.Linear(2 D, D)(cat([x, aggr], 1))) for (R, D) inputs (subgraph_mpn.py:233-241).  The fused HIP form
    for D in UPDATE_DIMS; other widths keep the library GEMM (torch) around the same arithmetic."""
    if aggr.dim() == 3 and not (x.is_cuda and x.dim() == 2 and x.shape[1] in UPDATE_DIMS and x.dtype == torch.float32
                                and (aggr.shape[0] == 1 or x.shape[0] <= update_chunks_max_rows())):
        aggr = aggr[0] if aggr.shape[0] == 1 else aggr.sum(0)       # chunk partials (ops.mpn(keep_chunks=True)) nobody adds in passing
    if x.is_cuda and x.dim() == 2 and x.shape[1] in UPDATE_DIMS and x.dtype == torch.float32:
        return _UpdateLayer.apply(x.contiguous(), aggr.contiguous(), weight, bias)
    return torch.relu(linear(torch.cat([x, aggr], dim=1), weight, bias))


class PendingUpdate:
    """An update layer that has not run yet: relu([x | aggr] W^T + b) for x (R, D), aggr (R, D) or the (chunks, R, D) partials
    of ops.mpn -- what SG_MPN hands back when the caller collects the bodies of a layer (``update_layers``)."""

    def __init__(self, x, aggr, weight, bias, shape):
        self.x, self.aggr, self.weight, self.bias, self.shape = x, aggr, weight, bias, tuple(shape)


def _ptr_table(ts):
    return np.array([0 if t is None else t.data_ptr() for t in ts], dtype=np.uint64)


class _UpdateLayerMany(torch.autograd.Function):
    """n update layers of one shape in one launch each way (sgnn_update_fwd_many / sgnn_update_bwd_many): the bodies of one
    message-passing layer of a batch-sized step.  Inputs: n, then (x, aggr, W, b) per body; outputs: n tensors (R, D)."""

    @staticmethod
    def forward(ctx, n, *ts):
        lib = _lib.load()
        xs, ags, Ws, bs = ts[0::4], ts[1::4], ts[2::4], ts[3::4]
        R, D = xs[0].shape
        for x, a, W, b in zip(xs, ags, Ws, bs):
            for t, nm in ((x, 'x'), (a, 'aggr'), (W, 'W'), (b, 'b')):
                _req(t, torch.float32, nm)
            if tuple(x.shape) != (R, D) or tuple(a.shape[-2:]) != (R, D) or tuple(W.shape) != (D, 2 * D) or b is None:
                raise ValueError('update layers of different shapes in one launch')
        dev = xs[0].device
        outs = [torch.empty((R, D), dtype=torch.float32, device=dev) for _ in range(n)]
        chunks = [a.shape[0] if a.dim() == 3 else 1 for a in ags]
        sums = [torch.empty((R, D), dtype=torch.float32, device=dev) if c > 1 else None for c in chunks]
        nch = np.array(chunks, dtype=np.int64)
        px, pa, pW, pb, po, ps = (_ptr_table(v) for v in (xs, ags, Ws, bs, outs, sums))
        check(lib.sgnn_update_fwd_many(n, px.ctypes.data, pa.ctypes.data, nch.ctypes.data, pW.ctypes.data, pb.ctypes.data, R, D,
                                       po.ctypes.data, ps.ctypes.data, _stream()), 'sgnn_update_fwd_many')
        kept = [s_ if s_ is not None else (a[0] if a.dim() == 3 else a) for a, s_ in zip(ags, sums)]
        ctx.save_for_backward(*xs, *kept, *Ws, *outs)
        ctx.n, ctx.chunks, ctx.three_d = n, chunks, [a.dim() == 3 for a in ags]
        ctx.set_materialize_grads(False)             # a body whose output nobody reads arrives as None: its backward is skipped
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.load()
        n = ctx.n
        sv = ctx.saved_tensors
        xs, ags, Ws, outs = sv[:n], sv[n:2 * n], sv[2 * n:3 * n], sv[3 * n:4 * n]
        R, D = xs[0].shape
        dev = xs[0].device
        gs = [g.contiguous() if g is not None else None for g in gs]
        need = ctx.needs_input_grad
        gx = [torch.empty_like(xs[k]) if gs[k] is not None and need[1 + 4 * k] else None for k in range(n)]
        ga = [torch.empty((R, D), dtype=torch.float32, device=dev) if gs[k] is not None and need[2 + 4 * k] else None for k in range(n)]
        gW = [torch.empty_like(Ws[k]) if gs[k] is not None else None for k in range(n)]
        gb = [torch.empty(D, dtype=torch.float32, device=dev) if gs[k] is not None else None for k in range(n)]
        per = lib.sgnn_update_bwd_workspace_bytes(R, D)
        ws = torch.empty(n * per // 4 + 1, dtype=torch.float32, device=dev)
        pg, po, px, pa, pW, pgx, pga, pgW, pgb = (_ptr_table(v) for v in (gs, outs, xs, ags, Ws, gx, ga, gW, gb))
        check(lib.sgnn_update_bwd_many(n, pg.ctypes.data, po.ctypes.data, px.ctypes.data, pa.ctypes.data, pW.ctypes.data, R, D,
                                       pgx.ctypes.data, pga.ctypes.data, pgW.ctypes.data, pgb.ctypes.data, _ptr(ws), n * per, _stream()),
              'sgnn_update_bwd_many')
        res = [None]
        for k in range(n):
            a = ga[k]
            if a is not None and ctx.three_d[k]:
                a = a.unsqueeze(0).expand(ctx.chunks[k], R, D)       # d(sum over chunks): the same gradient for every chunk
            res += [gx[k], a, gW[k] if need[3 + 4 * k] else None, gb[k] if need[4 + 4 * k] else None]
        return tuple(res)


def update_layers(pending):
    """The update layers of a list of PendingUpdate (the bodies of one message-passing layer) -> their outputs, in order: one
    launch each way for the batch-sized shape, else one after the other (``update_layer``)."""
    flush_lazy_mpn()                                 # the aggregates below come from layer bodies that may still be queued
    if not pending:
        return []
    p0 = pending[0]
    R, D = p0.x.shape
    same = all(tuple(p.x.shape) == (R, D) and p.bias is not None for p in pending)
    if len(pending) >= 2 and same and p0.x.is_cuda and p0.x.dtype == torch.float32 and D in UPDATE_DIMS and 0 < R <= update_chunks_max_rows():
        outs, cap = [], int(_lib.load().sgnn_update_many_max_bodies())
        for lo in range(0, len(pending), cap):
            group = pending[lo:lo + cap]
            flat = []
            for p in group:
                flat += [p.x.contiguous(), p.aggr.contiguous(), p.weight, p.bias]
            outs += list(_UpdateLayerMany.apply(len(group), *flat))
        return outs
    return [update_layer(p.x, p.aggr, p.weight, p.bias) for p in pending]


@functools.lru_cache(maxsize=None)
def update_chunks_max_rows():
    return int(_lib.load().sgnn_update_fwd_chunks_max_rows())


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, betas, eps, step, grad_scale=None, zero_grad=False, step_counter=None):
    """One Adam update of a large float32 parameter in one pass (sgnn_adam_step); ``grad_scale``: device scalar.
    ``step_counter``: int64 (1,) device tensor that replaces the host count ``step`` (incremented on the stream before the
    update: the form a recorded step replays)."""
    for t, nm in ((param, 'param'), (grad, 'grad'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq'), (grad_scale, 'grad_scale')):
        _req(t, torch.float32, nm)
    if step_counter is not None:
        _req(step_counter, torch.int64, 'step_counter')
        check(_lib.load().sgnn_adam_step_counted(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr),
                                                 float(betas[0]), float(betas[1]), float(eps), _ptr(step_counter), _ptr(grad_scale),
                                                 1 if zero_grad else 0, _stream()), 'sgnn_adam_step_counted')
        return
    check(_lib.load().sgnn_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr),
                                     float(betas[0]), float(betas[1]), float(eps), int(step), _ptr(grad_scale),
                                     1 if zero_grad else 0, _stream()), 'sgnn_adam_step')


class OptimTail:
    """clip_grad_norm_ + Adam over a fixed list of float32 CUDA parameters in two launches (sgnn_optim_sumsq, sgnn_optim_adam:
    train_config.py Trainer(gradient_clip_val) + SubGNN/SubGNN.py:1156-1161).  Holds the pointer tables of the parameters and
    their moments (they never move) and rebuilds the gradient pointers per step (they do, in eager mode)."""

    def __init__(self, params, exp_avg, exp_avg_sq, zero_grad, row_skip=()):
        """``row_skip``: indices of 2-D parameters whose untouched rows the update may skip (a byte per row, kept here: set once
        a row has had a non-zero gradient; a row without one has m = v = 0 and Adam does not move it)."""
        self.n = len(params)
        self.params, self.exp_avg, self.exp_avg_sq = list(params), list(exp_avg), list(exp_avg_sq)
        for t in self.params + self.exp_avg + self.exp_avg_sq:
            _req(t, torch.float32, 'parameter / moment')
        self.numels = np.array([p.numel() for p in params], dtype=np.int64)
        self.p_ptr = np.array([p.data_ptr() for p in params], dtype=np.uint64)
        self.m_ptr = np.array([t.data_ptr() for t in exp_avg], dtype=np.uint64)
        self.v_ptr = np.array([t.data_ptr() for t in exp_avg_sq], dtype=np.uint64)
        self.zero = np.array([1 if z else 0 for z in zero_grad], dtype=np.int32)
        self.row_len = np.zeros(self.n, dtype=np.int64)
        self.seen_ptr = np.zeros(self.n, dtype=np.uint64)
        self.seen = {}
        for i in row_skip:
            p = params[i]
            D = p.shape[-1] if p.dim() == 2 else 0
            if D >= 4 and D <= 256 and (D & (D - 1)) == 0 and p.data_ptr() % 16 == 0:
                self.seen[i] = torch.zeros(p.shape[0], dtype=torch.uint8, device=p.device)
                self.row_len[i], self.seen_ptr[i] = D, self.seen[i].data_ptr()

    def step(self, which, grads, lr, betas, eps, max_norm, steps=None, step_counters=None):
        """``which``: indices (ascending) of the parameters that have a gradient this step, ``grads`` theirs (float32,
        contiguous).  ``steps``: host step numbers per listed parameter -- or ``step_counters``: int64 device tensor over ALL
        parameters of the list (those of ``which`` advance).  Returns the (2,) device tensor [coefficient, total norm] when
        clipping, else None."""
        lib = _lib.load()
        k = len(which)
        if k == 0:
            return None
        # (a parameter whose storage was replaced since construction -- ``p.data = ...`` -- must not be updated at its old address)
        now = np.fromiter((p.data_ptr() for p in self.params), dtype=np.uint64, count=self.n)
        if not np.array_equal(now, self.p_ptr):
            moved = [i for i in range(self.n) if now[i] != self.p_ptr[i]]
            for i in moved:
                _req(self.params[i], torch.float32, 'parameter')
                if self.params[i].numel() != int(self.numels[i]):
                    raise ValueError('parameter %d changed its size under the optimizer' % i)
            self.p_ptr = now
        for p_i, g in zip(which, grads):
            _req(g, torch.float32, 'gradient')
            if g.numel() != int(self.numels[p_i]):
                raise ValueError('gradient of parameter %d has %d elements, the parameter %d' % (p_i, g.numel(), int(self.numels[p_i])))
        idx = np.asarray(which, dtype=np.int64)
        full = k == self.n
        numels = self.numels if full else np.ascontiguousarray(self.numels[idx])
        p_ptr, m_ptr, v_ptr = ((self.p_ptr, self.m_ptr, self.v_ptr) if full
                               else tuple(np.ascontiguousarray(a[idx]) for a in (self.p_ptr, self.m_ptr, self.v_ptr)))
        zero = self.zero if full else np.ascontiguousarray(self.zero[idx])
        row_len = self.row_len if full else np.ascontiguousarray(self.row_len[idx])
        seen_ptr = self.seen_ptr if full else np.ascontiguousarray(self.seen_ptr[idx])
        g_ptr = np.fromiter((g.data_ptr() for g in grads), dtype=np.uint64, count=k)
        dev = grads[0].device
        st = _stream()
        counters = None
        slots = None
        if step_counters is not None:
            _req(step_counters, torch.int64, 'step_counters')
            if step_counters.numel() != self.n:
                raise ValueError('step_counters holds %d counts for %d parameters' % (step_counters.numel(), self.n))
            counters = _ptr(step_counters)
            slots = None if full else idx.ctypes.data
        out = partial = None
        n_partial = 0
        if max_norm is not None and max_norm > 0:
            n_partial = int(lib.sgnn_optim_partials(numels.ctypes.data, k))
            partial = torch.empty(n_partial, dtype=torch.float32, device=dev)
            out = torch.empty(2, dtype=torch.float32, device=dev)
            check(lib.sgnn_optim_sumsq(g_ptr.ctypes.data, numels.ctypes.data, k, _ptr(partial), counters, slots, st), 'sgnn_optim_sumsq')
        elif counters is not None:
            check(lib.sgnn_optim_count(counters, slots, k, st), 'sgnn_optim_count')
        host_steps = None if counters is not None else np.asarray(steps, dtype=np.int64)
        check(lib.sgnn_optim_adam(p_ptr.ctypes.data, g_ptr.ctypes.data, m_ptr.ctypes.data, v_ptr.ctypes.data, numels.ctypes.data,
                                  zero.ctypes.data, k, float(lr), float(betas[0]), float(betas[1]), float(eps),
                                  host_steps.ctypes.data if host_steps is not None else None, counters, slots,
                                  row_len.ctypes.data if self.seen else None, seen_ptr.ctypes.data if self.seen else None,
                                  _ptr(partial), n_partial,
                                  float(max_norm) if partial is not None else 0.0, _ptr(out), st), 'sgnn_optim_adam')
        return out


def clip_coefficient(big_grads, small_grads, max_norm):
    """clip_grad_norm_'s coefficient min(1, max_norm / (total + 1e-6)) as a (1,) device tensor: the large gradients are
    reduced by sgnn_grad_sumsq (one launch each), the small ones by one multi-tensor norm."""
    lib = _lib.load()
    dev = (big_grads + small_grads)[0].device
    P = int(lib.sgnn_grad_sumsq_partials())
    partial = torch.empty(max(len(big_grads), 1) * P, dtype=torch.float32, device=dev)
    for i, g in enumerate(big_grads):
        _req(g, torch.float32, 'gradient')
        check(lib.sgnn_grad_sumsq(_ptr(g), g.numel(), ctypes.c_void_p(partial.data_ptr() + 4 * i * P), _stream()), 'sgnn_grad_sumsq')
    other = torch.stack(torch._foreach_norm(small_grads)).float() if small_grads else None
    coef = torch.empty(1, dtype=torch.float32, device=dev)
    check(lib.sgnn_clip_coefficient(_ptr(partial), len(big_grads) * P, _ptr(other), other.numel() if other is not None else 0,
                                    float(max_norm), _ptr(coef), None, _stream()), 'sgnn_clip_coefficient')
    return coef


_ZEROS = {}


def zeros_cached(shape, dtype, device):
    """A READ-ONLY all-zero tensor, kept per (shape, dtype, device): the aggregate of a layer body whose similarities are all
    zero is the same zeros every step -- no fill launch per step.  (One that does not exist yet while a stream is capturing is
    made for that recording only: a tensor created there belongs to the recording's pool.)"""
    key = (tuple(shape), dtype, str(device))
    t = _ZEROS.get(key)
    if t is None and torch.device(device).type == 'cuda' and torch.cuda.is_current_stream_capturing():
        return torch.zeros(shape, dtype=dtype, device=device)
    if t is None:
        if len(_ZEROS) > 16:
            _ZEROS.clear()
        t = _ZEROS[key] = torch.zeros(shape, dtype=dtype, device=device)
    return t


class ZeroSims:
    """Edge weights known to be all zero (the similarity of an anchor that lies inside its own
    component -- every N-internal edge, every P-internal edge of a single-component subgraph): the
    layer body then contributes agg = 0 and the read-out z = bp without launching anything."""

    def __init__(self, shape, device):
        self.shape, self.device = tuple(shape), device

    def index_select(self, dim, idx):
        assert dim == 0
        return ZeroSims((idx.numel(),) + self.shape[1:], self.device)

    def dense(self):
        return torch.zeros(self.shape, dtype=torch.float32, device=self.device)


SHARED_GEMM_MIN_ROWS = 4096


def _mpn_shared_gemm(x, wp, bp, sims2, ids, row_mask, sim_col, sims_per_edge, R, A, need_agg=True):
    """SRC_SHARED for shard-sized row counts: with one anchor matrix X (A, D) for all R rows the layer
    body IS a dense contraction -- agg = W X, read-out z = W * (X wp) + bp with W the (R, A) edge
    weights -- and so is its backward (dX = W^T g_agg + ...).  Plain GEMMs go to the library
    (rocBLAS / hipBLASLt through torch.matmul); autograd derives the backward GEMMs.  The hand-written
    kernel keeps the batch-sized calls, where launch count matters and the GEMM would be tiny."""
    if sim_col is not None:
        W = sims2.index_select(1, sim_col)
    elif sims_per_edge:
        W = sims2[:, :A]
    else:
        W = sims2.index_select(1, (ids - 1).clamp(min=0))
    edge = None
    if ids is not None:
        edge = (ids != 0).to(W.dtype).view(1, A)
    if row_mask is not None:
        rm = (row_mask != 0).to(W.dtype).view(R, 1)
        edge = rm if edge is None else edge * rm
    if edge is not None:
        W = W * edge
    agg = W @ x if need_agg else None          # (a layer whose updated embeddings nobody reads: read-out only)
    z = _ReadoutShared.apply(W, x @ wp.view(-1), bp)
    return agg, z


def mpn_edge_plan(sims, ids, row_mask, *, R, A, D, max_key, id_div=1, sim_col=None, sims_per_edge=False):
    """What the table-gradient scatter of a GATHER layer needs that does not depend on the gradients: per edge the target
    row (0 = masked / zero weight) and the weight, and the edges' stable order by target row.  Computed with the prepared
    pass (beside the sampling stages) instead of in the backward; handed to ``mpn(..., edge_plan=)``."""
    lib = _lib.load()
    sims2 = sims.reshape(R, -1).contiguous()
    dummy = torch.zeros(max(D, 4), dtype=torch.float32, device=sims2.device)
    ids = ids.reshape(-1, A).contiguous()
    a = _mpn_args(SRC_GATHER, dummy, ids, id_div, None, row_mask, sims2, sim_col, sims_per_edge, dummy, dummy, R, A, D)
    keys = torch.empty(R * A, dtype=torch.int32, device=sims2.device)
    c1 = torch.empty(R * A, dtype=torch.float32, device=sims2.device)
    check(lib.sgnn_mpn_bwd_edges(ctypes.byref(a), None, _ptr(keys), _ptr(c1), None, _stream()), 'sgnn_mpn_bwd_edges')
    return {'keys': keys, 'c1': c1, 'sorted': sort_edges_by_key(keys, max_key), 'ids': ids}


def mpn(x, wp, bp, sims, *, src, R, A, ids=None, id_div=1, edge_mask=None, row_mask=None, sim_col=None,
        sims_per_edge=False, need_agg=True, edge_plan=None, keep_chunks=False, relu_z=False, lazy=False):
    """Fused anchor->component layer body.  x: DENSE (R,A,D) | GATHER E (rows,D) | SHARED (A,D).
    Returns agg (R,D) and the pre-activation read-out z (R,A).  ``keep_chunks``: agg may come back as the (chunks, R, D)
    anchor-chunk partials of a batch-sized call, for a consumer that adds them itself (``update_layer``).  ``relu_z``: the
    read-out comes back activated, relu(z) (what generate_pos_struc_embeddings, mpn:122-131, makes of it next).  ``lazy`` (with
    keep_chunks): the forward launch is queued and goes out with the other bodies of its layer (``flush_lazy_mpn``) -- the caller
    must not read agg or z before that."""
    sims2 = sims.reshape(R, -1)
    if not sims2.is_contiguous():
        sims2 = sims2.contiguous()
    if src == SRC_SHARED and A > 0 and R >= SHARED_GEMM_MIN_ROWS:
        agg, z = _mpn_shared_gemm(x, wp, bp, sims2, ids, row_mask, sim_col, sims_per_edge, R, A, need_agg)
        return agg, (torch.relu(z) if relu_z else z)
    return _MPN.apply(x.contiguous(), wp.contiguous().view(-1), bp.contiguous().view(-1), sims2, ids, edge_mask,
                      row_mask, sim_col, src, id_div, sims_per_edge, R, A, edge_plan, keep_chunks, relu_z, lazy)


class _MaskedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        lib = _lib.load()
        _req(x, torch.float32, 'x')
        _req(mask, torch.uint8, 'mask')
        B, C, H = x.shape
        out = torch.empty((B, H), dtype=torch.float32, device=x.device)
        check(lib.sgnn_masked_sum_fwd(_ptr(x), _ptr(mask), B, C, H, _ptr(out), _stream()), 'sgnn_masked_sum_fwd')
        ctx.save_for_backward(mask)
        ctx.shape = (B, C, H)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        B, C, H = ctx.shape
        g = g.contiguous()
        gx = torch.empty((B, C, H), dtype=torch.float32, device=g.device)
        check(lib.sgnn_masked_sum_bwd(_ptr(g), _ptr(mask), B, C, H, _ptr(gx), _stream()), 'sgnn_masked_sum_bwd')
        return gx, None


def masked_sum(x, mask):
    """subgraph_utils.masked_sum(x, mask.unsqueeze(-1), dim=1) for x (B,C,H), mask (B,C)."""
    return _MaskedSum.apply(x.contiguous(), mask.to(torch.uint8).contiguous())


class ReadoutPiece:
    """The read-out of a layer over SHARED anchors (or with all-zero edge weights) that nobody has materialised yet:
    relu(W * s + bp) with W (R, A) = similarity columns x row mask, s (A) = X wp.  ``subgraph_embedding`` sums it over a
    subgraph's components straight into the embedding's column slot (sgnn_readout_sum_fwd); ``dense()`` is the (B, C, A)
    tensor for a consumer that wants it (attention read-out, gathered heads)."""

    def __init__(self, sims2, sim_col, s, bp, A, row_mask, R, X=None, wp=None, ids=None):
        """``s`` given: the scores as a tensor (the caller made them).  ``s`` None: the scores are X wp (0 where ids == 0; all zero
        without X) and ``subgraph_embedding`` computes them inside its own launches, for all such pieces of a step together
        (sgnn_readout_many_fwd / _bwd: the gradients of X, wp and bp come out of the same two launches)."""
        self.sims2, self.sim_col, self.s, self.bp, self.A, self.row_mask, self.R = sims2, sim_col, s, bp, int(A), row_mask, int(R)
        self.X, self.wp, self.ids = X, (wp.reshape(-1) if wp is not None else None), ids

    def scores(self):
        if self.s is not None:
            return self.s
        if self.X is None:
            return torch.zeros(self.A, dtype=self.bp.dtype, device=self.bp.device)
        s = self.X @ self.wp
        return s * (self.ids != 0).to(s.dtype) if self.ids is not None else s

    def dense(self, B, C):
        if self.sims2 is None:
            return torch.relu(self.bp.view(1, 1, 1).expand(B, C, self.A))
        W = self.sims2.index_select(1, self.sim_col) if self.sim_col is not None else self.sims2[:, :self.A]
        if self.row_mask is not None:
            W = W * (self.row_mask != 0).to(W.dtype).view(-1, 1)
        return torch.relu(_ReadoutShared.apply(W, self.scores(), self.bp)).view(B, C, self.A)


SLOTS_TOGETHER_BELOW = 1 << 22          # (B C H) elements: below, the tensor pieces of a read-out share one launch


class _SubgraphEmbedding(torch.autograd.Function):
    """(B, H) subgraph embedding = [masked sum over components of every piece], written slot by slot: no (B, C, H)
    concatenation (S.py:286-312).  pieces: tensors (B, C, w) and ReadoutPiece objects; tensors: the differentiable
    inputs in piece order (x | s, bp)."""

    @staticmethod
    def forward(ctx, mask, B, C, pieces, *tensors):
        lib = _lib.load()
        _req(mask, torch.uint8, 'mask')
        widths = [p.A if isinstance(p, ReadoutPiece) else p.shape[-1] for p in pieces]
        H = sum(widths)
        out = torch.empty((B, H), dtype=torch.float32, device=mask.device)
        off, k, plan = 0, 0, []
        # batch-sized calls: all tensor pieces in one launch each way (scalar lanes; the shard-sized call keeps a vectorised
        # launch per piece, where the bytes matter and the launches do not)
        n_x = sum(1 for p in pieces if not isinstance(p, ReadoutPiece))
        together = n_x >= 2 and B * C * H <= SLOTS_TOGETHER_BELOW
        slots = []
        many = []                                    # (piece, its column offset, index of its first tensor, its scores buffer)
        for p, w in zip(pieces, widths):
            dst = ctypes.c_void_p(out.data_ptr() + 4 * off)
            if isinstance(p, ReadoutPiece) and p.s is None:
                X, wp, bp = tensors[k], tensors[k + 1], tensors[k + 2]
                _req(X, torch.float32, 'X'), _req(wp, torch.float32, 'wp'), _req(bp, torch.float32, 'bp')
                _req(p.sims2, torch.float32, 'sims'), _req(p.sim_col, torch.int64, 'sim_col'), _req(p.row_mask, torch.uint8, 'row_mask')
                _req(p.ids, torch.int64, 'ids')
                if p.R != B * C or (X is not None and (X.shape[0] != w or wp is None or wp.numel() != X.shape[1])):
                    raise ValueError('read-out piece: %d rows for B C = %d, anchors %s for width %d' % (p.R, B * C, None if X is None else tuple(X.shape), w))
                many.append((p, off, k, torch.empty(w, dtype=torch.float32, device=mask.device)))
                plan.append(('m', off, w, k, p))
                k += 3
            elif isinstance(p, ReadoutPiece):
                s, bp = tensors[k], tensors[k + 1]
                _req(s, torch.float32, 's'), _req(bp, torch.float32, 'bp'), _req(p.sims2, torch.float32, 'sims')
                _req(p.sim_col, torch.int64, 'sim_col'), _req(p.row_mask, torch.uint8, 'row_mask')
                if p.R != B * C or s.numel() != w:
                    raise ValueError('read-out piece: %d rows for B C = %d, %d scores for %d anchors' % (p.R, B * C, s.numel(), w))
                ld = p.sims2.shape[1] if p.sims2 is not None else 0
                check(lib.sgnn_readout_sum_fwd(_ptr(p.sims2), ld, _ptr(p.sim_col), _ptr(s), _ptr(bp), _ptr(p.row_mask), B, C, w,
                                               dst, H, _stream()), 'sgnn_readout_sum_fwd')
                plan.append(('r', off, w, k, p))
                k += 2
            else:
                x = tensors[k]
                _req(x, torch.float32, 'piece')
                if tuple(x.shape) != (B, C, w):
                    raise ValueError('piece %s for (B, C) = (%d, %d)' % (tuple(x.shape), B, C))
                if together:
                    slots.append((x.data_ptr(), w, off))
                else:
                    check(lib.sgnn_masked_sum_slot_fwd(_ptr(x), _ptr(mask), B, C, w, dst, H, _stream()), 'sgnn_masked_sum_slot_fwd')
                plan.append(('x', off, w, k, None))
                k += 1
            off += w
        if slots:
            ptrs, ws, offs = (np.array(v, dtype=t) for v, t in zip(zip(*slots), (np.uint64, np.int64, np.int64)))
            check(lib.sgnn_masked_sum_slots_fwd(ptrs.ctypes.data, ws.ctypes.data, offs.ctypes.data, len(slots), _ptr(mask), B, C,
                                                _ptr(out), H, _stream()), 'sgnn_masked_sum_slots_fwd')
        ctx.many = []
        for group in _readout_groups(many, tensors):
            t = _readout_tables(group, tensors)
            check(lib.sgnn_readout_many_fwd(len(group), t['sims'].ctypes.data, t['ld'].ctypes.data, t['col'].ctypes.data, t['X'].ctypes.data,
                                            t['wp'].ctypes.data, t['bp'].ctypes.data, t['ids'].ctypes.data, t['mask'].ctypes.data,
                                            t['s'].ctypes.data, t['A'].ctypes.data, t['off'].ctypes.data, t['D'], B, C, _ptr(out), H,
                                            _stream()), 'sgnn_readout_many_fwd')
            ctx.many.append(group)
        ctx.plan, ctx.dims, ctx.together = plan, (B, C, H), together
        ctx.save_for_backward(mask, *tensors)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        mask, *tensors = ctx.saved_tensors
        B, C, H = ctx.dims
        g = g.contiguous()
        grads = [None] * len(tensors)
        if ctx.together:
            slots = []
            for kind, off, w, k, p in ctx.plan:
                if kind == 'x' and ctx.needs_input_grad[4 + k]:
                    grads[k] = torch.empty((B, C, w), dtype=torch.float32, device=g.device)
                    slots.append((grads[k].data_ptr(), w, off))
            if slots:
                ptrs, ws, offs = (np.array(v, dtype=t) for v, t in zip(zip(*slots), (np.uint64, np.int64, np.int64)))
                check(lib.sgnn_masked_sum_slots_bwd(_ptr(g), H, _ptr(mask), B, C, ptrs.ctypes.data, ws.ctypes.data, offs.ctypes.data,
                                                    len(slots), _stream()), 'sgnn_masked_sum_slots_bwd')
        for group in ctx.many:
            t = _readout_tables(group, tensors)
            gX, gwp, gbp = [], [], []
            for p, off, k, s_buf in group:
                X = tensors[k]
                gX.append(torch.empty_like(X) if (X is not None and ctx.needs_input_grad[4 + k]) else None)
                gwp.append(torch.empty(X.shape[1], dtype=torch.float32, device=g.device) if (X is not None and ctx.needs_input_grad[5 + k]) else None)
                gbp.append(torch.empty(1, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[6 + k] else None)
            wsb = lib.sgnn_readout_many_bwd_workspace_bytes(len(group), t['A'].ctypes.data, B, C)
            ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=g.device)
            pgX, pgw, pgb = _ptr_table(gX), _ptr_table(gwp), _ptr_table(gbp)
            check(lib.sgnn_readout_many_bwd(len(group), _ptr(g), H, t['sims'].ctypes.data, t['ld'].ctypes.data, t['col'].ctypes.data,
                                            t['X'].ctypes.data, t['wp'].ctypes.data, t['bp'].ctypes.data, t['ids'].ctypes.data,
                                            t['mask'].ctypes.data, t['s'].ctypes.data, t['A'].ctypes.data, t['off'].ctypes.data, t['D'],
                                            B, C, pgX.ctypes.data, pgw.ctypes.data, pgb.ctypes.data, _ptr(ws), wsb,
                                            _ptr(_readout_tickets(g.device)), _stream()), 'sgnn_readout_many_bwd')
            for (p, off, k, s_buf), a, b_, c in zip(group, gX, gwp, gbp):
                grads[k] = a
                grads[k + 1] = b_.view_as(tensors[k + 1]) if b_ is not None else None
                grads[k + 2] = c.view_as(tensors[k + 2]) if c is not None else None
        for kind, off, w, k, p in ctx.plan:
            src = ctypes.c_void_p(g.data_ptr() + 4 * off)
            if kind == 'm':
                continue
            if kind == 'x':
                if ctx.needs_input_grad[4 + k] and not ctx.together:
                    gx = torch.empty((B, C, w), dtype=torch.float32, device=g.device)
                    check(lib.sgnn_masked_sum_slot_bwd(src, H, _ptr(mask), B, C, w, _ptr(gx), _stream()), 'sgnn_masked_sum_slot_bwd')
                    grads[k] = gx
                continue
            s, bp = tensors[k], tensors[k + 1]
            need_s, need_b = ctx.needs_input_grad[4 + k], ctx.needs_input_grad[5 + k]
            if not (need_s or need_b):
                continue
            gs = torch.empty(w, dtype=torch.float32, device=g.device) if need_s else None
            gb = torch.empty(1, dtype=torch.float32, device=g.device) if need_b else None
            wsb = lib.sgnn_readout_sum_bwd_workspace_bytes(B, C, w)
            ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=g.device)
            ld = p.sims2.shape[1] if p.sims2 is not None else 0
            check(lib.sgnn_readout_sum_bwd(src, H, _ptr(p.sims2), ld, _ptr(p.sim_col), _ptr(s), _ptr(bp), _ptr(p.row_mask), B, C, w,
                                           _ptr(gs), _ptr(gb), _ptr(ws), wsb, _stream()), 'sgnn_readout_sum_bwd')
            grads[k] = gs.view_as(s) if gs is not None else None
            grads[k + 1] = gb.view_as(bp) if gb is not None else None
        return (None, None, None, None) + tuple(grads)


_RO_TICKETS = {}


def _readout_tickets(device):
    t = _RO_TICKETS.get(str(device))
    if t is None:
        t = _RO_TICKETS[str(device)] = torch.zeros(int(_lib.load().sgnn_readout_many_max()), dtype=torch.int32, device=device)
    return t


def _readout_groups(many, tensors):
    """The in-kernel read-out pieces of a call in launch groups: up to sgnn_readout_many_max() pieces of one anchor width."""
    if not many:
        return []
    cap = int(_lib.load().sgnn_readout_many_max())
    by_d = {}
    for ent in many:
        X = tensors[ent[2]]
        by_d.setdefault(None if X is None else int(X.shape[1]), []).append(ent)
    free = by_d.pop(None, [])                         # (pieces without anchors -- all-zero similarities -- ride with any width)
    widths = sorted(by_d) or [1]
    by_d.setdefault(widths[0], [])
    by_d[widths[0]] = free + by_d[widths[0]]
    groups = []
    for d in widths:
        ents = sorted(by_d[d], key=lambda e: e[1])
        groups += [ents[lo:lo + cap] for lo in range(0, len(ents), cap)]
    return groups


def _readout_tables(group, tensors):
    """Host tables of one many-piece launch (kept alive by the caller for the duration of the call)."""
    D = 1
    for p, off, k, s_buf in group:
        if tensors[k] is not None:
            D = int(tensors[k].shape[1])
    return {'sims': _ptr_table([p.sims2 for p, *_ in group]),
            'ld': np.array([p.sims2.shape[1] if p.sims2 is not None else 0 for p, *_ in group], dtype=np.int64),
            'col': _ptr_table([p.sim_col for p, *_ in group]), 'X': _ptr_table([tensors[k] for _, _, k, _ in group]),
            'wp': _ptr_table([tensors[k + 1] for _, _, k, _ in group]), 'bp': _ptr_table([tensors[k + 2] for _, _, k, _ in group]),
            'ids': _ptr_table([p.ids for p, *_ in group]), 'mask': _ptr_table([p.row_mask for p, *_ in group]),
            's': _ptr_table([s_buf for *_, s_buf in group]), 'A': np.array([p.A for p, *_ in group], dtype=np.int64),
            'off': np.array([off for _, off, _, _ in group], dtype=np.int64), 'D': D}


def subgraph_embedding(pieces, mask, B, C):
    """cat(pieces, -1) summed over each subgraph's real components -> (B, H); mask (B C) uint8."""
    tensors = []
    for p in pieces:
        if isinstance(p, ReadoutPiece) and p.s is None:
            tensors += [p.X.contiguous() if p.X is not None else None, p.wp.contiguous() if p.wp is not None else None,
                        p.bp.contiguous().view(-1)]
        elif isinstance(p, ReadoutPiece):
            tensors += [p.s.contiguous(), p.bp.contiguous().view(-1)]
        else:
            tensors.append(p.contiguous())
    return _SubgraphEmbedding.apply(mask.reshape(-1).contiguous(), int(B), int(C), list(pieces), *tensors)


class _CrossEntropy(torch.autograd.Function):
    """(mean cross entropy, accuracy) of logits (B, K) against int64 labels (sgnn_cross_entropy_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, logits, labels):
        lib = _lib.load()
        _req(logits, torch.float32, 'logits')
        _req(labels, torch.int64, 'labels')
        B, K = logits.shape
        lse = torch.empty(B + 1, dtype=torch.float32, device=logits.device)      # [B]: rows not ignored (label -100)
        res = torch.empty(2, dtype=torch.float32, device=logits.device)
        wsb = lib.sgnn_cross_entropy_workspace_bytes(B)
        ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=logits.device)
        check(lib.sgnn_cross_entropy_fwd(_ptr(logits), _ptr(labels), B, K, _ptr(lse), _ptr(res), ctypes.c_void_p(res.data_ptr() + 4),
                                         _ptr(ws), wsb, _stream()), 'sgnn_cross_entropy_fwd')
        ctx.save_for_backward(logits, labels, lse)
        loss, acc = res[0], res[1:2]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g_loss, _g_acc):
        lib = _lib.load()
        logits, labels, lse = ctx.saved_tensors
        B, K = logits.shape
        g = torch.empty_like(logits)
        check(lib.sgnn_cross_entropy_bwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(g_loss.reshape(1).contiguous().float()), B, K,
                                         _ptr(g), _stream()), 'sgnn_cross_entropy_bwd')
        return g, None


def cross_entropy_with_accuracy(logits, labels):
    """nn.CrossEntropyLoss()(logits, labels) and calc_accuracy(logits, labels) -> (0-d loss, (1,) accuracy) in one pass."""
    return _CrossEntropy.apply(logits.contiguous(), labels.contiguous())


# ---------------------------------------------------------------------------------------
# the MLP head + loss of a step (csrc/head.hip)
# ---------------------------------------------------------------------------------------

_HEAD_WS = {}


def _head_workspace(device, B):
    """The forward's per-workgroup loss partials + its ticket: one persistent buffer per device and workgroup count (the ticket
    must be zero before the first launch and every launch leaves it zero; the partials are rewritten by every launch)."""
    lib = _lib.load()
    nb = int(lib.sgnn_head_blocks(int(B)))
    key = (str(device), nb)
    ws = _HEAD_WS.get(key)
    if ws is None:
        ws = _HEAD_WS[key] = torch.zeros(int(lib.sgnn_head_fwd_workspace_bytes(int(B))) // 4 + 1, dtype=torch.int32, device=device)
    return ws


def head_supported(H1, H2, K):
    return bool(_lib.load().sgnn_head_supported(int(H1), int(H2), int(K)))


def contract_rows_many(pairs, column_sums=()):
    """[a^T b for (a, b) in pairs] for tall row-major a (R, M), b (R, N) (row strides may exceed the widths): block partials on the
    matrix cores in one launch per group of pairs (sgnn_contract_rows_partial), added in block order by one sgnn_reduce_partials
    launch -- bit-reproducible.  -> list of (M, N) tensors.  ``column_sums``: {pair index: k} -- pair i also yields k separate
    copies of a.sum(0) (a Linear's bias gradient rides along with its weight's: no extra pass over a); they follow the products in
    the returned list, in index order."""
    lib = _lib.load()
    if column_sums:
        return _contract_rows_with_sums(pairs, dict(column_sums))
    outs = []
    cap = min(int(lib.sgnn_contract_rows_max_jobs()), int(lib.sgnn_reduce_partials_max_jobs()))
    for lo in range(0, len(pairs), cap):
        group = pairs[lo:lo + cap]
        for a, b in group:
            _req2d(a, 'a'), _req2d(b, 'b')
            if a.shape[0] != b.shape[0]:
                raise ValueError('contract_rows_many: %d and %d rows' % (a.shape[0], b.shape[0]))
        R = np.array([a.shape[0] for a, _ in group], dtype=np.int64)
        M = np.array([a.shape[1] for a, _ in group], dtype=np.int64)
        N = np.array([b.shape[1] for _, b in group], dtype=np.int64)
        lda = np.array([a.stride(0) for a, _ in group], dtype=np.int64)
        ldb = np.array([b.stride(0) for _, b in group], dtype=np.int64)
        nb = np.array([int(lib.sgnn_contract_rows_blocks(int(r), int(m), int(n_))) for r, m, n_ in zip(R, M, N)], dtype=np.int64)
        dev = group[0][0].device
        parts = [torch.empty((int(nb[k]), int(M[k]), int(N[k])), dtype=torch.float32, device=dev) for k in range(len(group))]
        res = [torch.empty((int(M[k]), int(N[k])), dtype=torch.float32, device=dev) for k in range(len(group))]
        pa, pb, pp, po = (_ptr_table(v) for v in ([a for a, _ in group], [b for _, b in group], parts, res))
        check(lib.sgnn_contract_rows_partial(len(group), pa.ctypes.data, pb.ctypes.data, lda.ctypes.data, ldb.ctypes.data,
                                             M.ctypes.data, N.ctypes.data, R.ctypes.data, pp.ctypes.data, None, _stream()),
              'sgnn_contract_rows_partial')
        n = M * N
        check(lib.sgnn_reduce_partials(len(group), pp.ctypes.data, nb.ctypes.data, n.ctypes.data, po.ctypes.data, _stream()),
              'sgnn_reduce_partials')
        outs += res
    return outs


def _contract_rows_with_sums(pairs, sums):
    """contract_rows_many for one launch group whose pairs ``sums`` (index -> copies) also deliver column sums of ``a``."""
    lib = _lib.load()
    n_red = len(pairs) + sum(sums.values())
    if len(pairs) > int(lib.sgnn_contract_rows_max_jobs()) or n_red > int(lib.sgnn_reduce_partials_max_jobs()):
        raise ValueError('contract_rows_many(column_sums=): %d products + %d sums exceed one launch group' % (len(pairs), n_red - len(pairs)))
    for a, b in pairs:
        _req2d(a, 'a'), _req2d(b, 'b')
    dev = pairs[0][0].device
    R = np.array([a.shape[0] for a, _ in pairs], dtype=np.int64)
    M = np.array([a.shape[1] for a, _ in pairs], dtype=np.int64)
    N = np.array([b.shape[1] for _, b in pairs], dtype=np.int64)
    lda = np.array([a.stride(0) for a, _ in pairs], dtype=np.int64)
    ldb = np.array([b.stride(0) for _, b in pairs], dtype=np.int64)
    nb = [int(lib.sgnn_contract_rows_blocks(int(r), int(m), int(n_))) for r, m, n_ in zip(R, M, N)]
    parts = [torch.empty((nb[k], int(M[k]), int(N[k])), dtype=torch.float32, device=dev) for k in range(len(pairs))]
    cparts = [torch.empty((nb[k], int(M[k])), dtype=torch.float32, device=dev) if k in sums else None for k in range(len(pairs))]
    res = [torch.empty((int(M[k]), int(N[k])), dtype=torch.float32, device=dev) for k in range(len(pairs))]
    extra, red_part, red_nb, red_n = [], list(parts), list(nb), [int(M[k] * N[k]) for k in range(len(pairs))]
    for k in sorted(sums):
        for _ in range(sums[k]):
            extra.append(torch.empty(int(M[k]), dtype=torch.float32, device=dev))
            red_part.append(cparts[k]), red_nb.append(nb[k]), red_n.append(int(M[k]))
    pa, pb, pp, pc = (_ptr_table(v) for v in ([a for a, _ in pairs], [b for _, b in pairs], parts, cparts))
    check(lib.sgnn_contract_rows_partial(len(pairs), pa.ctypes.data, pb.ctypes.data, lda.ctypes.data, ldb.ctypes.data, M.ctypes.data,
                                         N.ctypes.data, R.ctypes.data, pp.ctypes.data, pc.ctypes.data, _stream()),
          'sgnn_contract_rows_partial')
    rp, ro = _ptr_table(red_part), _ptr_table(res + extra)
    rnb, rn = np.array(red_nb, dtype=np.int64), np.array(red_n, dtype=np.int64)
    check(lib.sgnn_reduce_partials(len(red_part), rp.ctypes.data, rnb.ctypes.data, rn.ctypes.data, ro.ctypes.data, _stream()),
          'sgnn_reduce_partials')
    return res + extra


def _req2d(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        raise ValueError('%s must be a float32 CUDA matrix with unit column stride' % name)


class _FusedHead(torch.autograd.Function):
    """logits, loss, accuracy = head(x): lin -> relu -> dropout -> lin2 -> relu -> dropout -> lin3 [-> cross entropy + accuracy]
    (SubGNN.py:304-312, 1116-1124).  The first layer's GEMMs (z1 = x W1^T + b1, dx = dz1 W1) are the library's; everything else
    is sgnn_head_fwd / sgnn_head_bwd, gW1 = dz1^T x on the matrix cores (sgnn_contract_rows_partial), and ONE reduction launch for
    every weight and bias gradient of the head.  9 + 22 launches -> 2 + 4."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, labels, p, rng):
        lib = _lib.load()
        for t, nm in ((x, 'x'), (W1, 'W1'), (b1, 'b1'), (W2, 'W2'), (b2, 'b2'), (W3, 'W3'), (b3, 'b3')):
            _req(t, torch.float32, nm)
        _req(labels, torch.int64, 'labels')
        _req(rng, torch.int64, 'rng')
        B, H1, H2, K = x.shape[0], W1.shape[0], W2.shape[0], W3.shape[0]
        dev = x.device
        z1 = torch.addmm(b1, x, W1.t()) if b1 is not None else x @ W1.t()
        a1 = torch.empty((B, H1), dtype=torch.float32, device=dev)
        a2 = torch.empty((B, H2), dtype=torch.float32, device=dev)
        logits = torch.empty((B, K), dtype=torch.float32, device=dev)
        lse = torch.empty(B + 1, dtype=torch.float32, device=dev) if labels is not None else None
        out = torch.empty(3, dtype=torch.float32, device=dev) if labels is not None else None
        ws = _head_workspace(dev, B)
        check(lib.sgnn_head_fwd(_ptr(z1), B, H1, H2, K, _ptr(W2), _ptr(b2), _ptr(W3), _ptr(b3), _ptr(labels), float(p), _ptr(rng),
                                _ptr(a1), _ptr(a2), _ptr(logits), _ptr(lse), _ptr(out), _ptr(ws), ws.numel() * 4, _stream()),
              'sgnn_head_fwd')
        ctx.save_for_backward(x, W1, W2, W3, a1, a2, logits, lse, labels, out)
        ctx.p, ctx.dims = float(p), (B, H1, H2, K)
        ctx.has_bias = (b1 is not None, b2 is not None, b3 is not None)
        ctx.set_materialize_grads(False)
        if labels is None:
            return logits, None, None
        loss, acc = out[0], out[1:2]
        ctx.mark_non_differentiable(acc)
        return logits, loss, acc

    @staticmethod
    def backward(ctx, g_logits, g_loss, _g_acc):
        lib = _lib.load()
        x, W1, W2, W3, a1, a2, logits, lse, labels, out = ctx.saved_tensors
        B, H1, H2, K = ctx.dims
        dev = x.device
        if g_logits is None and g_loss is None:
            return (None,) * 10
        g_logits = g_logits.contiguous() if g_logits is not None else None
        g_loss = g_loss.reshape(1).contiguous().float() if g_loss is not None else None
        P = int(lib.sgnn_head_partial_floats(H1, H2, K))
        nb = int(lib.sgnn_head_blocks(B))
        dz1 = torch.empty((B, H1), dtype=torch.float32, device=dev)
        partial = torch.empty((nb, P), dtype=torch.float32, device=dev)
        check(lib.sgnn_head_bwd(_ptr(logits), _ptr(lse), _ptr(labels), _ptr(g_loss), _ptr(g_logits),
                                ctypes.c_void_p(out.data_ptr() + 8) if out is not None else None, _ptr(a1), _ptr(a2), _ptr(W2),
                                _ptr(W3), B, H1, H2, K, ctx.p, _ptr(dz1), _ptr(partial), _stream()), 'sgnn_head_bwd')
        # gW1 = dz1^T x: block partials on the matrix cores; its blocks and the head kernel's partials are added by ONE launch
        flat = torch.empty(P, dtype=torch.float32, device=dev)
        jobs_part, jobs_nb, jobs_n, jobs_out = [partial], [nb], [P], [flat]
        gW1 = None
        if ctx.needs_input_grad[1]:
            H0 = x.shape[1]
            nbw = int(lib.sgnn_contract_rows_blocks(B, H1, H0))
            pw = torch.empty((nbw, H1, H0), dtype=torch.float32, device=dev)
            gW1 = torch.empty((H1, H0), dtype=torch.float32, device=dev)
            # (the host arrays must outlive the call: named, not temporaries)
            pa, pb, pp = _ptr_table([dz1]), _ptr_table([x]), _ptr_table([pw])
            dims = np.array([H1, x.stride(0), H1, H0, B], dtype=np.int64)
            check(lib.sgnn_contract_rows_partial(1, pa.ctypes.data, pb.ctypes.data, dims[0:1].ctypes.data, dims[1:2].ctypes.data,
                                                 dims[2:3].ctypes.data, dims[3:4].ctypes.data, dims[4:5].ctypes.data,
                                                 pp.ctypes.data, None, _stream()), 'sgnn_contract_rows_partial')
            jobs_part.append(pw), jobs_nb.append(nbw), jobs_n.append(H1 * H0), jobs_out.append(gW1)
        rp, ro = _ptr_table(jobs_part), _ptr_table(jobs_out)
        rnb, rn = np.array(jobs_nb, dtype=np.int64), np.array(jobs_n, dtype=np.int64)
        check(lib.sgnn_reduce_partials(len(jobs_part), rp.ctypes.data, rnb.ctypes.data, rn.ctypes.data, ro.ctypes.data, _stream()),
              'sgnn_reduce_partials')
        o = 0
        gW3 = flat[o:o + K * H2].view(K, H2); o += K * H2
        gb3 = flat[o:o + K]; o += K
        gW2 = flat[o:o + H2 * H1].view(H2, H1); o += H2 * H1
        gb2 = flat[o:o + H2]; o += H2
        gb1 = flat[o:o + H1]
        hb = ctx.has_bias
        need = ctx.needs_input_grad
        dx = dz1 @ W1 if need[0] else None
        return (dx, gW1, gb1 if (hb[0] and need[2]) else None, gW2 if need[3] else None, gb2 if (hb[1] and need[4]) else None,
                gW3 if need[5] else None, gb3 if (hb[2] and need[6]) else None, None, None, None)


def fused_head(x, lin, lin2, lin3, labels=None, p=0.0, rng=None):
    """The head's three Linear layers with relu + dropout(p) between them, and -- with ``labels`` -- the mean cross entropy and
    the accuracy of the logits (``lin*``: nn.Linear modules).  -> (logits, loss or None, accuracy (1,) or None).
    ``rng``: int64 (2,) device tensor {seed, step} when p > 0 (the launch advances step)."""
    if p > 0 and rng is None:
        raise ValueError('fused_head: dropout needs the {seed, step} tensor')
    if x.stride(1) != 1 or x.stride(0) != x.shape[1]:
        x = x.contiguous()
    return _FusedHead.apply(x, lin.weight, lin.bias, lin2.weight, lin2.bias, lin3.weight, lin3.bias,
                            labels.contiguous() if labels is not None else None, float(p), rng if p > 0 else None)


class _GatherRows(torch.autograd.Function):
    """``nn.Embedding(padding_idx=0)`` lookup for a handful of ids (the shared P anchors, the walks
    of the structure patches): the backward scatters the few rows with ``index_add_`` instead of
    torch's dense-embedding backward, which serialises a short id list on two workgroups."""

    @staticmethod
    def forward(ctx, weight, ids, presorted=None):
        flat = ids.reshape(-1)
        ctx.presorted = presorted
        ctx.save_for_backward(flat)
        ctx.n_rows = weight.shape[0]
        ctx.det = _det_now()
        ctx.acc = getattr(weight, '_sgnn_acc', None)
        half = getattr(weight, '_sgnn_half', None)
        src = weight if half is None else half
        return src.index_select(0, flat).to(torch.float32).view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, grad):
        flat, = ctx.saved_tensors
        if ctx.det and grad.is_cuda and grad.shape[-1] <= 256:
            g = grad.reshape(flat.numel(), -1).to(torch.float32).contiguous()
            buf = ctx.acc.buffer((ctx.n_rows, g.shape[1]), grad.device) if ctx.acc is not None else \
                torch.zeros(ctx.n_rows, g.shape[1], dtype=torch.float32, device=grad.device)
            pre = ctx.presorted
            if pre is not None and pre[0].numel() == flat.numel():
                scatter_add_rows(buf, pre[2], G=g, edges_per_row=1, presorted=pre[:2], together=ctx.acc)
            else:
                scatter_add_rows(buf, flat.to(torch.int32).contiguous(), G=g, edges_per_row=1, together=ctx.acc)      # key 0 = PAD: skipped
            return (None if ctx.acc is not None else buf.to(grad.dtype)), None, None
        g = grad.reshape(flat.numel(), -1) * (flat != 0).unsqueeze(1).to(grad.dtype)     # PAD row takes no gradient
        if ctx.acc is not None:
            ctx.acc.buffer((ctx.n_rows, g.shape[1]), grad.device).index_add_(0, flat, g.to(torch.float32))
            return None, None, None
        return torch.zeros(ctx.n_rows, g.shape[1], dtype=grad.dtype, device=grad.device).index_add_(0, flat, g), None, None


_INDEX_FLAGS = {}            # device -> (int32 flag tensor the gather kernels set, pinned host twin, event of the last copy)


def _index_flag(device):
    ent = _INDEX_FLAGS.get(str(device))
    if ent is None:
        ent = _INDEX_FLAGS[str(device)] = [torch.zeros(1, dtype=torch.int32, device=device),
                                           torch.zeros(1, dtype=torch.int32).pin_memory(), None]
    return ent


def poll_index_errors(block=False):
    """Raise IndexError if a device-side row gather (index_rows_many with a device-resident index vector, whose values the host
    never sees) met an index outside its source since the last poll.  The kernel cannot raise: it writes a zero row and sets a
    flag; the flag is copied to pinned memory behind every gather and read here once the copy has landed (``block``: wait for
    it -- the epoch-end callers, which read results back anyway).  A recorded step's gathers set the same flag."""
    if not _INDEX_FLAGS or torch.cuda.is_current_stream_capturing():
        return
    for dev, ent in list(_INDEX_FLAGS.items()):
        flag, host, ev = ent
        if block:
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
        elif ev is None or not ev.query():
            continue
        ent[2] = None
        if int(host[0]) != 0:
            flag.zero_()
            host.zero_()
            raise IndexError('a batch index on %s was out of range for its split (device-side gather: the rows of such an index '
                             'are zero-filled)' % dev)


def index_rows_many(tensors, idx):
    """[t.index_select(0, idx) for t in tensors] in one launch (sgnn_gather_rows_many): the row gathers that assemble a batch
    from a split's per-subgraph tensors.  ``tensors``: contiguous CUDA tensors with the same number of rows; ``idx``: int64 device
    tensor.  Tensors it cannot take (not contiguous, other devices) go through index_select."""
    lib = _lib.load()
    if idx.is_cuda:
        poll_index_errors()                          # (an earlier gather's flag, if its copy has landed: never a wait)
    out = [None] * len(tensors)
    take = [k for k, t in enumerate(tensors) if t.is_cuda and t.is_contiguous() and t.dim() >= 1 and t.shape[0] > 0 and t.numel() > 0]
    if not idx.is_cuda or idx.dtype != torch.int64 or not idx.is_contiguous() or idx.numel() == 0 or len(take) < 2:
        take = []
    for k, t in enumerate(tensors):
        if k not in take:
            out[k] = t.index_select(0, idx)
    cap = int(lib.sgnn_gather_rows_many_max()) if take else 1
    B = idx.numel()
    for lo in range(0, len(take), cap):
        group = take[lo:lo + cap]
        for k in group:
            out[k] = torch.empty((B,) + tuple(tensors[k].shape[1:]), dtype=tensors[k].dtype, device=tensors[k].device)
        src, dst = _ptr_table([tensors[k] for k in group]), _ptr_table([out[k] for k in group])
        rb = np.array([tensors[k][0].numel() * tensors[k].element_size() for k in group], dtype=np.int64)
        rows = np.array([tensors[k].shape[0] for k in group], dtype=np.int64)
        ent = _index_flag(idx.device)
        check(lib.sgnn_gather_rows_many(len(group), src.ctypes.data, dst.ctypes.data, rb.ctypes.data, rows.ctypes.data, _ptr(idx), B,
                                        _ptr(ent[0]), _stream()), 'sgnn_gather_rows_many')
        if not torch.cuda.is_current_stream_capturing():
            ent[1].copy_(ent[0], non_blocking=True)            # read by poll_index_errors once it has landed
            ent[2] = torch.cuda.Event()
            ent[2].record()
    return out


def presort_ids(ids, max_key):
    """Hang the stable order of ``ids`` by value on the tensor (``_sgnn_sorted`` = (sorted keys, order, int32 ids)): the
    backward of ``gather_rows(table, ids)`` then skips its sort.  For id tensors that live as long as a prepared pass."""
    k32 = ids.reshape(-1).to(torch.int32).contiguous()
    ids._sgnn_sorted = sort_edges_by_key(k32, max_key) + (k32,)
    return ids


def gather_rows(weight, ids):
    """weight[ids] with the PAD row (id 0) excluded from the gradient (aps:404-411 embed_anchor_patch)."""
    return _GatherRows.apply(weight, ids.to(torch.int64), getattr(ids, '_sgnn_sorted', None))


class _BiLSTMLayer(torch.autograd.Function):
    """One bidirectional LSTM layer: x (B, T, I) -> (B, T, 2H), zero initial state.  Parameters in
    torch's nn.LSTM layout (weight_ih (4H, I), weight_hh (4H, H), two biases, per direction).  The
    recurrence runs in sgnn_lstm_fwd / _bwd; the input projection and the weight / input gradients
    are GEMMs over all (sequence, step) rows at once."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        lib = _lib.load()
        _req(x, torch.float32, 'x')
        B, T, I = x.shape
        H = w_hh_f.shape[1]
        x2 = x.reshape(B * T, I)
        dev = x.device
        # direction-major projected inputs: each direction's rows are one GEMM's contiguous output (no concatenation of the two
        # weight matrices); bias_hh is added inside the recurrence kernel (no sum of the bias vectors, no stacked W_hh)
        pre_x = torch.empty((2, B * T, 4 * H), dtype=torch.float32, device=dev)
        torch.addmm(b_ih_f, x2, w_ih_f.t(), out=pre_x[0])
        torch.addmm(b_ih_r, x2, w_ih_r.t(), out=pre_x[1])
        y = torch.empty((B, T, 2 * H), dtype=torch.float32, device=dev)
        gates = torch.empty((2, B, T, 4 * H), dtype=torch.float32, device=dev)
        cell = torch.empty((2, B, T, H), dtype=torch.float32, device=dev)
        hprev = torch.empty((2, B, T, H), dtype=torch.float32, device=dev)
        for t, nm in ((w_hh_f, 'weight_hh'), (w_hh_r, 'weight_hh_reverse'), (b_hh_f, 'bias_hh'), (b_hh_r, 'bias_hh_reverse')):
            _req(t, torch.float32, nm)
        check(lib.sgnn_lstm_fwd(_ptr(pre_x), _ptr(w_hh_f), _ptr(w_hh_r), _ptr(b_hh_f), _ptr(b_hh_r), B, T, H, _ptr(y), _ptr(gates),
                                _ptr(cell), _ptr(hprev), _stream()), 'sgnn_lstm_fwd')
        ctx.save_for_backward(x2, w_ih_f, w_ih_r, w_hh_f, w_hh_r, gates, cell, hprev)
        ctx.dims = (B, T, I, H)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, w_ih_f, w_ih_r, w_hh_f, w_hh_r, gates, cell, hprev = ctx.saved_tensors
        B, T, I, H = ctx.dims
        dy = dy.contiguous()
        dgates = torch.empty_like(gates)                                        # (2, B, T, 4H)
        check(lib.sgnn_lstm_bwd(_ptr(w_hh_f), _ptr(w_hh_r), _ptr(gates), _ptr(cell), _ptr(dy), B, T, H, _ptr(dgates), _stream()),
              'sgnn_lstm_bwd')
        dg = dgates.view(2, B * T, 4 * H)
        dx = torch.addmm(dg[0] @ w_ih_f, dg[1], w_ih_r).view(B, T, I) if ctx.needs_input_grad[0] else None
        # both directions' weight gradients in ONE batched contraction each, the rows split into blocks when there are enough
        # of them (library GEMMs of 512 x 128 outputs over a few thousand rows ran 26 us apiece on a handful of workgroups);
        # a direction's gate gradients are contiguous: no transposed copy of dgates (75 MB at 1850 x 10 steps, H = 128)
        dwih = contract_rows_batched(dg, x2.unsqueeze(0).expand(2, B * T, I))   # (2, 4H, I)
        dwhh = contract_rows_batched(dg, hprev.view(2, B * T, H))               # (2, 4H, H)
        # bias_ih and bias_hh receive the same gradient VALUES but must not receive the same MEMORY: autograd hands a view
        # over to .grad as it is, and an in-place multi-tensor update of the gradient list (clip_grad_norm_'s _foreach_mul_)
        # then scales the shared buffer once per alias, from concurrently running chunks -- g c or g c^2 depending on timing
        # (round 4: the cross-process 7th-digit loss drift was this race on lstm.bias_{ih,hh}_l0_reverse).  One copy:
        db_f, db_r = column_sum(dg[0]), column_sum(dg[1])
        db = torch.stack((db_f, db_r, db_f, db_r))                              # [ih: forward, reverse | hh: forward, reverse]
        return (dx, dwih[0], dwhh[0], db[0], db[2], dwih[1], dwhh[1], db[1], db[3])


LSTM_OWN_GEMM_MAX_H = 64       # hidden sizes up to here take this library's projection / dx kernels; wider ones the library's GEMMs


class _BiLSTMLayerFused(torch.autograd.Function):
    """_BiLSTMLayer with its dense products in this library's launches (csrc/lstm.hip): the input projection of both directions
    in ONE fp32-MFMA launch whose operand load IS the embedding lookup when ``ids`` is given (src = the table, ids (B, T) node ids;
    else src = the dense input (B, T, I)); backward: the recurrence, ONE launch for the block partials of every weight and bias
    gradient of both directions + one reduction (ops.contract_rows_many), one launch for dx -- which, for a lookup, joins the
    step's combined table-gradient scatter.  3 + 2 launches forward -> 2, ~20 backward -> 4."""

    @staticmethod
    def forward(ctx, src, ids, presorted, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        lib = _lib.load()
        for t, nm in ((src, 'input'), (w_ih_f, 'weight_ih'), (w_hh_f, 'weight_hh'), (b_ih_f, 'bias_ih'), (b_hh_f, 'bias_hh'),
                      (w_ih_r, 'weight_ih_reverse'), (w_hh_r, 'weight_hh_reverse'), (b_ih_r, 'bias_ih_reverse'), (b_hh_r, 'bias_hh_reverse')):
            _req(t, torch.float32, nm)
        gather = ids is not None
        if gather:
            _req(ids, torch.int64, 'ids')
            B, T = ids.shape
            I = src.shape[1]
        else:
            B, T, I = src.shape
        H = w_hh_f.shape[1]
        R = B * T
        dev = src.device
        pre_x = torch.empty((2, R, 4 * H), dtype=torch.float32, device=dev)
        x2 = torch.empty((R, I), dtype=torch.float32, device=dev) if gather else src.view(R, I)
        if gather or H <= LSTM_OWN_GEMM_MAX_H:
            check(lib.sgnn_rows_gemm(_ptr(src), _ptr(ids), I, R, I, _ptr(w_ih_f), _ptr(w_ih_r), _ptr(b_ih_f), _ptr(b_ih_r), 4 * H,
                                     _ptr(pre_x), _ptr(x2) if gather else None, _stream()), 'sgnn_rows_gemm')
        else:
            # a plain dense GEMM of a few thousand rows by 512 columns over K = 256: the library's tiles run it at twice the rate
            # of the one-wavefront-per-32x32 kernel (measured on the PPI-BP stand-in's second LSTM layer: 31 us against 67)
            torch.addmm(b_ih_f, x2, w_ih_f.t(), out=pre_x[0])
            torch.addmm(b_ih_r, x2, w_ih_r.t(), out=pre_x[1])
        y = torch.empty((B, T, 2 * H), dtype=torch.float32, device=dev)
        gates = torch.empty((2, B, T, 4 * H), dtype=torch.float32, device=dev)
        cell = torch.empty((2, B, T, H), dtype=torch.float32, device=dev)
        hprev = torch.empty((2, B, T, H), dtype=torch.float32, device=dev)
        check(lib.sgnn_lstm_fwd(_ptr(pre_x), _ptr(w_hh_f), _ptr(w_hh_r), _ptr(b_hh_f), _ptr(b_hh_r), B, T, H, _ptr(y), _ptr(gates),
                                _ptr(cell), _ptr(hprev), _stream()), 'sgnn_lstm_fwd')
        ctx.save_for_backward(x2, ids, w_ih_f, w_ih_r, w_hh_f, w_hh_r, gates, cell, hprev)
        ctx.dims = (B, T, I, H)
        ctx.gather, ctx.presorted = gather, presorted
        ctx.acc = getattr(src, '_sgnn_acc', None) if gather else None
        ctx.n_rows = src.shape[0] if gather else 0
        ctx.det = _det_now()
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, ids, w_ih_f, w_ih_r, w_hh_f, w_hh_r, gates, cell, hprev = ctx.saved_tensors
        B, T, I, H = ctx.dims
        R = B * T
        dy = dy.contiguous()
        dgates = torch.empty_like(gates)
        check(lib.sgnn_lstm_bwd(_ptr(w_hh_f), _ptr(w_hh_r), _ptr(gates), _ptr(cell), _ptr(dy), B, T, H, _ptr(dgates), _stream()),
              'sgnn_lstm_bwd')
        dg = dgates.view(2, R, 4 * H)
        hp = hprev.view(2, R, H)
        # every weight and bias gradient of both directions: block partials in one launch, one reduction; the bias gradients are
        # the column sums of the gate gradients the dW_ih jobs read anyway.  (bias_ih and bias_hh receive the same VALUES in
        # DIFFERENT memory -- two outputs of the reduction: an in-place multi-tensor update of the gradient list must not meet one
        # buffer twice, ops._BiLSTMLayer's note.)
        dwih_f, dwih_r, dwhh_f, dwhh_r, dbi_f, dbh_f, dbi_r, dbh_r = contract_rows_many(
            [(dg[0], x2), (dg[1], x2), (dg[0], hp[0]), (dg[1], hp[1])], column_sums={0: 2, 1: 2})
        g_src = None
        if ctx.needs_input_grad[0]:
            if H <= LSTM_OWN_GEMM_MAX_H:
                dx = torch.empty((R, I), dtype=torch.float32, device=dy.device)
                check(lib.sgnn_rows_gemm_nt(_ptr(dgates), R, 4 * H, _ptr(w_ih_f), _ptr(w_ih_r), I, _ptr(dx), _stream()), 'sgnn_rows_gemm_nt')
            else:
                dx = torch.addmm(dg[0] @ w_ih_f, dg[1], w_ih_r)       # (K = 4 H = 512 per direction: the library's GEMM, as above)
            if not ctx.gather:
                g_src = dx.view(B, T, I)
            else:
                flat = ids.reshape(-1)
                if ctx.det and I <= 256:
                    buf = ctx.acc.buffer((ctx.n_rows, I), dy.device) if ctx.acc is not None else \
                        torch.zeros(ctx.n_rows, I, dtype=torch.float32, device=dy.device)
                    pre = ctx.presorted
                    if pre is not None and pre[0].numel() == flat.numel():
                        scatter_add_rows(buf, pre[2], G=dx, edges_per_row=1, presorted=pre[:2], together=ctx.acc)
                    else:
                        scatter_add_rows(buf, flat.to(torch.int32).contiguous(), G=dx, edges_per_row=1, together=ctx.acc)   # key 0 = PAD: skipped
                    g_src = None if ctx.acc is not None else buf
                else:
                    dxm = dx * (flat != 0).unsqueeze(1).to(dx.dtype)                       # PAD row takes no gradient
                    if ctx.acc is not None:
                        ctx.acc.buffer((ctx.n_rows, I), dy.device).index_add_(0, flat, dxm)
                    else:
                        g_src = torch.zeros(ctx.n_rows, I, dtype=torch.float32, device=dy.device).index_add_(0, flat, dxm)
        return (g_src, None, None, dwih_f, dwhh_f, dbi_f.view(-1), dbh_f.view(-1), dwih_r, dwhh_r, dbi_r.view(-1), dbh_r.view(-1))


class _LSTMTail(torch.autograd.Function):
    """X (n, D) = sum over a patch's n_walks sequences of Linear(last step | sum over steps) (SubGNN.py:60-88 + aps:413-433's sum):
    one launch each way (sgnn_lstm_tail_fwd / _bwd) instead of the slice, the Linear's GEMM and the sum -- and, backward, the
    zero-filled gradient of the slice, its copy, the Linear's three products and the expansion of the sum."""

    @staticmethod
    def forward(ctx, y, W_lin, b_lin, n_walks, last_only):
        lib = _lib.load()
        for t, nm in ((y, 'y'), (W_lin, 'linear.weight'), (b_lin, 'linear.bias')):
            _req(t, torch.float32, nm)
        Bq, T, H2 = y.shape
        n = Bq // n_walks
        D = W_lin.shape[0]
        s = torch.empty((n, H2), dtype=torch.float32, device=y.device)
        X = torch.empty((n, D), dtype=torch.float32, device=y.device)
        check(lib.sgnn_lstm_tail_fwd(_ptr(y), n, n_walks, T, H2, 1 if last_only else 0, _ptr(W_lin), _ptr(b_lin), D, _ptr(s), _ptr(X),
                                     _stream()), 'sgnn_lstm_tail_fwd')
        ctx.save_for_backward(s, W_lin)
        ctx.dims = (n, n_walks, T, H2, D, bool(last_only), b_lin is not None)
        return X

    @staticmethod
    def backward(ctx, dX):
        lib = _lib.load()
        s, W_lin = ctx.saved_tensors
        n, n_walks, T, H2, D, last_only, has_bias = ctx.dims
        dX = dX.contiguous()
        dy = torch.empty((n * n_walks, T, H2), dtype=torch.float32, device=dX.device)
        dW = torch.empty_like(W_lin) if ctx.needs_input_grad[1] else None
        db = torch.empty(D, dtype=torch.float32, device=dX.device) if (has_bias and ctx.needs_input_grad[2]) else None
        check(lib.sgnn_lstm_tail_bwd(_ptr(dX), _ptr(s), n, n_walks, T, H2, 1 if last_only else 0, _ptr(W_lin), D, _ptr(dy), _ptr(dW),
                                     _ptr(db), _stream()), 'sgnn_lstm_tail_bwd')
        return dy, dW, db, None, None


def bilstm_layer_fused(src, params, ids=None):
    """One bidirectional LSTM layer, dense products included (``_BiLSTMLayerFused``).  ``ids`` (B, T) int64: ``src`` is the
    embedding table and the layer's input is its rows ``ids`` (PAD = row 0); else ``src`` is the dense input (B, T, I)."""
    pre = getattr(ids, '_sgnn_sorted', None) if ids is not None else None
    return _BiLSTMLayerFused.apply(src if ids is not None else src.contiguous(), ids.contiguous() if ids is not None else None, pre,
                                   *[p.contiguous() for p in params])


def lstm_tail(y, weight, bias, n_walks, last_only=True):
    return _LSTMTail.apply(y.contiguous(), weight, bias, int(n_walks), bool(last_only))


def lstm_fused_supported(input_size, hidden_size):
    """The fused products need K % 8 == 0 on the input side (and the recurrence kernel's hidden sizes)."""
    return lstm_supported(input_size, hidden_size) and int(input_size) % 8 == 0 and int(hidden_size) % 2 == 0


def lstm_supported(input_size, hidden_size):
    return bool(_lib.load().sgnn_lstm_supported(int(hidden_size)))


def bilstm_layer(x, params):
    """``params``: (weight_ih, weight_hh, bias_ih, bias_hh) of the forward direction followed by the
    same four of the reverse direction."""
    return _BiLSTMLayer.apply(x.contiguous(), *[p.contiguous() for p in params])


class _AttnScores(torch.autograd.Function):
    """Additive-attention scores: forward on the matrix cores (sgnn_attn_scores_fwd); the backward
    recomputes tanh(qW + X U) with library GEMMs (plain dense contractions)."""

    @staticmethod
    def forward(ctx, X, U, qW, v, rows_per_batch, half_operands):
        lib = _lib.load()
        for t, nm in ((X, 'X'), (U, 'U'), (qW, 'qW'), (v, 'v')):
            _req(t, torch.float32, nm)
        R, H = X.shape
        out = torch.empty(R, dtype=torch.float32, device=X.device)
        if half_operands and H <= ATTN_F16_MAX_H and R >= ATTN_F16_MIN_ROWS:
            wsb = lib.sgnn_attn_scores_f16_workspace_bytes(H)
            ws = torch.empty(wsb // 2 + 1, dtype=torch.float16, device=X.device)
            check(lib.sgnn_attn_scores_fwd_f16(_ptr(X), _ptr(U), _ptr(qW), _ptr(v), R, H, rows_per_batch, _ptr(out),
                                               _ptr(ws), wsb, _stream()), 'sgnn_attn_scores_fwd_f16')
        else:
            # exact form: the contraction is a plain dense GEMM (library); + qW, tanh, x v and the sum over the columns
            # are one fused pass over it.  With half_operands the operands are rounded first (a batch of a few hundred
            # rows is four workgroups of the hand-written kernel: the library fills the chip better there).
            XU = X.half().float() @ U.half().float() if half_operands else X @ U      # (products of halves are exact in fp32)
            check(lib.sgnn_attn_scores_epilogue(_ptr(XU), _ptr(qW), _ptr(v), R, H, rows_per_batch, _ptr(out), _stream()),
                  'sgnn_attn_scores_epilogue')
        ctx.save_for_backward(X, U, qW, v)
        ctx.rpb = rows_per_batch
        return out

    @staticmethod
    def backward(ctx, g):
        X, U, qW, v = ctx.saved_tensors
        t = torch.tanh(torch.repeat_interleave(qW, ctx.rpb, dim=0) + X @ U)           # (R, H)
        d = g.unsqueeze(1) * v.view(1, -1) * (1 - t * t)
        gX = d @ U.t() if ctx.needs_input_grad[0] else None
        gU = X.t() @ d if ctx.needs_input_grad[1] else None
        gq = d.view(-1, ctx.rpb, d.shape[1]).sum(1) if ctx.needs_input_grad[2] else None
        gv = column_sum(t * g.unsqueeze(1)) if ctx.needs_input_grad[3] else None
        return gX, gU, gq, gv, None, None


ATTN_F16_MAX_H = 640
ATTN_F16_MIN_ROWS = 2048     # below: the library GEMM (on half-rounded operands) + the fused epilogue


def attn_scores(X, U, qW, v, rows_per_batch, half_operands=False):
    """score[r] = sum_j v_j tanh(qW[r // rows_per_batch, j] + (X U)[r, j]) for X (R, H).
    Exact form: library GEMM + the fused epilogue (sgnn_attn_scores_epilogue).  ``half_operands``: X and U are
    rounded to IEEE half and the whole thing is one hand-written kernel on the matrix cores
    (v_mfma_f32_32x32x16_f16, fp32 accumulate; for H <= 640 and at least a few thousand rows); the backward pass
    recomputes in fp32 either way."""
    return _AttnScores.apply(X.contiguous(), U.contiguous(), qW.contiguous(), v.contiguous().view(-1), int(rows_per_batch),
                             bool(half_operands))
