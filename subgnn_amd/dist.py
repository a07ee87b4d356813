"""Data parallelism over subgraph shards: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm).  The hot path shards embarrassingly -- every per-subgraph quantity
depends only on its subgraph plus read-only shared state (CSR graph, embedding table, shared
anchors), SURVEY.md section 8(e) -- so the data path has exactly one exchange step, the all-gather
of the per-component channel embeddings; data-parallel TRAINING adds the gradient all-reduce.

xGMI is point-to-point (7 links per GPU), so both collectives are issued once per step on large
flat buffers (one bucket), never per tensor.
"""
import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def flat_collectives(group=None):
    """Whether the group's backend has the flat-tensor collectives the fast path uses (all_gather_into_tensor,
    reduce_scatter_tensor, all_to_all_single).  RCCL ("nccl") has; gloo (the CPU tests, the one-GPU multi-rank check)
    takes the list forms.  Decided from the backend's NAME, once per call site -- a failing production collective is
    an error to surface, not a reason to switch to another collective sequence on one rank."""
    return is_initialized() and 'nccl' in str(dist.get_backend(group)).lower()


def shard_range(n_items, rank, world):
    """Contiguous block of items owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class EmulatedPeers:
    """ONE process standing in for rank ``rank`` of ``world`` (bench.py's ``strong_rank8`` object, tests): the rank computes its
    own share of every dealt stage exactly as a real rank does, and what the collectives would deliver from the peers comes
    from buffers recorded beforehand -- on the first pass the emulator computes every rank's share itself (draws are keyed by
    global numbers, anchors are redrawn identically every pass without ``resample_anchor_patches``), later passes copy the
    recorded result (the device-side cost of receiving it) and overwrite the rank's own block with what it has just computed.
    Values are therefore those of the real sharded run; no byte crosses a link (the caller prices the links)."""

    def __init__(self, rank, world):
        self.rank, self.world = int(rank), int(world)
        self.kept = {}            # key -> the full result of an exchange, as every rank would hold it
        self.provided = {}        # what only the other ranks could compute, supplied by the caller (all ranks' component ids)
        self.maxima = {}          # numel -> device tensor: the global maxima a MAX all-reduce of that many values would return
        self.received_bytes = {}  # key -> bytes per pass the rank would receive over the links

    def exchange(self, key, mine, lo, hi, make_full, dim=0):
        """The result of an all-gather / all-to-all whose ``lo:hi`` block (along ``dim``) this rank contributes."""
        full = self.kept.get(key)
        if full is None:
            full = self.kept[key] = make_full().contiguous()
            own = mine.numel() * mine.element_size()
            self.received_bytes[key] = full.numel() * full.element_size() - own
        out = full.clone()                                      # (what arrives from the peers: one copy of the buffer)
        out.narrow(dim, lo, hi - lo).copy_(mine)
        return out

    def reduce_max(self, t):
        m = self.maxima.get(t.numel())
        if m is None:
            raise RuntimeError('EmulatedPeers: no global maxima for a reduction of %d values' % t.numel())
        return torch.maximum(t, m.to(t.device, t.dtype).view_as(t))


class Shard:
    """This rank's contiguous block of a split's ``total`` subgraphs.  ``start`` is what makes a sharded
    pass reproduce the unsharded one bit for bit: every per-subgraph draw reads the tape item of the
    subgraph's GLOBAL number (hotpath.prepare_sparse passes it on as the samplers' ``item_base``).
    ``deal_shared``: also deal the shared (per-layer, subgraph-independent) work across ranks -- the
    sources of the position channel's multi-source BFS -- and exchange the results; pays when the
    shard is a slice of a fixed total (strong scaling), needs equal shard sizes."""

    def __init__(self, total, rank=None, world=None, deal_shared=False, collectives=True, group=None, emulator=None):
        self.world = world if world is not None else (dist.get_world_size() if is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if is_initialized() else 0)
        self.total = int(total)
        self.start, self.stop = shard_range(self.total, self.rank, self.world)
        self.emulator = emulator  # EmulatedPeers: the peers' shares come from recorded buffers instead of collectives
        self.collectives = collectives and self.world > 1 and emulator is None   # False: a single process replaying one rank's shard
        self.group = group        # communicator of the width reductions (its own when passes are pipelined: hotpath.PassPipeline)
        self.deal_shared = bool(deal_shared) and (self.collectives or emulator is not None) and self.total % self.world == 0

    @property
    def size(self):
        return self.stop - self.start

    def reduce_max(self, t):
        if self.emulator is not None:
            return self.emulator.reduce_max(t)
        return all_reduce_max_(t, self.group) if self.collectives else t


class _Pending:
    """Handle of a collective in flight: ``wait()`` returns its result."""

    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._finish()


def all_gather_rows(x, equal_rows=False, async_op=False):
    """(rows_r, H) per rank -> (sum rows, H) on every rank, ranks in order.  Row counts may
    differ between ranks (last shard shorter): rows are padded to the maximum for the collective
    and trimmed afterwards; ``equal_rows`` promises equal counts and skips that exchange (no host
    round trip).  ``async_op``: returns a handle whose ``wait()`` gives the result -- the collective
    runs on RCCL's stream while the caller keeps computing (the benchmark overlaps it with backward)."""
    if not is_initialized():
        return _Pending(None, lambda: x) if async_op else x
    world = dist.get_world_size()
    if equal_rows:
        sizes = [x.shape[0]] * world
    else:
        n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n)
        sizes = [int(s.item()) for s in got]
    m = max(sizes)
    if x.shape[0] < m:
        x = torch.cat([x, x.new_zeros((m - x.shape[0],) + tuple(x.shape[1:]))], 0)
    x = x.contiguous()
    out = torch.empty((world * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    parts = None
    if flat_collectives():
        work = dist.all_gather_into_tensor(out, x, async_op=True)
    else:                                                     # backends without the flat form
        parts = [torch.empty_like(x) for _ in range(world)]
        work = dist.all_gather(parts, x, async_op=True)

    def finish():
        res = out if parts is None else torch.cat(parts, 0)
        if all(s == m for s in sizes):
            return res
        return torch.cat([res[r * m:r * m + sizes[r]] for r in range(world)], 0)
    pending = _Pending(work, finish)
    return pending if async_op else pending.wait()


def all_reduce_gradients(params, average=True, big_bytes=16 << 20):
    """All-reduce of the gradients of ``params`` (every rank passes the same parameters in the same
    order).  The bucket layout is a function of the PARAMETERS, not of which gradients happen to exist
    on this rank: a parameter whose gradient is None here (a layer this rank's shard never exercised --
    e.g. the P-internal read-out weight when all of the shard's subgraphs have one component) enters as
    zeros and receives the reduced gradient, so ranks can never disagree on the collective's size.
    Tensors of at least ``big_bytes`` (the dense embedding-table gradient: 256 MB at N = 1M, D = 64) are
    reduced in place, each as its own collective; the small ones travel together in one flat bucket.
    All collectives are issued before any is waited for."""
    if not is_initialized():
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    world = dist.get_world_size()
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    is_big = [p.numel() * p.element_size() >= big_bytes and p.grad.is_contiguous() for p in params]
    big = [p.grad for p, b in zip(params, is_big) if b]
    small = [p.grad for p, b in zip(params, is_big) if not b]
    works = [dist.all_reduce(g, async_op=True) for g in big]
    flat = None
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        works.append(dist.all_reduce(flat, async_op=True))
    for w in works:
        w.wait()
    if average:
        for g in big:
            g /= world
    if flat is not None:
        if average:
            flat /= world
        off = 0
        for g in small:
            k = g.numel()
            g.copy_(flat[off:off + k].view_as(g))
            off += k


# ---------------------------------------------------------------------------------------------
# the data path's exchange step as a differentiable operator
# ---------------------------------------------------------------------------------------------

class _GatherRowsReplicated(torch.autograd.Function):
    """All-gather of row blocks whose consumer is REPLICATED: every rank runs the same computation on
    the gathered matrix and arrives at the same scalar loss.  The gradient of that loss with respect to
    rank r's rows is then simply rows r of the gradient every rank holds -- no reduction (a reduction
    would count the loss ``world`` times)."""

    @staticmethod
    def forward(ctx, x):
        ctx.rows = x.shape[0]
        return all_gather_rows(x.contiguous(), equal_rows=True)

    @staticmethod
    def backward(ctx, g):
        r = dist.get_rank() if is_initialized() else 0
        return g[r * ctx.rows:(r + 1) * ctx.rows]


def gather_rows_replicated(x):
    """(rows, H) per rank (equal row counts) -> (world * rows, H) on every rank, differentiable; see
    _GatherRowsReplicated for the contract on the consumer."""
    if not is_initialized() or dist.get_world_size() == 1:
        return x
    return _GatherRowsReplicated.apply(x)


def all_to_all_row_blocks(x):
    """x (world * rows, H): block b goes to rank b.  Returns (world * rows, H) whose block b came from
    rank b.  (Each rank computed some columns of a matrix for ALL ranks' rows; afterwards each rank
    holds every rank's columns for ITS rows.)"""
    if not is_initialized() or dist.get_world_size() == 1:
        return x
    x = x.contiguous()
    out = torch.empty_like(x)
    if flat_collectives():
        dist.all_to_all_single(out, x)
    else:                                                   # backends without all-to-all: gather everything, keep ours
        world, r = dist.get_world_size(), dist.get_rank()
        rows = x.shape[0] // world
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        out = torch.cat([p[r * rows:(r + 1) * rows] for p in parts], 0)
    return out


def all_reduce_max_(t, group=None):
    """In-place MAX over ranks of a small device tensor (global padded widths, border sizes)."""
    if is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


# ---------------------------------------------------------------------------------------------
# embedding-table gradient: owner-computes update (reduce-scatter -> Adam on 1/world of the rows ->
# all-gather of the updated rows, overlapped with the next pass's integer stages)
# ---------------------------------------------------------------------------------------------

class ShardedTableAdam:
    """Adam for ONE large parameter (the (N+1, D) embedding table) under data parallelism, without the
    dense all-reduce + replicated update: the gradient is reduce-scattered (each rank receives the sum
    of one contiguous 1/world slice), the owner updates its slice of the parameter (its moments exist
    only there: optimizer state / world), and the updated slices are all-gathered back -- asynchronously:
    ``wait()`` is called right before the next reader of the table, so the gather travels on RCCL's stream
    while the next pass's sampling and similarity stages (which never read the table) compute.  Same
    bytes on the wire as a ring all-reduce, half of them hidden, and the 2 x N x D moment update drops
    to 1/world.  The update rule is torch.optim.Adam's (bias-corrected, eps outside the square root).

    Why not a sparse (row id, row) exchange: on the benchmark shard ~40 % of the table's
    rows are touched per rank and ~95 % by the union of 8 ranks -- the gradient is dense; see
    ``sparse_row_all_reduce`` for the regime where it is not."""

    def __init__(self, param, lr, betas=(0.9, 0.999), eps=1e-8, average=False, emulate=None):
        """``emulate`` = (rank, world): one process standing in for that rank (EmulatedPeers): the slices, the moments and the
        update are the rank's; the reduce-scatter delivers the rank's own gradient slice (the peers' contributions are not
        available in one process -- the caller prices the links) and nothing is gathered."""
        self.p, self.lr, self.b1, self.b2, self.eps, self.average = param, lr, betas[0], betas[1], eps, average
        self.world = dist.get_world_size() if is_initialized() else 1
        self.rank = dist.get_rank() if is_initialized() else 0
        self.emulated = emulate is not None
        if self.emulated:
            self.rank, self.world = int(emulate[0]), int(emulate[1])
        n = param.numel()
        self.chunk = n // self.world                      # equal slices; the < world trailing elements are replicated
        self.tail = n - self.chunk * self.world
        self.lo, self.hi = self.rank * self.chunk, (self.rank + 1) * self.chunk
        dev = param.device
        self.m = torch.zeros(self.chunk + self.tail, dtype=torch.float32, device=dev)
        self.v = torch.zeros_like(self.m)
        self.gslice = torch.empty(self.chunk, dtype=torch.float32, device=dev)
        self.t = 0
        self._pending = None

    def reduce_grad(self):
        """Start of the update: reduce-scatter of param.grad.  Returns the squared norm of this rank's
        reduced slice (+ the replicated tail on rank 0 only), for a global clip norm."""
        g = self.p.grad.reshape(-1)
        body = g[:self.chunk * self.world]
        if self.emulated:
            self.gslice.copy_(body[self.lo:self.hi])
        elif self.world > 1:
            if flat_collectives():
                dist.reduce_scatter_tensor(self.gslice, body)
            else:                                           # backends without reduce-scatter
                full = body.clone()
                dist.all_reduce(full)
                self.gslice.copy_(full[self.lo:self.hi])
            if self.tail:
                dist.all_reduce(g[self.chunk * self.world:])
            if self.average:
                self.gslice /= self.world
                if self.tail:
                    g[self.chunk * self.world:] /= self.world
        else:
            self.gslice.copy_(body)
        sq = (self.gslice * self.gslice).sum()
        if self.tail and self.rank == 0:
            sq = sq + (g[self.chunk * self.world:] ** 2).sum()
        return sq

    def step(self, grad_scale=None):
        """Adam on the owned slice (and the replicated tail), then the asynchronous all-gather."""
        self.t += 1
        flat = self.p.data.view(-1)
        g = self.p.grad.reshape(-1)
        own = flat[self.lo:self.hi]
        if self.p.is_cuda and not self.tail and self.chunk > 0 and self.p.dtype == torch.float32 \
                and (own.data_ptr() | self.gslice.data_ptr() | self.m.data_ptr() | self.v.data_ptr()) % 16 == 0:
            # the owned slice in ONE pass (sgnn_adam_step: the rule below, the clip coefficient read from the device) instead of
            # ten element-wise launches over a slice of millions of elements
            from . import ops
            scale = grad_scale.reshape(1).float() if torch.is_tensor(grad_scale) else (
                None if grad_scale is None else torch.full((1,), float(grad_scale), dtype=torch.float32, device=self.p.device))
            ops.adam_step(own, self.gslice, self.m, self.v, self.lr, (self.b1, self.b2), self.eps, self.t, grad_scale=scale)
            self._gather_owned(flat)
            return
        gs = self.gslice if not self.tail else torch.cat([self.gslice, g[self.chunk * self.world:]])
        if grad_scale is not None:
            gs = gs * grad_scale
        self.m.mul_(self.b1).add_(gs, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(gs, gs, value=1 - self.b2)
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        upd = (self.m / bc1) / ((self.v / bc2).sqrt_().add_(self.eps))
        flat[self.lo:self.hi].add_(upd[:self.chunk], alpha=-self.lr)
        if self.tail:
            flat[self.chunk * self.world:].add_(upd[self.chunk:], alpha=-self.lr)
        self._gather_owned(flat)

    def _gather_owned(self, flat):
        """The asynchronous all-gather of the updated slices (``wait`` lands it)."""
        if self.world > 1 and not self.emulated:
            body = flat[:self.chunk * self.world]
            mine = flat[self.lo:self.hi].clone()            # the collective must not read and write the same bytes
            if flat_collectives():
                self._pending = dist.all_gather_into_tensor(body, mine, async_op=True)
            else:
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                work = dist.all_gather(parts, mine, async_op=True)   # asynchronous like the flat form; wait() lands the rows

                class _Land:
                    def wait(_self):
                        work.wait()
                        body.copy_(torch.cat(parts))
                self._pending = _Land()

    def wait(self):
        """Call before anything reads the parameter."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None


def sparse_row_all_reduce(grad, max_fraction=0.25):
    """Sum over ranks of a (rows, D) gradient that is mostly zero rows (a batch of a few dozen subgraphs
    touches a few thousand rows of a table of millions): all-gather of (row id, row) pairs of the
    touched rows instead of an all-reduce of the dense table.  Falls back to the dense
    all-reduce when more than ``max_fraction`` of the rows are touched on some rank (then the pairs would
    outweigh the table).  In place; returns the number of rows exchanged per rank (0 = dense path)."""
    if not is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    rows, D = grad.shape
    touched = (grad != 0).any(dim=1)
    n = touched.sum().view(1)
    nmax = n.clone()
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
    k = int(nmax.item())
    if k == 0:
        return 0
    if k > max_fraction * rows:
        dist.all_reduce(grad)
        return 0
    ids = torch.full((k,), -1, dtype=torch.int64, device=grad.device)
    mine = touched.nonzero().view(-1)
    ids[:mine.numel()] = mine
    vals = torch.zeros((k, D), dtype=grad.dtype, device=grad.device)
    vals[:mine.numel()] = grad[mine]
    all_ids = [torch.empty_like(ids) for _ in range(world)]
    all_vals = [torch.empty_like(vals) for _ in range(world)]
    w1 = dist.all_gather(all_ids, ids, async_op=True)
    w2 = dist.all_gather(all_vals, vals, async_op=True)
    w1.wait()
    w2.wait()
    grad.zero_()
    # ranks added in rank order on every rank: the sum is the same bits everywhere
    for i, v in zip(all_ids, all_vals):
        keep = i >= 0
        grad.index_add_(0, i[keep], v[keep])
    return k
