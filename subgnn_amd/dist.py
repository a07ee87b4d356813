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


def shard_range(n_items, rank, world):
    """Contiguous block of items owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


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
    try:
        work = dist.all_gather_into_tensor(out, x, async_op=True)
    except (RuntimeError, NotImplementedError):               # backends without the flat form
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
