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


def all_gather_rows(x):
    """(rows_r, H) per rank -> (sum rows, H) on every rank, ranks in order.  Row counts may
    differ between ranks (last shard shorter): rows are padded to the maximum for the collective
    and trimmed afterwards."""
    if not is_initialized():
        return x
    world = dist.get_world_size()
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    if x.shape[0] < m:
        x = torch.cat([x, x.new_zeros((m - x.shape[0],) + tuple(x.shape[1:]))], 0)
    out = torch.empty((world * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    try:
        dist.all_gather_into_tensor(out, x.contiguous())
    except (RuntimeError, NotImplementedError):               # backends without the flat form
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous())
        out = torch.cat(parts, 0)
    if all(s == m for s in sizes):
        return out
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], 0)


def all_reduce_gradients(params, average=True):
    """One flat all-reduce over every existing gradient (single bucket)."""
    if not is_initialized():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        k = g.numel()
        g.copy_(flat[off:off + k].view_as(g))
        off += k
