"""CPU, world_size 2, gloo: the N>1 path of the benchmark (shard assignment, all-gather of the
per-component channel embeddings, flat gradient all-reduce)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from subgnn_amd import dist as D
    n = 11
    a, b = D.shard_range(n, rank, world)
    full = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    got = D.all_gather_rows(full[a:b].clone())                # uneven shards: 6 + 5 rows
    ok_gather = torch.equal(got, full)
    # the benchmark's form: equal shards, issued asynchronously, waited for after other work
    eq = torch.arange(8, dtype=torch.float32).view(4, 2) + 100 * rank
    pending = D.all_gather_rows(eq, equal_rows=True, async_op=True)
    busy = (eq @ eq.t()).sum()                                 # "backward" while the gather travels
    ge = pending.wait()
    ok_gather = ok_gather and torch.equal(ge, torch.cat([eq - 100 * rank + 100 * r for r in range(world)], 0)) \
        and bool(torch.isfinite(busy))
    w = torch.nn.Linear(3, 2)
    torch.manual_seed(0)
    with torch.no_grad():
        w.weight.fill_(0.5)
        w.bias.zero_()
    loss = w(full[a:b]).sum()
    loss.backward()
    D.all_reduce_gradients(list(w.parameters()), big_bytes=20)     # weight (24 B) in place, bias in the bucket
    ref = torch.nn.Linear(3, 2)
    with torch.no_grad():
        ref.weight.fill_(0.5)
        ref.bias.zero_()
    parts = [ref(full[slice(*D.shard_range(n, r, world))]).sum() for r in range(world)]
    (sum(parts) / world).backward()
    ok_grad = torch.allclose(w.weight.grad, ref.weight.grad) and torch.allclose(w.bias.grad, ref.bias.grad)
    # a parameter only rank 0's shard exercises: rank 1 has no gradient for it, yet both ranks must issue
    # the same collective (the bucket layout follows the parameters, not the existing gradients)
    extra = torch.nn.Parameter(torch.ones(4))
    w.zero_grad(set_to_none=True)
    loss = w(full[a:b]).sum() + ((extra * torch.arange(4.0)).sum() if rank == 0 else 0.0)
    loss.backward()
    assert (extra.grad is None) == (rank != 0)
    D.all_reduce_gradients(list(w.parameters()) + [extra], big_bytes=1 << 20)
    ok_grad = ok_grad and torch.allclose(extra.grad, torch.arange(4.0) / world) and \
        torch.allclose(w.weight.grad, ref.weight.grad)
    q.put((rank, (a, b), bool(ok_gather), bool(ok_grad)))
    dist.destroy_process_group()


def test_two_rank_gather_and_allreduce():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 6) and res[1][1] == (6, 11)
    assert all(r[2] and r[3] for r in res)


def test_shard_range_partitions():
    from subgnn_amd.dist import shard_range
    for n in (0, 1, 7, 50000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
