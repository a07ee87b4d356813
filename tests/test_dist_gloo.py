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
    # strong scaling's dealing of shared per-pass work (hotpath._deal_rows): 7 rows of a result every rank needs, computed
    # as ceil(7 / 2) = 4 + 3 rows keyed by GLOBAL row numbers, gathered in rank order -> what one rank would compute alone
    from subgnn_amd import hotpath
    shard = D.Shard(8, deal_shared=True)
    table = torch.arange(7 * 5, dtype=torch.int64).view(7, 5) * 3 + 1
    dealt = hotpath._deal_rows(shard, 7, lambda lo, hi: table[lo:hi].clone(), (5,), torch.int64, 'cpu')
    ok_gather = ok_gather and shard.deal_shared and torch.equal(dealt, table)
    one = hotpath._deal_rows(shard, 1, lambda lo, hi: table[lo:hi].clone(), (5,), torch.int64, 'cpu')      # rank 1's share is empty
    ok_gather = ok_gather and torch.equal(one, table[:1])
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


def _worker_dp_ops(rank, world, port, q):
    """The data-parallel operators of the sharded pass, 2 ranks over gloo, each checked against the
    single-process computation on the full data."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from subgnn_amd import dist as D
    torch.manual_seed(0)
    ok = {}
    # --- replicated head on gathered rows: gradients equal the single-process ones ------------------
    B, H = 6, 5
    X = torch.randn(world * B, H)
    Wc = torch.randn(H, H)                       # a "channel" parameter (sharded rows flow through it)
    Wh = torch.randn(H, 3)                       # the head (replicated)
    y = torch.randint(0, 3, (world * B,))

    def loss_of(xrows, wc, wh, gather):
        e = torch.tanh(xrows @ wc)
        e = D.gather_rows_replicated(e) if gather else e
        return torch.nn.functional.cross_entropy(e @ wh, y)
    wc_f, wh_f = Wc.clone().requires_grad_(True), Wh.clone().requires_grad_(True)
    loss_of(X, wc_f, wh_f, False).backward()
    wc, wh = Wc.clone().requires_grad_(True), Wh.clone().requires_grad_(True)
    a, b = D.shard_range(world * B, rank, world)
    loss_of(X[a:b], wc, wh, True).backward()
    D.all_reduce_gradients([wc], average=False)                      # channel params: SUM of the shares
    ok['head'] = torch.allclose(wh.grad, wh_f.grad, atol=1e-6)       # head: complete on every rank, no reduction
    ok['channel'] = torch.allclose(wc.grad, wc_f.grad, atol=1e-6)
    # --- all-to-all of row blocks ---------------------------------------------------------------------
    rows = 3
    part = torch.arange(world * rows * 2, dtype=torch.float32).view(world * rows, 2) + 1000 * rank
    got = D.all_to_all_row_blocks(part)
    want = torch.cat([torch.arange(world * rows * 2, dtype=torch.float32).view(world * rows, 2)[rank * rows:(rank + 1) * rows]
                      + 1000 * j for j in range(world)], 0)
    ok['a2a'] = torch.equal(got, want)
    # --- sharded table Adam == torch Adam on the summed gradient, over several steps ---------------------
    N, Dm = 13, 3                                                    # 39 elements: a replicated tail of 1
    table = torch.nn.Parameter(torch.randn(N, Dm))
    ref = torch.nn.Parameter(table.detach().clone())
    opt_ref = torch.optim.Adam([ref], lr=0.05)
    sh = D.ShardedTableAdam(table, 0.05)
    gen = torch.Generator().manual_seed(5)
    for it in range(4):
        g_all = [torch.randn(N, Dm, generator=gen) * (torch.rand(N, 1, generator=gen) < 0.6) for _ in range(world)]
        table.grad = g_all[rank].clone()
        ref.grad = sum(g_all)
        sq = sh.reduce_grad()
        dist.all_reduce(sq)
        ok['norm%d' % it] = torch.allclose(sq, (ref.grad ** 2).sum(), rtol=1e-5)
        sh.step()
        sh.wait()
        opt_ref.step()
        ok['adam%d' % it] = torch.allclose(table.detach(), ref.detach(), atol=1e-6)
    # --- the sharded-head form of data parallelism (bench.py --head sharded): every rank's loss is the mean over its
    #     own rows; averaged small gradients + ShardedTableAdam(average=True) == one process on the global batch
    gen = torch.Generator().manual_seed(9)
    Xall, yall = torch.randn(world * 6, 5, generator=gen), torch.randint(0, 3, (world * 6,), generator=gen)
    emb0, W0 = torch.randn(7, 5, generator=gen), torch.randn(3, 5, generator=gen)
    ids_all = torch.randint(0, 7, (world * 6,), generator=gen)

    def loss_of(emb, W, sl):
        return torch.nn.functional.cross_entropy((Xall[sl] + emb[ids_all[sl]]) @ W.t(), yall[sl])
    emb_r, W_r = torch.nn.Parameter(emb0.clone()), torch.nn.Parameter(W0.clone())
    opt_r = torch.optim.Adam([emb_r, W_r], lr=0.05)
    emb_d, W_d = torch.nn.Parameter(emb0.clone()), torch.nn.Parameter(W0.clone())
    opt_d = torch.optim.Adam([W_d], lr=0.05)
    sh2 = D.ShardedTableAdam(emb_d, 0.05, average=True)
    mine_sl = slice(rank * 6, (rank + 1) * 6)
    for it in range(3):
        opt_r.zero_grad()
        loss_of(emb_r, W_r, slice(0, world * 6)).backward()
        opt_r.step()
        emb_d.grad = W_d.grad = None
        loss_of(emb_d, W_d, mine_sl).backward()
        D.all_reduce_gradients([W_d], average=True)
        sh2.reduce_grad()
        opt_d.step()
        sh2.step()
        sh2.wait()
        ok['dp_mean_of_means%d' % it] = torch.allclose(W_d.detach(), W_r.detach(), atol=1e-6) and \
            torch.allclose(emb_d.detach(), emb_r.detach(), atol=1e-6)
    # --- sparse (row id, row) exchange ----------------------------------------------------------------
    g = torch.zeros(200, 4)
    mine = torch.tensor([3, 50, 77]) + rank
    g[mine] = torch.randn(3, 4, generator=torch.Generator().manual_seed(rank))
    dense = g.clone()
    dist.all_reduce(dense)
    k = D.sparse_row_all_reduce(g)
    ok['sparse'] = k == 3 and torch.allclose(g, dense)
    g2 = torch.ones(8, 2) * (rank + 1)                               # dense gradient: falls back to the all-reduce
    ok['sparse_dense_fallback'] = D.sparse_row_all_reduce(g2) == 0 and torch.equal(g2, torch.full((8, 2), 3.0))
    # --- Shard bookkeeping -----------------------------------------------------------------------------
    shd = D.Shard(10, deal_shared=True)
    ok['shard'] = (shd.start, shd.stop) == D.shard_range(10, rank, world) and shd.deal_shared and \
        int(shd.reduce_max(torch.tensor([rank + 4]))[0]) == world + 3
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_dp_operators():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dp_ops, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok.values()), (rank, ok)
