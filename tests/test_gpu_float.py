"""-m gpu parity tests of the float-half HIP kernels (through the C ABI) against the oracle
(dense torch-CPU restatement pinned by reference goldens g10/g11).  Tolerance: 1e-4 relative
(BASELINE.json north_star), stated in helpers.REL_TOL."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import float_half as FH
from helpers import T, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ops():
    from subgnn_amd import ops
    return ops


@pytest.fixture(params=['sorted', 'atomics'])
def det(request):
    """Both backward forms of the ops that add into the embedding table: sorted segmented sums / tile partials added
    in a fixed order (the default, bit-reproducible) and float atomics."""
    from subgnn_amd import ops
    ops.DETERMINISTIC = request.param == 'sorted'
    yield request.param
    ops.DETERMINISTIC = True


@pytest.mark.parametrize('agg', ['sum', 'max'])
@pytest.mark.parametrize('D', [8, 32, 64, 128])
def test_cc_embed(agg, D, det):
    ops = _ops()
    g = torch.Generator().manual_seed(D)
    N, S, C, L = 500, 40, 3, 9
    E = torch.randn(N + 1, D, generator=g)
    E[0] = 0
    cc = torch.randint(1, N + 1, (S, C, L), generator=g)
    lens = torch.randint(0, L + 1, (S, C), generator=g)
    lens[0, 0] = L
    cc[torch.arange(L).view(1, 1, L) >= lens.unsqueeze(-1)] = 0
    Ec = E.double().requires_grad_(True)                          # exact reference: compared element by element
    ref = FH.cc_embeddings(Ec, cc, agg)
    gout = torch.randn(ref.shape, generator=g)
    (ref * gout.double()).sum().backward()
    Eg = E.to(DEV).requires_grad_(True)
    sets = ops.Ragged.from_padded(cc.view(S * C, L).to(DEV))
    out = ops.cc_embed(Eg, sets, agg, padded_len=L).view(S, C, D)
    (out * gout.to(DEV)).sum().backward()
    assert_close(out, ref, 'cc_embed fwd', norm_tol=1e-6)
    assert_close(Eg.grad, Ec.grad, 'cc_embed bwd')
    assert float(Eg.grad[0].abs().max()) == 0.0


def _g10(g, tag, side):
    t = 'g10_%s_%s_' % (tag, side)
    d = {k: T(g[t + k]) for k in ('W', 'b', 'wp', 'bp', 'sims', 'cc_embeds', 'patches', 'mask', 'anchor_embeds',
                                  'gout_cc', 'gout_pos')}
    d['idx'] = [int(i) for i in g[t + 'sim_index']] if tag == 'S' else None
    return t, d


@pytest.mark.parametrize('tag', ['N', 'P', 'S'])
@pytest.mark.parametrize('side', ['in', 'out'])
def test_mpn_dense_matches_reference_golden(tiny, tag, side, det):
    """SRC_DENSE = the reference's own SG_MPN inputs (materialised anchor_embeds + mask)."""
    ops = _ops()
    t, d = _g10(tiny, tag, side)
    B, C, A, D = d['anchor_embeds'].shape
    R = B * C
    ae = d['anchor_embeds'].to(DEV).requires_grad_(True)
    wp = d['wp'].to(DEV).requires_grad_(True)
    bp = d['bp'].to(DEV).requires_grad_(True)
    ids = d['patches'][..., 0].contiguous().view(R, A).to(DEV)
    em = d['mask'][..., 0].contiguous().view(R, A).to(torch.uint8).to(DEV)
    sim_col = torch.tensor(d['idx'], dtype=torch.int64, device=DEV) if d['idx'] is not None else None
    agg, z = ops.mpn(ae.view(R, A, D), wp, bp, d['sims'].to(DEV), src=ops.SRC_DENSE, R=R, A=A, ids=ids,
                     edge_mask=em, sim_col=sim_col)
    W, b = d['W'].to(DEV).requires_grad_(True), d['b'].to(DEV).requires_grad_(True)
    cc = d['cc_embeds'].to(DEV).requires_grad_(True)
    out = torch.relu(torch.nn.functional.linear(torch.cat([cc.view(R, D), agg], -1), W, b)).view(B, C, D)
    pos = torch.relu(z).view(B, C, A)
    ((out * d['gout_cc'].to(DEV)).sum() + (pos * d['gout_pos'].to(DEV)).sum()).backward()
    g = tiny
    assert_close(out, g[t + 'out_cc'], 'out_cc')
    assert_close(pos, g[t + 'out_pos'], 'out_pos')
    assert_close(cc.grad, g[t + 'grad_cc_embeds'], 'grad cc')
    assert_close(ae.grad, g[t + 'grad_anchor_embeds'], 'grad anchor embeds')
    assert_close(W.grad, g[t + 'grad_W'], 'grad W')
    assert_close(wp.grad, g[t + 'grad_wp'], 'grad wp')
    assert_close(bp.grad, g[t + 'grad_bp'], 'grad bp')


def _rand_case(seed, R, A, D, N, C):
    g = torch.Generator().manual_seed(seed)
    E = torch.randn(N + 1, D, generator=g)
    E[0] = 0
    ids = torch.randint(0, N + 1, (R, A), generator=g)
    ids[torch.rand(R, A, generator=g) < 0.15] = 0
    row_mask = (torch.rand(R, generator=g) > 0.2)
    sims = torch.rand(R, N, generator=g) * 4
    wp = torch.randn(1, D, generator=g) * 0.3
    bp = torch.randn(1, generator=g) * 0.1
    gagg = torch.randn(R, D, generator=g)
    gz = torch.randn(R, A, generator=g)
    return E, ids, row_mask, sims, wp, bp, gagg, gz


def _ref_mpn(x_rows, edge, w, wp, bp):
    """agg, z from per-edge rows (R,A,D), edge mask (R,A), weights (R,A)."""
    m = (w * edge).unsqueeze(-1) * x_rows
    agg = m.sum(1)
    z = (m @ wp.view(-1, 1)).squeeze(-1) + bp
    return agg, z


@pytest.mark.parametrize('A', [11, 69])            # 69: batch-sized calls split their anchors over grid.y
@pytest.mark.parametrize('D', [8, 32, 64, 128])
def test_mpn_gather_random(D, A, det):
    """SRC_GATHER (anchor rows gathered from the embedding table by id, NP slab indexed by id-1)."""
    ops = _ops()
    R, N, C = 96, 300, 4
    E, ids, row_mask, sims, wp, bp, gagg, gz = _rand_case(D, R, A, D, N, C)
    Ec, wpc, bpc = E.clone().requires_grad_(True), wp.clone().requires_grad_(True), bp.clone().requires_grad_(True)
    edge = ((ids != 0) & row_mask.unsqueeze(-1)).float()
    w = torch.gather(sims, 1, (ids - 1).clamp(min=0))
    agg_r, z_r = _ref_mpn(torch.nn.functional.embedding(ids, Ec, padding_idx=0), edge, w, wpc, bpc)
    ((agg_r * gagg).sum() + (z_r * gz).sum()).backward()
    Eg, wpg, bpg = E.to(DEV).requires_grad_(True), wp.to(DEV).requires_grad_(True), bp.to(DEV).requires_grad_(True)
    agg, z = ops.mpn(Eg, wpg, bpg, sims.to(DEV), src=ops.SRC_GATHER, R=R, A=A, ids=ids.to(DEV),
                     row_mask=row_mask.to(torch.uint8).to(DEV))
    ((agg * gagg.to(DEV)).sum() + (z * gz.to(DEV)).sum()).backward()
    assert_close(agg, agg_r, 'agg')
    assert_close(z, z_r, 'z')
    assert_close(Eg.grad, Ec.grad, 'grad E')
    assert_close(wpg.grad, wpc.grad, 'grad wp')
    assert_close(bpg.grad, bpc.grad, 'grad bp')
    assert float(Eg.grad[0].abs().max()) == 0.0


@pytest.mark.parametrize('D', [32, 128])
def test_layer_bodies_together_equal_one_after_the_other(D, det):
    """The batch-sized step's per-layer form -- the bodies' forward launches queued and sent as one (ops.mpn(lazy=True) +
    ops.flush_lazy_mpn -> sgnn_mpn_fwd_many), their update layers in one launch each way (ops.update_layers ->
    sgnn_update_fwd_many / _bwd_many) -- against the same bodies run one after the other: a GATHER body with 70 anchors (its
    anchors split into chunks), a GATHER body with 9, a SHARED body, a body whose aggregate is all zero, one whose updated
    embeddings nobody reads (no gradient arrives: its backward is skipped).  Outputs and every gradient bit for bit."""
    ops = _ops()
    R, N, C = 96, 300, 4
    g = torch.Generator().manual_seed(D)
    cases = []
    for A in (70, 9):
        E, ids, row_mask, sims, wp, bp, _, _ = _rand_case(D + A, R, A, D, N, C)
        cases.append(dict(src=ops.SRC_GATHER, x=E, ids=ids, sims=sims, A=A, wp=wp, bp=bp, row_mask=row_mask, sim_col=None))
    A = 40
    X = torch.randn(A, D, generator=g)
    cases.append(dict(src=ops.SRC_SHARED, x=X, ids=None, sims=torch.rand(R, A + 5, generator=g), A=A, wp=torch.randn(1, D, generator=g) * 0.3,
                      bp=torch.randn(1, generator=g) * 0.1, row_mask=torch.rand(R, generator=g) > 0.2,
                      sim_col=torch.randperm(A + 5, generator=g)[:A]))
    cc = [torch.randn(R, D, generator=g) for _ in range(5)]
    Ws = [torch.randn(D, 2 * D, generator=g) / (2 * D) ** 0.5 for _ in range(5)]
    bs = [torch.randn(D, generator=g) * 0.1 for _ in range(5)]
    go = [torch.randn(R, D, generator=g) for _ in range(5)]
    gz = [torch.randn(R, c['A'], generator=g) for c in cases]

    def run(together):
        leaves = {}

        def leaf(name, t):
            leaves[name] = t.to(DEV).clone().requires_grad_(True)
            return leaves[name]
        pend, zs = [], []
        for k, c in enumerate(cases):
            agg, z = ops.mpn(leaf('x%d' % k, c['x']), leaf('wp%d' % k, c['wp']), leaf('bp%d' % k, c['bp']), c['sims'].to(DEV), src=c['src'],
                             R=R, A=c['A'], ids=c['ids'].to(DEV) if c['ids'] is not None else None,
                             row_mask=c['row_mask'].to(torch.uint8).to(DEV), sim_col=c['sim_col'].to(DEV) if c['sim_col'] is not None else None,
                             keep_chunks=True, relu_z=True, lazy=together)
            zs.append(z)
            pend.append(ops.PendingUpdate(leaf('cc%d' % k, cc[k]), agg, leaf('W%d' % k, Ws[k]), leaf('b%d' % k, bs[k]), (R // C, C)))
        zero = torch.zeros(R, D, device=DEV)
        for k in (3, 4):                                           # all-zero aggregate; a body nobody reads
            pend.append(ops.PendingUpdate(leaf('cc%d' % k, cc[k]), zero, leaf('W%d' % k, Ws[k]), leaf('b%d' % k, bs[k]), (R // C, C)))
        if together:
            assert ops.lazy_mpn_pending() == 3
            outs = ops.update_layers(pend)
            assert ops.lazy_mpn_pending() == 0
        else:
            outs = [ops.update_layer(p.x, p.aggr, p.weight, p.bias) for p in pend]
        loss = sum((o * go[k].to(DEV)).sum() for k, o in enumerate(outs) if k != 4) + sum((z * gz[k].to(DEV)).sum() for k, z in enumerate(zs))
        loss.backward()
        return [o.detach() for o in outs] + [z.detach() for z in zs], {n: t.grad for n, t in leaves.items()}
    a_out, a_g = run(True)
    b_out, b_g = run(False)
    for i, (u, v) in enumerate(zip(a_out, b_out)):
        assert torch.equal(u, v), i
    assert a_g['W4'] is None and a_g['cc4'] is None and b_g['W4'] is None
    for n in a_g:
        if a_g[n] is None or b_g[n] is None:
            assert a_g[n] is None and b_g[n] is None, n
        elif det == 'sorted':
            assert torch.equal(a_g[n], b_g[n]), n
        else:
            assert_close(a_g[n], b_g[n], 'grad ' + n, norm_tol=1e-5)


@pytest.mark.parametrize('A', [9, 70])
def test_mpn_dense_random(A, det):
    """SRC_DENSE (the reference-shaped (R, A, D) anchor tensor) on random inputs, with few and with many
    anchors per row (the latter takes the anchor-split launch at batch size)."""
    ops = _ops()
    R, D, N, C = 90, 32, 250, 3
    E, ids, row_mask, sims, wp, bp, gagg, gz = _rand_case(A, R, A, D, N, C)
    edge_b = (ids != 0) & row_mask.unsqueeze(-1)
    w = torch.gather(sims, 1, (ids - 1).clamp(min=0))
    X = torch.nn.functional.embedding(ids, E, padding_idx=0)                     # (R, A, D)
    Xc, wpc, bpc = X.clone().requires_grad_(True), wp.clone().requires_grad_(True), bp.clone().requires_grad_(True)
    agg_r, z_r = _ref_mpn(Xc, edge_b.float(), w, wpc, bpc)
    ((agg_r * gagg).sum() + (z_r * gz).sum()).backward()
    Xg, wpg, bpg = X.to(DEV).requires_grad_(True), wp.to(DEV).requires_grad_(True), bp.to(DEV).requires_grad_(True)
    agg, z = ops.mpn(Xg, wpg, bpg, sims.to(DEV), src=ops.SRC_DENSE, R=R, A=A, ids=ids.to(DEV),
                     edge_mask=edge_b.to(torch.uint8).to(DEV))
    ((agg * gagg.to(DEV)).sum() + (z * gz.to(DEV)).sum()).backward()
    assert_close(agg, agg_r, 'agg')
    assert_close(z, z_r, 'z')
    assert_close(Xg.grad, Xc.grad, 'grad X')
    assert_close(wpg.grad, wpc.grad, 'grad wp')
    assert_close(bpg.grad, bpc.grad, 'grad bp')


def test_mpn_gather_shared_ids_over_components(det):
    """P-internal: one id row per subgraph shared by its C components (id_div = C)."""
    ops = _ops()
    B, C, A, D, N = 10, 3, 7, 16, 100
    R = B * C
    E, _, row_mask, sims, wp, bp, gagg, gz = _rand_case(3, R, A, D, N, C)
    g = torch.Generator().manual_seed(4)
    ids_b = torch.randint(1, N + 1, (B, A), generator=g)
    ids = ids_b.unsqueeze(1).repeat(1, C, 1).view(R, A)
    edge = row_mask.unsqueeze(-1).float().expand(R, A)
    w = torch.gather(sims, 1, ids - 1)
    agg_r, z_r = _ref_mpn(E[ids], edge, w, wp, bp)
    agg, z = ops.mpn(E.to(DEV), wp.to(DEV), bp.to(DEV), sims.to(DEV), src=ops.SRC_GATHER, R=R, A=A, ids=ids_b.to(DEV),
                     id_div=C, row_mask=row_mask.to(torch.uint8).to(DEV))
    assert_close(agg, agg_r, 'agg')
    assert_close(z, z_r, 'z')


@pytest.mark.parametrize('D', [8, 64, 128])
@pytest.mark.parametrize('mode', ['sim_col', 'per_edge', 'by_id'])
@pytest.mark.parametrize('R,A', [(150, 13), (150, 123), (66000, 13)])
def test_mpn_shared_random(D, mode, R, A, det):
    """SRC_SHARED: P-border (ids -> column id-1) and structure (index list) anchors; a batch-sized
    row count (short row tiles x item chunks in the backward) and a shard-sized one (64-row tiles)."""
    ops = _ops()
    N, C = 200, 3
    E, _, row_mask, sims, wp, bp, gagg, gz = _rand_case(D + 1, R, A, D, N, C)
    g = torch.Generator().manual_seed(8)
    X = torch.randn(A, D, generator=g)
    ids = torch.randint(1, N + 1, (A,), generator=g)
    col = torch.randint(0, N, (A,), generator=g)
    # the reference in float64: over 66000 rows the float32 CPU sums are themselves off by more than 1e-4 of an element
    # that is small by cancellation -- the kernel is compared with the exact value, element by element
    Xc, wpc, bpc = (t.double().requires_grad_(True) for t in (X, wp, bp))
    if mode == 'sim_col':
        w = sims[:, col]
        kw = dict(sim_col=col.to(DEV))
        s_in = sims
    elif mode == 'by_id':
        w = sims[:, ids - 1]
        kw = dict(ids=ids.to(DEV))
        s_in = sims
    else:
        w = sims[:, :A].contiguous()
        kw = dict(sims_per_edge=True)
        s_in = w
    edge = row_mask.unsqueeze(-1).float().expand(R, A)
    agg_r, z_r = _ref_mpn(Xc.unsqueeze(0).expand(R, A, D), edge.double(), w.double(), wpc, bpc)
    ((agg_r * gagg.double()).sum() + (z_r * gz.double()).sum()).backward()
    Xg, wpg, bpg = X.to(DEV).requires_grad_(True), wp.to(DEV).requires_grad_(True), bp.to(DEV).requires_grad_(True)
    agg, z = ops.mpn(Xg, wpg, bpg, s_in.to(DEV), src=ops.SRC_SHARED, R=R, A=A,
                     row_mask=row_mask.to(torch.uint8).to(DEV), **kw)
    ((agg * gagg.to(DEV)).sum() + (z * gz.to(DEV)).sum()).backward()
    assert_close(agg, agg_r, 'agg')
    assert_close(z, z_r, 'z')
    # shard-sized calls: dX = W^T g_agg is a float32 library contraction over R = 66000 rows (ops._mpn_shared_gemm); against
    # the float64 value its smallest elements sit at 1.05e-4 of the element-wise measure -- 2e-4 there, 1e-4 at batch size
    assert_close(Xg.grad, Xc.grad, 'grad X', 2e-4 if R > 16384 else 1e-4)
    assert_close(wpg.grad, wpc.grad, 'grad wp')
    assert_close(bpg.grad, bpc.grad, 'grad bp')


def test_mpn_shared_batch_backward_is_bit_reproducible():
    """Batch-sized SHARED layer, default (deterministic) mode: the fused kernels with row-tile partials added in tile
    order -- two runs give the same bits for every gradient."""
    ops = _ops()
    R, A, D, N, C = 450, 185, 128, 300, 3
    E, _, row_mask, sims, wp, bp, gagg, gz = _rand_case(3, R, A, D, N, C)
    g = torch.Generator().manual_seed(2)
    X, col = torch.randn(A, D, generator=g), torch.randint(0, N, (A,), generator=g)
    runs = []
    for _ in range(2):
        Xg, wpg, bpg = X.to(DEV).requires_grad_(True), wp.to(DEV).requires_grad_(True), bp.to(DEV).requires_grad_(True)
        agg, z = ops.mpn(Xg, wpg, bpg, sims.to(DEV), src=ops.SRC_SHARED, R=R, A=A, row_mask=row_mask.to(torch.uint8).to(DEV),
                         sim_col=col.to(DEV))
        ((agg * gagg.to(DEV)).sum() + (z * gz.to(DEV)).sum()).backward()
        runs.append((agg.detach(), z.detach(), Xg.grad, wpg.grad, bpg.grad))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_gather_rows_matches_embedding_with_padding_idx(det):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    W = torch.randn(50, 16, generator=g)
    W[0] = 0
    ids = torch.randint(0, 50, (7, 3, 5), generator=g)
    ids[0, 0, :] = 0
    go = torch.randn(7, 3, 5, 16, generator=g)
    Wc = W.clone().requires_grad_(True)
    ref = F.embedding(ids, Wc, padding_idx=0)
    (ref * go).sum().backward()
    Wg = W.to(DEV).requires_grad_(True)
    out = ops.gather_rows(Wg, ids.to(DEV))
    (out * go.to(DEV)).sum().backward()
    assert torch.equal(out.cpu(), ref.detach())
    assert_close(Wg.grad, Wc.grad, 'grad W')
    assert float(Wg.grad[0].abs().max()) == 0.0


@pytest.mark.parametrize('shape', [(17, 5, 37), (33, 3, 448), (50, 1, 64)])     # scalar form; 16-byte form (H % 4 == 0)
def test_masked_sum(shape):
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(*shape, generator=g)
    mask = torch.rand(shape[0], shape[1], generator=g) > 0.3
    xc = x.clone().requires_grad_(True)
    ref = (xc * mask.unsqueeze(-1)).sum(1)
    go = torch.randn(ref.shape, generator=g)
    (ref * go).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    out = ops.masked_sum(xg, mask.to(DEV))
    (out * go.to(DEV)).sum().backward()
    assert_close(out, ref, 'masked_sum', norm_tol=1e-6)
    assert_close(xg.grad, xc.grad, 'masked_sum grad', norm_tol=1e-6)


@pytest.mark.parametrize('together', [True, False])
@pytest.mark.parametrize('B,C,D,A1,A2', [(5, 1, 8, 7, 3), (37, 3, 30, 57, 183), (300, 2, 64, 42, 260), (1, 4, 12, 1, 33)])
def test_subgraph_embedding_slots_equal_cat_and_masked_sum(B, C, D, A1, A2, together, monkeypatch):
    """ops.subgraph_embedding (component-embedding pieces + deferred shared-anchor read-outs, each summed into its column
    slot) against the reference's tail written out in float64: relu(W * s + b) per (component, anchor), concatenation,
    masked sum over the components (S.py:286-303, mpn:122-131).  Values and every gradient; twice -> identical bits.
    ``together``: the tensor pieces in one launch each way (sgnn_masked_sum_slots_fwd / _bwd: batch-sized calls) or one
    vectorised launch per piece (shard-sized calls)."""
    ops = _ops()
    monkeypatch.setattr(ops, 'SLOTS_TOGETHER_BELOW', (1 << 22) if together else 0)
    g = torch.Generator().manual_seed(B * 131 + A2)
    R = B * C
    mask = torch.rand(B, C, generator=g) > 0.3
    x0, x1 = torch.randn(B, C, D, generator=g), torch.randn(B, C, D, generator=g)
    sims1 = torch.rand(R, A1, generator=g)                            # per-edge weights, columns in anchor order
    wide = torch.rand(R, A2 + 9, generator=g)                         # a wider similarity row, columns picked by sim_col
    col2 = torch.randperm(A2 + 9, generator=g)[:A2]
    s1, s2 = torch.randn(A1, generator=g), torch.randn(A2, generator=g)
    b1, b2, b0 = torch.randn(1, generator=g) * 0.3, torch.randn(1, generator=g) * 0.3, torch.tensor([0.2])
    go = torch.randn(B, 2 * D + A1 + A2 + 5, generator=g)

    def reference():
        leaves = [t.double().clone().requires_grad_(True) for t in (x0, x1, s1, b1, s2, b2, b0)]
        X0, X1, S1, B1, S2, B2, B0 = leaves
        m = mask.double().view(B, C, 1)
        r1 = torch.relu(sims1.double().view(B, C, A1) * S1 + B1)
        r2 = torch.relu(wide.double()[:, col2].view(B, C, A2) * S2 + B2)
        r0 = torch.relu(B0.view(1, 1, 1).expand(B, C, 5))              # all-zero edge weights: read-out = bias
        out = (torch.cat([X0, r1, X1, r2, r0], dim=-1) * m).sum(1)
        (out * go.double()).sum().backward()
        return out, [t.grad for t in leaves]

    def product():
        leaves = [t.to(DEV).clone().requires_grad_(True) for t in (x0, x1, s1, b1, s2, b2, b0)]
        X0, X1, S1, B1, S2, B2, B0 = leaves
        mu8 = mask.reshape(-1).to(torch.uint8).to(DEV)
        p1 = ops.ReadoutPiece(sims1.to(DEV), None, S1, B1, A1, mu8, R)
        p2 = ops.ReadoutPiece(wide.to(DEV), col2.to(DEV), S2, B2, A2, mu8, R)
        p0 = ops.ReadoutPiece(None, None, torch.zeros(5, device=DEV), B0, 5, mu8, R)
        out = ops.subgraph_embedding([X0, p1, X1, p2, p0], mu8, B, C)
        (out * go.to(DEV)).sum().backward()
        dense = torch.cat([p1.dense(B, C), p2.dense(B, C), p0.dense(B, C)], dim=-1)
        return out, [t.grad for t in leaves], dense

    ref, ref_g = reference()
    out, out_g, dense = product()
    assert_close(out, ref.float(), 'subgraph embedding', norm_tol=1e-6)
    for nm, a, b in zip(('x0', 'x1', 's1', 'b1', 's2', 'b2', 'b0'), out_g, ref_g):
        assert_close(a, b.float(), 'subgraph embedding grad ' + nm, norm_tol=1e-5)
    # the materialised form of a deferred piece (attention read-out, gathered head) is the same read-out
    m = mask.view(B, C, 1).to(DEV)
    off = D
    assert_close((dense[..., :A1] * m).sum(1), out[:, off:off + A1], 'dense piece 1', norm_tol=1e-6)
    out2, out_g2, _ = product()
    assert torch.equal(out, out2) and all(torch.equal(a, b) for a, b in zip(out_g, out_g2))


@pytest.mark.parametrize('B,K', [(1, 2), (77, 3), (5000, 5), (50000, 3)])
def test_cross_entropy_with_accuracy_matches_torch(B, K):
    """sgnn_cross_entropy_fwd/_bwd against nn.CrossEntropyLoss() + calc_accuracy (S.py:133, su:108-124) in float64."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + K)
    x = torch.randn(B, K, generator=g) * 3
    x[::7] = x[::7].round()                                     # ties between classes: argmax takes the first maximum
    y = torch.randint(0, K, (B,), generator=g)
    xr = x.double().clone().requires_grad_(True)
    ref = F.cross_entropy(xr, y)
    (ref * 1.7).backward()
    acc_ref = (torch.argmax(x, 1) == y).float().mean()
    xg = x.to(DEV).requires_grad_(True)
    loss, acc = ops.cross_entropy_with_accuracy(xg, y.to(DEV))
    (loss * 1.7).backward()
    assert loss.dim() == 0 and acc.shape == (1,) and not acc.requires_grad
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert abs(float(acc) - float(acc_ref)) <= 1e-6
    assert_close(xg.grad, xr.grad.float(), 'cross entropy grad', tol=1e-4, norm_tol=1e-6)
    x2 = x.to(DEV).requires_grad_(True)
    l2, _ = ops.cross_entropy_with_accuracy(x2, y.to(DEV))
    (l2 * 1.7).backward()
    assert torch.equal(l2, loss) and torch.equal(x2.grad, xg.grad)


def test_cross_entropy_ignore_index_and_bad_labels():
    """nn.CrossEntropyLoss() semantics for labels outside [0, K): ignore_index (-100) rows are left out of the MEAN as
    well as of the sum; any other bad label raises in the library and must not pass silently here (NaN loss)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, K = 3000, 4
    x = torch.randn(B, K, generator=g)
    y = torch.randint(0, K, (B,), generator=g)
    y[::3] = -100
    xr = x.double().clone().requires_grad_(True)
    ref = F.cross_entropy(xr, y)
    ref.backward()
    xg = x.to(DEV).requires_grad_(True)
    loss, acc = ops.cross_entropy_with_accuracy(xg, y.to(DEV))
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert_close(xg.grad, xr.grad.float(), 'cross entropy grad with ignored rows', tol=1e-4, norm_tol=1e-6)
    assert torch.equal(xg.grad[::3], torch.zeros_like(xg.grad[::3]))
    assert abs(float(acc) - float((torch.argmax(x, 1) == y).float().mean())) <= 1e-6      # accuracy: every row counts
    y[1] = K
    bad, _ = ops.cross_entropy_with_accuracy(x.to(DEV), y.to(DEV))
    assert torch.isnan(bad)
    with pytest.raises((IndexError, RuntimeError)):
        F.cross_entropy(x, y)


def test_missing_library_fails_loudly(monkeypatch):
    """No silent CPU fallback: a CPU tensor is rejected, and so is a missing library."""
    ops = _ops()
    from subgnn_amd import _lib
    with pytest.raises(_lib.SubgnnHipError):
        ops.masked_sum(torch.zeros(2, 2, 4), torch.ones(2, 2, dtype=torch.bool))


def test_warm_up_moves_no_random_stream():
    """ops.warm_up runs once per process inside the FIRST model's constructor, after the caller has seeded torch: it must leave
    the CPU generator and the device's Philox state where they were, or the first model of a process draws other dropout masks
    than every later same-seed model."""
    ops = _ops()
    torch.manual_seed(1234)
    cpu0, dev0 = torch.get_rng_state(), torch.cuda.get_rng_state()
    want = torch.rand(4, device=DEV)
    torch.manual_seed(1234)
    ops._WARM.clear()
    assert ops.warm_up(torch.device(DEV)) > 0
    assert torch.equal(torch.get_rng_state(), cpu0) and torch.equal(torch.cuda.get_rng_state(), dev0)
    assert torch.equal(torch.rand(4, device=DEV), want)


@pytest.mark.parametrize('R,H,C', [(5, 8, 1), (70, 37, 7), (193, 420, 1), (64, 615, 4)])
def test_attn_scores_mfma(R, H, C):
    """Matrix-core attention scores vs dense torch fp32: asymmetric operands (catches a row/column
    swap of the accumulator map), H not a multiple of the MFMA tile, rows not a multiple of 32."""
    ops = _ops()
    g = torch.Generator().manual_seed(R * 1000 + H)
    X = torch.randn(R - R % C if R % C else R, H, generator=g)
    R = X.shape[0]
    U = torch.randn(H, H, generator=g) / H ** 0.5
    qW = torch.randn(R // C, H, generator=g)
    v = torch.randn(H, generator=g)
    Xc, Uc, qc, vc = [t.clone().requires_grad_(True) for t in (X, U, qW, v)]
    ref = (torch.tanh(torch.repeat_interleave(qc, C, dim=0) + Xc @ Uc) * vc).sum(1)
    go = torch.randn(R, generator=g)
    (ref * go).sum().backward()
    Xg, Ug, qg, vg = [t.to(DEV).requires_grad_(True) for t in (X, U, qW, v)]
    out = ops.attn_scores(Xg, Ug, qg, vg, C)
    (out * go.to(DEV)).sum().backward()
    assert_close(out, ref, 'scores', norm_tol=2e-5)
    for a, b, nm in ((Xg, Xc, 'X'), (Ug, Uc, 'U'), (qg, qc, 'qW'), (vg, vc, 'v')):
        assert_close(a.grad, b.grad, 'grad ' + nm)


@pytest.mark.parametrize('kernel', ['mfma', 'auto'])
@pytest.mark.parametrize('R,H,C', [(5, 8, 1), (70, 37, 7), (193, 420, 1), (640, 615, 4), (300, 640, 1), (129, 128, 1), (4100, 420, 1)])
def test_attn_scores_mfma_half_operands(R, H, C, kernel, monkeypatch):
    """The fp16 form (v_mfma_f32_32x32x16_f16, BASELINE configs[4]): against fp32 arithmetic on the half-ROUNDED
    operands it is exact up to the accumulation order (products of halves are exact in fp32); against the
    unrounded reference it is within half precision.  Sizes cover all three register instantiations, partial row
    tiles and column panels, several rows per batch; the backward pass is the fp32 recomputation."""
    ops = _ops()
    if kernel == 'mfma':
        monkeypatch.setattr(ops, 'ATTN_F16_MIN_ROWS', 0)       # the hand-written kernel at every size (by default a
    g = torch.Generator().manual_seed(R * 1000 + H)            # batch of a few hundred rows takes the library GEMM)
    R = R - R % C
    X = torch.randn(R, H, generator=g)
    U = torch.randn(H, H, generator=g) / H ** 0.5
    qW = torch.randn(R // C, H, generator=g)
    v = torch.randn(H, generator=g)
    ref_h = (torch.tanh(torch.repeat_interleave(qW, C, dim=0) + X.half().float() @ U.half().float()) * v).sum(1)
    ref = (torch.tanh(torch.repeat_interleave(qW, C, dim=0) + X @ U) * v).sum(1)
    Xg, Ug, qg, vg = [t.to(DEV).requires_grad_(True) for t in (X, U, qW, v)]
    out = ops.attn_scores(Xg, Ug, qg, vg, C, half_operands=True)
    assert_close(out, ref_h, 'scores vs half-rounded operands', norm_tol=5e-5)
    assert_close(out, ref, 'scores vs fp32', 2e-2)
    out.sum().backward()
    assert Xg.grad is not None and torch.isfinite(Xg.grad).all() and float(Ug.grad.abs().max()) > 0
    # beyond the register budget of the half kernel (H > 640) the library GEMM on the rounded operands serves the call
    if H == 8:
        Xb = torch.randn(40, 700, generator=g)
        Ub = torch.randn(700, 700, generator=g) / 26
        qb, vb = torch.randn(40, 700, generator=g), torch.randn(700, generator=g)
        a = ops.attn_scores(Xb.to(DEV), Ub.to(DEV), qb.to(DEV), vb.to(DEV), 1, half_operands=True)
        want = (torch.tanh(qb + Xb.half().float() @ Ub.half().float()) * vb).sum(1)
        assert_close(a, want, 'H = 700', 1e-3)


# ---- fused bidirectional LSTM layer (sgnn_lstm_fwd / _bwd) vs torch's nn.LSTM in fp32 on the CPU --------

def _lstm_params(m, layer):
    return [getattr(m, '%s_l%d%s' % (n, layer, sfx)) for sfx in ('', '_reverse')
            for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]


@pytest.mark.parametrize('H,I', [(64, 64), (64, 128), (32, 32), (32, 64), (128, 128), (128, 256), (64, 20)])
@pytest.mark.parametrize('B,Tn', [(1, 1), (7, 3), (210, 10), (19, 20), (8, 10)])
def test_bilstm_layer_matches_torch(H, I, B, Tn):
    from subgnn_amd import ops
    torch.manual_seed(H + I + B + Tn)
    ref = torch.nn.LSTM(I, H, num_layers=1, batch_first=True, bidirectional=True)
    x = torch.randn(B, Tn, I)
    g = torch.randn(B, Tn, 2 * H)
    xr = x.clone().requires_grad_()
    yr, _ = ref(xr)
    (yr * g).sum().backward()
    dev_params = [p.detach().to(DEV).requires_grad_() for p in _lstm_params(ref, 0)]
    xd = x.to(DEV).requires_grad_()
    y = ops.bilstm_layer(xd, dev_params)
    (y * g.to(DEV)).sum().backward()
    tol = dict(rtol=1e-4, atol=2e-5)
    assert torch.allclose(y.cpu(), yr.detach(), **tol)
    assert torch.allclose(xd.grad.cpu(), xr.grad, **tol)
    for p, q in zip(dev_params, _lstm_params(ref, 0)):
        scale = max(float(q.grad.abs().max()), 1.0)
        assert torch.allclose(p.grad.cpu() / scale, q.grad / scale, rtol=1e-4, atol=2e-5), (p.shape,)


@pytest.mark.parametrize('layers,agg,D', [(1, 'last', 64), (2, 'last', 64), (2, 'sum', 64), (2, 'last', 128)])
def test_lstm_module_uses_fused_layers_and_matches_library(layers, agg, D):
    """The module of SubGNN.py:60-88 with the kernel inside equals the same parameters run through the
    library nn.LSTM (state-dict compatible: same parameter names)."""
    from subgnn_amd.SubGNN import LSTM
    torch.manual_seed(3)
    m = LSTM(D, D, dropout=0.0, num_layers=layers, aggregator=agg).to(DEV)
    assert sorted(k for k in m.state_dict() if k.startswith('lstm.')) == sorted('lstm.' + k for k in m.lstm.state_dict())
    x = torch.randn(30, 12, D, device=DEV, requires_grad=True)
    out = m(x)
    out.square().sum().backward()
    gx, gw = x.grad.clone(), [p.grad.clone() for p in m.parameters()]
    # reference: the same parameters through the library's nn.LSTM + Linear in FLOAT64 on the CPU (the device library in
    # float32 is itself ~1e-4 away from it on the weight gradients: rounds 1-3 compared against that with rtol 1e-3)
    import copy
    r = copy.deepcopy(m).cpu().double()
    xr = x.detach().cpu().double().requires_grad_(True)
    lib_out, _ = r.lstm(xr)
    lib = r.linear(lib_out[:, -1, :] if agg == 'last' else lib_out.sum(dim=1))
    lib.square().sum().backward()
    assert_close(out, lib, 'LSTM module output')
    assert_close(gx, xr.grad, 'LSTM module d x')
    for (nm, q), a in zip(r.named_parameters(), gw):
        assert_close(a, q.grad, 'LSTM module d ' + nm)


def test_lstm_bias_gradients_do_not_share_memory():
    """bias_ih and bias_hh get equal gradient values in separate buffers: sharing one buffer made clip_grad_norm_'s in-place
    multi-tensor scaling hit it twice from concurrent chunks (g c or g c^2 by timing -- the cross-process loss drift of
    rounds 2-3).  After clipping, every bias gradient is exactly coefficient x the unclipped gradient."""
    from subgnn_amd.SubGNN import LSTM
    torch.manual_seed(5)
    m = LSTM(64, 64, dropout=0.0, num_layers=2, aggregator='last').to(DEV)
    x = torch.randn(40, 10, 64, device=DEV)
    m(x).square().sum().backward()
    named = dict(m.named_parameters())
    biases = [k for k in named if 'bias' in k and k.startswith('lstm.')]
    assert len(biases) == 8
    ptrs = [named[k].grad.untyped_storage().data_ptr() for k in biases]
    spans = sorted((named[k].grad.data_ptr(), named[k].grad.data_ptr() + named[k].grad.numel() * 4) for k in biases)
    assert all(a[1] <= b[0] for a, b in zip(spans[:-1], spans[1:])), 'bias gradients overlap in memory'
    for sfx in ('l0', 'l0_reverse', 'l1', 'l1_reverse'):
        assert torch.equal(named['lstm.bias_ih_' + sfx].grad, named['lstm.bias_hh_' + sfx].grad)
    before = {k: named[k].grad.clone() for k in biases}
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.01)
    coef = torch.clamp(0.01 / (total + 1e-6), max=1.0)
    assert float(coef) < 0.5
    for k in biases:
        assert torch.equal(named[k].grad, before[k] * coef), k
    del ptrs


def test_lstm_unsupported_sizes_stay_on_the_library():
    from subgnn_amd import ops
    from subgnn_amd.SubGNN import LSTM
    assert ops.lstm_supported(64, 64) and ops.lstm_supported(128, 64) and ops.lstm_supported(128, 128)
    assert not ops.lstm_supported(48, 48)
    m = LSTM(48, 48).to(DEV)
    assert m(torch.randn(5, 4, 48, device=DEV)).shape == (5, 48)


# ---- a12 update(): fused relu([x | aggr] W^T + b) ------------------------------------------------------

@pytest.mark.parametrize('D', [32, 64, 128])
@pytest.mark.parametrize('R', [1, 31, 33, 257, 5000, 17000])          # 17000: the one-wavefront-per-row-block form
def test_update_layer_matches_torch(R, D):
    """sgnn_update_fwd / sgnn_update_bwd against cat + nn.Linear + relu in float64 (subgraph_mpn.py:233-241):
    output and all four gradients within 1e-5; two runs give the same bits."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(R * 1000 + D)
    x, a = torch.randn(R, D, generator=g), torch.randn(R, D, generator=g)
    W, b = torch.randn(D, 2 * D, generator=g) / (2 * D) ** 0.5, torch.randn(D, generator=g)
    go = torch.randn(R, D, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (x, a, W, b)]
    ref = torch.relu(torch.cat([ref_in[0], ref_in[1]], 1) @ ref_in[2].t() + ref_in[3])
    ref.backward(go.double())
    runs = []
    for _ in range(2):
        ins = [t.to(DEV).requires_grad_(True) for t in (x, a, W, b)]
        out = ops.update_layer(*ins)
        out.backward(go.to(DEV))
        runs.append([out.detach()] + [t.grad for t in ins])
    for u, v in zip(*runs):
        assert torch.equal(u, v)
    assert_close(runs[0][0], ref.detach().float(), 'update out', norm_tol=1e-5)
    for nm, got, want in zip(('x', 'aggr', 'W', 'b'), runs[0][1:], ref_in):
        assert_close(got, want.grad.float(), 'update grad ' + nm, norm_tol=2e-5)


def test_update_layer_partial_gradients_and_other_widths():
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(1)
    R, D = 300, 64
    x, a = torch.randn(R, D, generator=g).to(DEV), torch.randn(R, D, generator=g).to(DEV).requires_grad_(True)
    W, b = (torch.randn(D, 2 * D, generator=g) / 11).to(DEV), torch.randn(D, generator=g).to(DEV)
    out = ops.update_layer(x, a, W, b)                                  # only aggr needs a gradient
    out.sum().backward()
    want = ((out > 0).float() @ W[:, D:])
    assert_close(a.grad, want, 'grad aggr only', norm_tol=1e-5)
    # a width without a fused kernel keeps the library form
    D = 48
    x, a = torch.randn(R, D, generator=g).to(DEV), torch.randn(R, D, generator=g).to(DEV)
    W, b = (torch.randn(D, 2 * D, generator=g) / 9).to(DEV), torch.randn(D, generator=g).to(DEV)
    assert_close(ops.update_layer(x, a, W, b), torch.relu(torch.cat([x, a], 1) @ W.t() + b), 'width 48', norm_tol=1e-5)


# ---- clip + Adam with the large parameter in one HIP pass (optim.ClipAdam) --------------------------------------

@pytest.mark.parametrize('max_norm', [None, 0.5, 1e6])
def test_clip_adam_matches_torch(max_norm):
    """optim.ClipAdam == clip_grad_norm_ + torch.optim.Adam over several steps (a 'large' table of 1003 x 64 floats,
    whose element count is not a multiple of 4 x 256, and three small parameters, one of them without a gradient)."""
    from subgnn_amd import optim, ops
    g = torch.Generator().manual_seed(7)
    shapes = [(1003, 64), (64, 128), (64,), (5, 3)]
    init = [torch.randn(*s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    got = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    o_ref = torch.optim.Adam(ref, lr=0.01)
    o_got = optim.ClipAdam(got, lr=0.01, max_norm=max_norm, big_bytes=1003 * 64 * 4)
    assert len(o_got.big) == 1 and len(o_got.small) == 3
    for it in range(4):
        grads = [torch.randn(*s, generator=g).to(DEV) * (5.0 if it == 1 else 0.1) for s in shapes]
        for ps, opt in ((ref, o_ref), (got, o_got)):
            for i, (p, gr) in enumerate(zip(ps, grads)):
                p.grad = None if i == 3 and it < 2 else gr.clone()
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        o_ref.step(); o_got.step()
        assert got[0].grad is None                                  # consumed, zeroed and handed back
        kept = got[0].__dict__.get('_sgnn_zeroed')                  # it hangs on ITS parameter, nowhere else
        buf = ops.take_zeroed(got[0], shapes[0], torch.device(DEV))
        assert buf is kept and float(buf.abs().max()) == 0.0 and '_sgnn_zeroed' not in got[0].__dict__
        assert ops.take_zeroed(ref[0], shapes[0], torch.device(DEV)) is not buf     # another parameter gets a fresh one
        ops.release_zeroed(got[0], buf)
        o_ref.zero_grad(); o_got.zero_grad()
        for a, b, s in zip(got, ref, shapes):
            assert_close(a.detach(), b.detach(), 'ClipAdam step %d %s' % (it, (s,)), norm_tol=2e-6)
    o_got.release()
    assert '_sgnn_zeroed' not in got[0].__dict__


@pytest.mark.parametrize('capturable', [False, True])
def test_clip_adam_fused_tail_many_tensors(capturable):
    """The two-launch optimizer tail (sgnn_optim_sumsq + sgnn_optim_adam) on 150 parameters -- more than one launch group of
    72 --: sizes from 1 to 70 001 elements (one spans several workgroups, most end in a scalar tail), every third one a view
    4 bytes past a 16-byte boundary (scalar path), parameters whose gradient is missing on some steps (torch.optim.Adam
    skips them and their step counts fall behind: host counts and device counts alike), against clip_grad_norm_ +
    torch.optim.Adam; the reported total norm against clip_grad_norm_'s; then make_eager() (device counts -> host counts)
    and two more steps.  Also ClipAdam(fuse_tail=False), the library-call form, gives the same parameters."""
    from subgnn_amd import optim
    g = torch.Generator().manual_seed(11)
    sizes = [1, 2, 3, 5, 64, 127, 1000, 4096, 4097, 70001] + [int(x) for x in torch.randint(1, 3000, (140,), generator=g)]
    init = [torch.randn(n, generator=g) for n in sizes]

    def make():
        out = []
        for i, t in enumerate(init):
            if i % 3 == 2:
                base = torch.zeros(t.numel() + 1, device=DEV)
                view = base[1:]
                view.copy_(t)
                assert view.data_ptr() % 16 == 4
                out.append(torch.nn.Parameter(view))
            else:
                out.append(torch.nn.Parameter(t.clone().to(DEV)))
        return out
    ref, got, lib_form = make(), make(), make()
    o_ref = torch.optim.Adam(ref, lr=0.01)
    o_got = optim.ClipAdam(got, lr=0.01, max_norm=0.7, big_bytes=70001 * 4, capturable=capturable)
    o_lib = optim.ClipAdam(lib_form, lr=0.01, max_norm=0.7, big_bytes=70001 * 4, capturable=capturable, fuse_tail=False)
    assert o_got.tail is not None and o_got.small_opt is None and len(o_got.big) == 1 and o_lib.tail is None
    for it in range(6):
        if it == 4:
            o_got.make_eager(); o_lib.make_eager()
            assert o_got.counters is None
        grads = [torch.randn(n, generator=g).to(DEV) * (3.0 if it % 2 else 0.05) for n in sizes]
        for ps in (ref, got, lib_form):
            for i, (p, gr) in enumerate(zip(ps, grads)):
                p.grad = None if (i % 7 == 3 and it in (1, 2)) or (i == 9 and it == 3) else gr.clone()
        total = torch.nn.utils.clip_grad_norm_(ref, 0.7)
        o_ref.step(); o_got.step(); o_lib.step()
        coef, norm = o_got.last_clip.tolist()
        assert abs(norm - float(total)) <= 1e-5 * float(total), (norm, float(total))
        assert abs(coef - min(1.0, 0.7 / (float(total) + 1e-6))) <= 1e-6
        for opt in (o_ref, o_got, o_lib):
            opt.zero_grad()
        for k, (a, b, c) in enumerate(zip(got, ref, lib_form)):
            assert_close(a.detach(), b.detach(), 'fused tail step %d parameter %d (%d elements)' % (it, k, sizes[k]), norm_tol=2e-6)
            assert_close(c.detach(), b.detach(), 'library form step %d parameter %d' % (it, k), norm_tol=2e-6)
    steps = [o_got.state[id(p)]['step'] for p in got]
    assert steps[0] == 6 and steps[3] == 4 and steps[9] == 5, steps[:12]


@pytest.mark.parametrize('D', [4, 32, 64, 128, 256, 48])
def test_clip_adam_skips_untouched_table_rows_bit_exactly(D):
    """The table's Adam update skips rows that have no gradient now and never had one (m = v = 0: Adam leaves such a row as it
    is): bit-identical to the dense update (skip_untouched_rows=False) over steps whose gradients touch different row sets --
    a row touched once keeps decaying afterwards --, and equal to clip_grad_norm_ + torch.optim.Adam.  D = 48 is not a power of
    two: no skipping there, same results."""
    from subgnn_amd import optim
    g = torch.Generator().manual_seed(D)
    rows = 5001
    init = [torch.randn(rows, D, generator=g), torch.randn(D, 7, generator=g)]
    models = [[torch.nn.Parameter(t.clone().to(DEV)) for t in init] for _ in range(3)]
    o_skip = optim.ClipAdam(models[0], lr=0.01, max_norm=0.5, big_bytes=rows * D * 4)
    o_dense = optim.ClipAdam(models[1], lr=0.01, max_norm=0.5, big_bytes=rows * D * 4, skip_untouched_rows=False)
    o_ref = torch.optim.Adam(models[2], lr=0.01)
    assert (len(o_skip.tail.seen) == 1) == (D != 48) and not o_dense.tail.seen
    touched = torch.zeros(rows, dtype=torch.bool)
    for it in range(6):
        pick = torch.rand(rows, generator=g) < (0.0 if it == 3 else 0.15)           # step 3: no row at all
        pick[0] = False                                                             # the PAD row never has a gradient
        touched |= pick
        gt = torch.randn(rows, D, generator=g) * pick.unsqueeze(1)
        gs = torch.randn(D, 7, generator=g)
        for ps in models:
            ps[0].grad, ps[1].grad = gt.clone().to(DEV), gs.clone().to(DEV)
        torch.nn.utils.clip_grad_norm_(models[2], 0.5)
        o_skip.step(); o_dense.step(); o_ref.step()
        for opt in (o_skip, o_dense, o_ref):
            opt.zero_grad()
        assert torch.equal(models[0][0], models[1][0]) and torch.equal(models[0][1], models[1][1]), it
        for name in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(o_skip.state[id(models[0][0])][name], o_dense.state[id(models[1][0])][name]), (it, name)
        assert_close(models[0][0].detach(), models[2][0].detach(), 'table after step %d' % it, norm_tol=2e-6)
        if D != 48:
            assert torch.equal(o_skip.tail.seen[0].cpu().bool(), touched), it
    assert torch.equal(models[0][0][~touched.to(DEV)].cpu(), init[0][~touched])    # untouched rows: the initial bits


@pytest.mark.parametrize('capturable,fuse_tail', [(False, True), (True, True), (False, False)])
def test_clip_adam_checkpoint_round_trip_and_param_groups(capturable, fuse_tail):
    """ClipAdam has torch's optimizer surface: ``param_groups[0]['lr']`` written by a scheduler is what the next
    step uses; ``state_dict`` / ``load_state_dict`` carry moments, step counts and the row-skip bytes, so that a restored
    optimizer continues bit for bit -- and a checkpoint WITHOUT the row-skip bytes (moments restored from elsewhere) does too
    (rows with a non-zero moment are marked seen on load: a skipped row would otherwise stop decaying)."""
    from subgnn_amd import optim
    g = torch.Generator().manual_seed(5)
    rows, D = 3001, 32
    init = [torch.randn(rows, D, generator=g), torch.randn(D, 9, generator=g), torch.randn(D, generator=g)]

    def fresh():
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        return ps, optim.ClipAdam(ps, lr=0.01, max_norm=0.5, big_bytes=rows * D * 4, capturable=capturable, fuse_tail=fuse_tail)

    def grads(it):
        gg = torch.Generator().manual_seed(100 + it)
        pick = (torch.rand(rows, generator=gg) < 0.1).unsqueeze(1)
        pick[0] = False
        return [torch.randn(rows, D, generator=gg) * pick, torch.randn(D, 9, generator=gg), torch.randn(D, generator=gg)]

    def run(ps, opt, its):
        for it in its:
            for p, gr in zip(ps, grads(it)):
                p.grad = gr.clone().to(DEV)
            if it == 2:
                opt.param_groups[0]['lr'] = 0.003                     # a scheduler's write
            opt.step()
            opt.zero_grad()
    a_ps, a = fresh()
    run(a_ps, a, range(5))
    assert a.lr == 0.003
    b_ps, b = fresh()
    run(b_ps, b, range(3))
    sd = b.state_dict()
    assert set(sd) >= {'state', 'param_groups'} and sd['param_groups'][0]['lr'] == 0.003 and sd['state'][0]['step'] == 3
    for strip in (False, True):
        c_ps, c = fresh()
        for p, q in zip(c_ps, b_ps):
            p.data.copy_(q.data)
        sd2 = {'state': {k: {n: v for n, v in e.items() if not (strip and n == 'rows_seen')} for k, e in sd['state'].items()},
               'param_groups': sd['param_groups'], 'small': sd.get('small')}
        c.load_state_dict(sd2)
        run(c_ps, c, range(3, 5))
        for p, q in zip(c_ps, a_ps):
            assert torch.equal(p.detach(), q.detach()), (capturable, fuse_tail, strip)


def test_clip_adam_follows_a_parameter_whose_storage_was_replaced():
    """``p.data = other_tensor`` between two steps (a reloaded checkpoint assigned rather than copied): the next step updates the
    parameter where it lives now."""
    from subgnn_amd import optim
    g = torch.Generator().manual_seed(5)
    init = [torch.randn(300, 64, generator=g), torch.randn(64, 9, generator=g)]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    got = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    o_ref, o_got = torch.optim.Adam(ref, lr=0.01), optim.ClipAdam(got, lr=0.01, max_norm=0.5, big_bytes=300 * 64 * 4)
    for it in range(3):
        if it == 1:
            for p in got:
                p.data = p.data.clone()                       # new storage, same values
        grads = [torch.randn(*t.shape, generator=g).to(DEV) for t in init]
        for ps in (ref, got):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        o_ref.step(); o_got.step()
        for a, b in zip(got, ref):
            assert_close(a.detach(), b.detach(), 'step %d' % it, norm_tol=2e-6)


def test_clip_adam_fused_tail_replays_from_a_hipgraph():
    """The tail recorded into a hipGraph (device step counts) and replayed five times == five eager steps of torch's."""
    from subgnn_amd import optim
    g = torch.Generator().manual_seed(12)
    sizes = [70001, 640, 3, 129]
    init = [torch.randn(n, generator=g) for n in sizes]
    ref = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    got = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    o_ref = torch.optim.Adam(ref, lr=0.01)
    o_got = optim.ClipAdam(got, lr=0.01, max_norm=0.3, big_bytes=70001 * 4, capturable=True)
    static = [torch.zeros(n, device=DEV) for n in sizes]
    for p, sg in zip(got, static):
        p.grad = sg
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(graph, stream=stream):
            for p, sg in zip(got, static):
                p.grad = sg
            o_got.step()
    torch.cuda.current_stream().wait_stream(stream)
    assert int(o_got.counters.sum()) == 0                            # recording runs nothing
    for it in range(5):
        grads = [torch.randn(n, generator=g).to(DEV) * (2.0 if it == 2 else 0.1) for n in sizes]
        for p, gr, sg in zip(ref, grads, static):
            p.grad = gr.clone()
            sg.copy_(gr)
        torch.nn.utils.clip_grad_norm_(ref, 0.3)
        o_ref.step()
        graph.replay()
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(got, ref)):
            assert_close(a.detach(), b.detach(), 'replayed tail step %d parameter %d' % (it, k), norm_tol=2e-6)
    assert o_got.counters.tolist() == [5, 5, 5, 5]


# ---- a18 deterministic table-gradient scatter -------------------------------------------------------

@pytest.mark.parametrize('D', [8, 64, 128, 200])
def test_scatter_add_rows_sorted_matches_index_add_and_is_reproducible(D):
    """sgnn_scatter_add_rows_sorted against a float64 index_add: PAD keys, a hub target with thousands of
    edges (a chain over hundreds of 64-edge runs), both coefficient terms, explicit and implicit source rows;
    two runs give the same bits."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(D)
    n_rows, R, A = 5000, 700, 13
    E = R * A
    keys = torch.randint(0, n_rows, (E,), generator=g)
    keys[torch.rand(E, generator=g) < 0.3] = 7                         # a hub: ~2700 edges
    keys[torch.rand(E, generator=g) < 0.1] = 0                         # PAD
    keys[:200] = 11                                                    # a segment that starts at position 0 after the sort? no: low key
    G = torch.randn(R, D, generator=g)
    c1, c2, v = torch.randn(E, generator=g), torch.randn(E, generator=g), torch.randn(D, generator=g)
    want = torch.zeros(n_rows, D, dtype=torch.float64)
    rows = torch.arange(E) // A
    contrib = c1.double().unsqueeze(1) * G.double()[rows] + c2.double().unsqueeze(1) * v.double()
    contrib[keys == 0] = 0
    want.index_add_(0, keys, contrib)
    want[0] = 0
    d = lambda t: t.to(DEV)
    outs = []
    for _ in range(2):
        table = torch.zeros(n_rows, D, device=DEV)
        ops.scatter_add_rows(table, d(keys.to(torch.int32)), G=d(G), edges_per_row=A, c1=d(c1), c2=d(c2), v=d(v))
        outs.append(table)
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0], want.float(), 'scatter', norm_tol=1e-5)
    # explicit rows, no coefficients, adds on top of what is there
    er = torch.randint(0, R, (E,), generator=g)
    table = torch.ones(n_rows, D, device=DEV)
    ops.scatter_add_rows(table, d(keys.to(torch.int32)), G=d(G), edge_row=d(er.to(torch.int32)))
    want2 = torch.ones(n_rows, D, dtype=torch.float64)
    c = G.double()[er]
    c[keys == 0] = 0
    want2.index_add_(0, keys, c)
    want2[0] = 1
    assert_close(table, want2.float(), 'scatter (rows)', norm_tol=1e-5)


@pytest.mark.parametrize('D,n_lists', [(32, 3), (64, 7), (128, 45), (200, 4)])
def test_scatter_lists_together_equal_one_after_the_other(D, n_lists):
    """sgnn_scatter_add_rows_multi (ops._GradAcc.flush: the short lists of a step packed, sorted once, scattered once) against
    the same lists scattered one after the other and against a float64 index_add: lists with and without source rows, with
    explicit rows and rows by division, with and without the two coefficient terms, PAD keys, a hub target shared by every
    list; 45 lists span two pack launches.  Two runs give the same bits."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(D + n_lists)
    n_rows = 3000
    table_owner = torch.nn.Parameter(torch.zeros(n_rows, D, device=DEV))
    want = torch.zeros(n_rows, D, dtype=torch.float64)
    jobs = []
    for k in range(n_lists):
        R, A = int(torch.randint(1, 60, (1,), generator=g)), int(torch.randint(1, 40, (1,), generator=g))
        E = R * A
        keys = torch.randint(0, n_rows, (E,), generator=g)
        keys[torch.rand(E, generator=g) < 0.2] = 11                                  # a hub every list adds to
        keys[torch.rand(E, generator=g) < 0.1] = 0                                   # PAD
        kind = k % 4
        G = torch.randn(R, D, generator=g) if kind != 3 else None                    # kind 3: only the c2 * v term
        rows = torch.randint(0, R, (E,), generator=g) if kind == 1 else torch.arange(E) // A
        c1 = torch.randn(E, generator=g) if kind in (0, 2) else None
        c2 = torch.randn(E, generator=g) if kind in (2, 3) else None
        v = torch.randn(D, generator=g) if c2 is not None else None
        contrib = torch.zeros(E, D, dtype=torch.float64)
        if G is not None:
            contrib += (c1.double().unsqueeze(1) if c1 is not None else 1.0) * G.double()[rows]
        if c2 is not None:
            contrib += c2.double().unsqueeze(1) * v.double()
        contrib[keys == 0] = 0
        want.index_add_(0, keys, contrib)
        d = lambda t: None if t is None else t.to(DEV)
        jobs.append(dict(keys=d(keys.to(torch.int32)), G=d(G), edge_row=d(rows.to(torch.int32)) if kind == 1 else None,
                         edges_per_row=A, c1=d(c1), c2=d(c2), v=d(v)))
    want[0] = 0
    outs = []
    for _ in range(2):
        acc = ops._GradAcc(table_owner)
        buf = acc.buffer((n_rows, D), torch.device(DEV))
        buf.zero_()
        for j in jobs:
            ops.scatter_add_rows(buf, j['keys'], G=j['G'], edge_row=j['edge_row'], edges_per_row=j['edges_per_row'], c1=j['c1'],
                                 c2=j['c2'], v=j['v'], together=acc)
        assert len(acc.jobs) == n_lists and float(buf.abs().max()) == 0.0            # nothing has been added yet
        acc.flush()
        assert not acc.jobs
        outs.append(buf.clone())
    assert torch.equal(outs[0], outs[1])
    one_by_one = torch.zeros(n_rows, D, device=DEV)
    for j in jobs:
        ops.scatter_add_rows(one_by_one, j['keys'], G=j['G'], edge_row=j['edge_row'], edges_per_row=j['edges_per_row'], c1=j['c1'],
                             c2=j['c2'], v=j['v'])
    assert_close(outs[0], want.float(), 'lists together vs float64', norm_tol=1e-5)
    assert_close(outs[0], one_by_one, 'lists together vs one after the other', norm_tol=1e-5)


@pytest.mark.parametrize('n,max_key', [(1, 5), (2, 3), (3, 1 << 30), (63, 1), (1000, 999), (4096, 1_000_000), (4097, 1_000_000), (5000, (1 << 31) - 1),
                                       (8191, 17_000), (8192, 1_000_000), (8193, 57_000), (300_000, 1_000_000), (70_000, (1 << 31) - 1)])
def test_sort_edges_by_key_is_a_stable_sort(n, max_key):
    """Up to 8192 keys: the one-workgroup LDS sort (32-bit words while key bits + position bits <= 32, else 64-bit); beyond: rocPRIM."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, min(max_key, 5000) + 1, (n,), generator=g).to(torch.int32)
    keys[torch.rand(n, generator=g) < 0.05] = max_key                       # the top of the range (all key bits in use)
    sk, order = ops.sort_edges_by_key(keys.to(DEV), max_key)
    want_k, want_o = torch.sort(keys, stable=True)
    assert sk.dtype == torch.int32 and order.dtype == torch.int32
    assert torch.equal(sk.cpu(), want_k) and torch.equal(order.cpu().long(), want_o)


def test_scatter_add_rows_large_launch_form():
    """More than 8192 runs of 64 edges: the D <= 64 launch keeps 8 source rows in flight instead of the whole run."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(11)
    n_rows, R, A, D = 20000, 46000, 13, 64
    E = R * A
    keys = torch.randint(0, n_rows, (E,), generator=g)
    keys[torch.rand(E, generator=g) < 0.2] = 0
    keys[torch.rand(E, generator=g) < 0.05] = 9                      # a hub: a chain over ~470 runs
    G, c1 = torch.randn(R, D, generator=g), torch.randn(E, generator=g)
    want = torch.zeros(n_rows, D, dtype=torch.float64)
    contrib = c1.double().unsqueeze(1) * G.double()[torch.arange(E) // A]
    contrib[keys == 0] = 0
    want.index_add_(0, keys, contrib)
    outs = []
    for _ in range(2):
        table = torch.zeros(n_rows, D, device=DEV)
        ops.scatter_add_rows(table, keys.to(torch.int32).to(DEV), G=G.to(DEV), edges_per_row=A, c1=c1.to(DEV))
        outs.append(table)
    assert torch.equal(outs[0], outs[1])
    assert_close(outs[0], want.float(), 'scatter (large)', norm_tol=1e-5)


def test_scatter_add_rows_takes_a_kept_order():
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(5)
    n_rows, R, A, D = 4000, 500, 9, 64
    keys = torch.randint(0, n_rows, (R * A,), generator=g).to(torch.int32).to(DEV)
    G = torch.randn(R, D, generator=g).to(DEV)
    a, b = torch.zeros(n_rows, D, device=DEV), torch.zeros(n_rows, D, device=DEV)
    ops.scatter_add_rows(a, keys, G=G, edges_per_row=A)
    pre = ops.sort_edges_by_key(keys, n_rows - 1)
    ops.scatter_add_rows(b, keys, G=G, edges_per_row=A, presorted=pre)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.scatter_add_rows(b, keys[:-1].contiguous(), G=G, edges_per_row=A, presorted=pre)


def test_table_gradient_is_bit_reproducible():
    """The embedding-table gradient of the fused ops (component embeddings sum / max, a GATHER message-passing
    layer with duplicated anchors, gather_rows) is the same bits on every run -- and equal to the atomics
    path within float tolerance."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(3)
    N, D, R, A, L = 3000, 64, 900, 11, 9
    E0 = torch.randn(N + 1, D, generator=g)
    E0[0] = 0
    cc = torch.randint(0, 40, (R, L), generator=g)                    # few distinct nodes: long segments
    ids = torch.randint(0, 60, (R, A), generator=g)
    sims = torch.rand(R, A, generator=g)
    wp, bp = torch.randn(D, generator=g) * 0.1, torch.zeros(1)
    walk = torch.randint(0, 50, (30, 5), generator=g)

    def run(det):
        ops.DETERMINISTIC = det
        try:
            E = E0.to(DEV).requires_grad_(True)
            Et = ops.tap_table(E)
            sets = ops.Ragged(torch.arange(R + 1, dtype=torch.int64, device=DEV) * L, cc.reshape(-1).to(torch.int32).to(DEV), max_len=L)
            a = ops.cc_embed(Et, sets, 'sum', stride=L)
            b = ops.cc_embed(Et, sets, 'max', stride=L)
            agg, z = ops.mpn(Et, wp.to(DEV).requires_grad_(True), bp.to(DEV), sims.to(DEV), src=ops.SRC_GATHER, R=R, A=A,
                             ids=ids.to(DEV), sims_per_edge=True)
            w = ops.gather_rows(Et, walk.to(DEV))
            ((a * 1.5).sum() + (b * b).sum() + (agg * a).sum() + z.sum() + (w ** 2).sum()).backward()
            return E.grad.clone()
        finally:
            ops.DETERMINISTIC = True
    g1, g2, g0 = run(True), run(True), run(False)
    assert torch.equal(g1, g2)
    assert float(g1[0].abs().max()) == 0
    assert_close(g1, g0, 'deterministic vs atomics', norm_tol=1e-5)


def test_deterministic_choice_is_per_forward_not_process_wide():
    """A model states its backward form for the duration of ITS forward (ops.deterministic) and the ops record it:
    a forward run under 'atomics' keeps its atomic backward although the scope has ended and another forward ran under
    'sorted' in between -- and the other way round."""
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(3)
    E0 = torch.randn(50, 32, generator=g)
    walk = torch.randint(1, 50, (40, 6), generator=g)

    def fwd(flag):
        E = E0.to(DEV).requires_grad_(True)
        with ops.deterministic(flag):
            out = ops.gather_rows(ops.tap_table(E), walk.to(DEV))
        return E, out
    Ea, oa = fwd(False)
    Es, os_ = fwd(True)
    assert oa.grad_fn.det is False and os_.grad_fn.det is True          # recorded per op
    assert ops._det_now() is ops.DETERMINISTIC                           # the scopes have ended
    (oa ** 2).sum().backward()
    (os_ ** 2).sum().backward()
    assert_close(Ea.grad, Es.grad, 'atomics vs sorted', norm_tol=1e-5)


@pytest.mark.parametrize('R,A,view', [(1024, 1, False), (5000, 37, False), (50000, 183, False), (20000, 64, True)])
def test_column_sum_matches_float64_and_repeats(R, A, view):
    """ops.column_sum (sgnn_column_sum: row-block partials + one wavefront per column) against a float64 sum; a column
    slice of a wider matrix (row stride > A) as the head's backward hands it over; twice -> identical bits."""
    ops = _ops()
    g = torch.Generator().manual_seed(R + A)
    wide = torch.randn(R, A + (7 if view else 0), generator=g).to(DEV)
    x = wide[:, 3:3 + A] if view else wide
    out = ops.column_sum(x)
    assert_close(out, x.double().sum(0).float(), 'column sum', norm_tol=1e-6)
    assert torch.equal(out, ops.column_sum(x))


def test_contract_rows_matches_the_plain_product():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(16800, 96, generator=g).to(DEV), torch.randn(16800, 64, generator=g).to(DEV)
    ref = (a.double().t() @ b.double()).float()
    assert_close(ops.contract_rows(a, b), ref, 'contract_rows', norm_tol=1e-6)
    assert_close(ops.contract_rows(a[:1000], b[:1000]), (a[:1000].double().t() @ b[:1000].double()).float(), 'small', norm_tol=1e-5)


# ---- fused MLP head + loss (csrc/head.hip) ---------------------------------------------------------------------------------------

def _head_modules(H0, H1, H2, K, seed):
    g = torch.Generator().manual_seed(seed)
    mods = [torch.nn.Linear(H0, H1), torch.nn.Linear(H1, H2), torch.nn.Linear(H2, K)]
    for m in mods:                                     # (activations and logits of order 1: a softmax over logits of order 30 carries
        m.weight.data = torch.randn(m.weight.shape, generator=g) / m.in_features ** 0.5      # 1e-6 x 30 of relative error by itself)
        m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.3
    return mods


@pytest.mark.parametrize('B,H0,H1,H2,K', [(1, 7, 3, 2, 1), (50, 33, 16, 8, 3), (64, 615, 64, 32, 6), (1000, 566, 64, 64, 3),
                                          (4133, 130, 128, 128, 32), (9000, 70, 100, 37, 5)])
def test_fused_head_matches_torch(B, H0, H1, H2, K):
    """ops.fused_head (p = 0) == lin -> relu -> lin2 -> relu -> lin3 -> CrossEntropyLoss + accuracy of plain torch on the CPU:
    logits, loss, accuracy and every gradient (x, three weights, three biases), rows not a multiple of the 32-row chunk, widths
    not multiples of the MFMA tile, an ignored row (-100), several workgroups and several chunks per workgroup."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + H0)
    x = torch.randn(B, H0, generator=g)
    labels = torch.randint(0, K, (B,), generator=g)
    if B > 10:
        labels[3] = -100
    # (the reference in float64: two fp32 evaluations with different summation orders differ by more than the tolerance on
    # elements that are small by cancellation)
    ref_m, got_m = [m.double() for m in _head_modules(H0, H1, H2, K, 3)], [m.to(DEV) for m in _head_modules(H0, H1, H2, K, 3)]
    xr = x.double().requires_grad_(True)
    lg_r = ref_m[2](F.relu(ref_m[1](F.relu(ref_m[0](xr)))))
    loss_r = F.cross_entropy(lg_r, labels)
    loss_r.backward()
    acc_r = (lg_r.argmax(1) == labels).float().mean()
    xg = x.to(DEV).requires_grad_(True)
    lg, loss, acc = ops.fused_head(xg, got_m[0], got_m[1], got_m[2], labels.to(DEV), 0.0, None)
    loss.backward()
    assert_close(lg.detach(), lg_r.detach(), 'logits')
    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-5 * max(1.0, abs(float(loss_r.detach())))
    assert abs(float(acc) - float(acc_r)) < 1e-6
    assert_close(xg.grad, xr.grad, 'grad x', norm_tol=2e-5)
    for a, b, nm in zip(got_m, ref_m, ('lin', 'lin2', 'lin3')):
        assert_close(a.weight.grad, b.weight.grad, 'grad %s.weight' % nm, norm_tol=2e-5)
        assert_close(a.bias.grad, b.bias.grad, 'grad %s.bias' % nm, norm_tol=2e-5)
    # without labels: logits only, and a gradient that arrives through the logits
    for m in got_m:
        m.zero_grad()
    xg2 = x.to(DEV).requires_grad_(True)
    lg2, l2, a2 = ops.fused_head(xg2, got_m[0], got_m[1], got_m[2], None, 0.0, None)
    assert l2 is None and a2 is None and torch.equal(lg2, lg.detach())
    go = torch.randn(B, K, generator=g)
    (lg2 * go.to(DEV)).sum().backward()
    for m in ref_m:
        m.zero_grad()
    xr2 = x.double().requires_grad_(True)
    (ref_m[2](F.relu(ref_m[1](F.relu(ref_m[0](xr2))))) * go.double()).sum().backward()
    assert_close(xg2.grad, xr2.grad, 'grad x through logits', norm_tol=2e-5)
    assert_close(got_m[1].weight.grad, ref_m[1].weight.grad, 'grad lin2.weight through logits', norm_tol=2e-5)
    # twice the same call: bit-identical (fixed summation orders, the ticket leaves its counter at zero)
    lg3, loss3, _ = ops.fused_head(x.to(DEV), got_m[0], got_m[1], got_m[2], labels.to(DEV), 0.0, None)
    assert torch.equal(lg3, lg.detach()) and torch.equal(loss3, loss.detach())


def test_fused_head_dropout_masks():
    """p > 0: every mask element is kept with probability 1 - p and scaled by 1 / (1 - p); the step counter advances on the
    device (the next call draws other masks, the same {seed, step} the same ones); the backward gates exactly the kept,
    activated elements."""
    ops = _ops()
    B, H0, H1, H2, K, p = 3000, 40, 64, 32, 4, 0.3
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H0, generator=g).to(DEV)
    mods = [m.to(DEV) for m in _head_modules(H0, H1, H2, K, 5)]
    labels = torch.randint(0, K, (B,), generator=g).to(DEV)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device=DEV)
    saved = {}

    def run(state):
        xx = x.clone().requires_grad_(True)
        lg, loss, acc = ops.fused_head(xx, mods[0], mods[1], mods[2], labels, p, state)
        a1 = loss.grad_fn.saved_tensors[4] if hasattr(loss.grad_fn, 'saved_tensors') else None
        loss.backward()
        return lg.detach(), a1, xx.grad
    lg_a, a1_a, gx_a = run(rng)
    assert rng.tolist() == [1234, 1]
    lg_b, a1_b, _ = run(rng)
    assert rng.tolist() == [1234, 2] and not torch.equal(lg_a, lg_b)
    lg_c, a1_c, gx_c = run(torch.tensor([1234, 0], dtype=torch.int64, device=DEV))
    assert torch.equal(lg_a, lg_c) and torch.equal(gx_a, gx_c)
    z1 = torch.relu(F.linear(x, mods[0].weight, mods[0].bias))
    live = z1 > 0
    kept = (a1_a > 0) & live
    frac = float(kept.sum()) / float(live.sum())
    assert abs(frac - (1 - p)) < 0.01, frac
    assert_close(a1_a[kept], z1[kept] / (1 - p), 'kept activations are scaled', norm_tol=1e-6)
    assert float(a1_a[~kept].abs().max()) == 0.0
    # eval-style call: p = 0 leaves the state alone
    ops.fused_head(x, mods[0], mods[1], mods[2], labels, 0.0, None)
    assert rng.tolist() == [1234, 2]


@pytest.mark.parametrize('R', [5, 300, 5000, 40000])
def test_contract_rows_many_matches_torch(R):
    """a^T b on the matrix cores for tall operands (block partials + ordered reduction), several jobs per launch, widths that are
    not multiples of 32, a strided operand (a column slice)."""
    ops = _ops()
    g = torch.Generator().manual_seed(R)
    shapes = [(64, 566), (3, 5), (256, 64), (100, 33), (32, 32), (1, 1), (130, 70), (64, 64), (17, 200)]
    pairs, want = [], []
    for M, N in shapes:
        a, b = torch.randn(R, M, generator=g), torch.randn(R, N + 3, generator=g)
        pairs.append((a.to(DEV), b.to(DEV)[:, 1:N + 1]))
        want.append(a.double().t() @ b[:, 1:N + 1].double())
    got = ops.contract_rows_many(pairs)
    assert len(got) == len(shapes)
    for (M, N), o, w in zip(shapes, got, want):
        assert tuple(o.shape) == (M, N)
        assert_close(o, w.float(), 'a^T b (%d x %d over %d rows)' % (M, N, R), norm_tol=2e-5)
    again = ops.contract_rows_many(pairs)
    assert all(torch.equal(a, b) for a, b in zip(got, again))


def test_device_side_batch_index_out_of_range_is_reported():
    """a device-resident batch index outside its split cannot raise inside the gather kernel; it yields zero rows and
    sets a flag that ops.poll_index_errors turns into an IndexError."""
    ops = _ops()
    a = torch.arange(40, dtype=torch.float32, device=DEV).view(10, 4)
    b = torch.arange(30, dtype=torch.int64, device=DEV).view(10, 3)
    ops.poll_index_errors(block=True)
    good = ops.index_rows_many([a, b], torch.tensor([9, 0, 3], device=DEV))
    assert torch.equal(good[0], a[[9, 0, 3]]) and torch.equal(good[1], b[[9, 0, 3]])
    ops.poll_index_errors(block=True)
    bad = ops.index_rows_many([a, b], torch.tensor([2, 10, -1], device=DEV))
    assert torch.equal(bad[0][0], a[2]) and float(bad[0][1:].abs().max()) == 0 and int(bad[1][1:].abs().max()) == 0
    with pytest.raises(IndexError):
        ops.poll_index_errors(block=True)
    ops.poll_index_errors(block=True)                 # reported once


@pytest.mark.parametrize('B,C', [(7, 3), (300, 2), (5000, 1)])
def test_readout_pieces_with_scores_in_kernel(B, C):
    """Read-out pieces whose scores s = X wp (0 for a PAD anchor) are computed inside ops.subgraph_embedding's own launches
    (sgnn_readout_many_fwd / _bwd: all pieces of a step in two launches each way) against the same pieces with the scores made
    by torch beforehand (one launch pair per piece + the autograd of s = X wp): embedding bit-equal, gradients of X, wp, bp and
    of the tensor pieces equal; two anchor widths in one call (two launch groups), a piece without anchors (all-zero
    similarities), nine pieces (more than one launch of eight)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + C)
    R = B * C
    mask = (torch.rand(B, C, generator=g) > 0.3).reshape(-1).to(torch.uint8).to(DEV)
    specs = [(11, 16, True), (40, 16, False), (5, 16, True), (183, 16, True), (7, 8, False), (42, 16, False), (3, 16, True), (9, 16, False),
             (21, 16, True)]
    data = []
    for A, D, with_ids in specs:
        wide = torch.rand(R, A + 4, generator=g)
        col = torch.randperm(A + 4, generator=g)[:A]
        ids = torch.randint(0, 3, (A,), generator=g) if with_ids else None
        data.append((wide.to(DEV), col.to(DEV), torch.randn(A, D, generator=g), torch.randn(D, generator=g), torch.randn(1, generator=g) * 0.3,
                     ids.to(DEV) if ids is not None else None, A))
    x0 = torch.randn(B, C, 6, generator=g)
    b0 = torch.tensor([0.25])
    H = 6 + sum(A for A, _, _ in specs) + 4
    go = torch.randn(B, H, generator=g).to(DEV)

    def run(in_kernel):
        X0 = x0.to(DEV).clone().requires_grad_(True)
        B0 = b0.to(DEV).clone().requires_grad_(True)
        leaves, pieces = [X0, B0], [X0]
        for wide, col, X, wp, bp, ids, A in data:
            Xl, wl, bl = [t.to(DEV).clone().requires_grad_(True) for t in (X, wp, bp)]
            leaves += [Xl, wl, bl]
            if in_kernel:
                pieces.append(ops.ReadoutPiece(wide, col, None, bl, A, mask, R, X=Xl, wp=wl, ids=ids))
            else:
                s = Xl @ wl
                if ids is not None:
                    s = s * (ids != 0).to(s.dtype)
                pieces.append(ops.ReadoutPiece(wide, col, s, bl, A, mask, R))
        pieces.append(ops.ReadoutPiece(None, None, None if in_kernel else torch.zeros(4, device=DEV), B0, 4, mask, R))
        out = ops.subgraph_embedding(pieces, mask, B, C)
        (out * go).sum().backward()
        return out.detach(), [t.grad for t in leaves]
    out_a, g_a = run(True)
    out_b, g_b = run(False)
    assert_close(out_a, out_b, 'embedding', norm_tol=2e-6)       # (the scores' dot products are summed in another order than the library's)
    for i, (a, b) in enumerate(zip(g_a, g_b)):
        assert_close(a, b, 'gradient of leaf %d' % i, norm_tol=1e-5)
    out_c, g_c = run(True)
    assert torch.equal(out_a, out_c) and all(torch.equal(a, b) for a, b in zip(g_a, g_c))


@pytest.mark.parametrize('layers,agg,D,W', [(1, 'last', 64, 5), (2, 'last', 128, 5), (2, 'sum', 32, 3), (1, 'sum', 64, 1)])
def test_walk_aggregator_in_library_launches_matches_float64(layers, agg, D, W):
    """aps.aggregate_structure_anchor_patch (aps:413-433) through LSTM.forward_walks -- the embedding lookup as the operand load of
    the first layer's input projection (sgnn_rows_gemm), weight / bias gradients of both directions as jobs of ONE contraction
    launch, dx by sgnn_rows_gemm_nt into the table's combined scatter, the tail (last step or sum, Linear, sum over a patch's
    walks) one launch each way -- against the same module in float64 on the CPU: the anchor embeddings, the gradient of the
    table (PAD row untouched, rows of repeated ids accumulated) and of every LSTM / Linear parameter; bias gradients in separate
    memory; and the unfused route (hparams['fused_forward'] = False) gives the same embeddings."""
    import copy
    from subgnn_amd.SubGNN import LSTM
    from subgnn_amd import anchor_patch_samplers as aps
    ops = _ops()
    torch.manual_seed(layers * 10 + D)
    n, T, N = 21, 7, 300
    m = LSTM(D, D, dropout=0.0, num_layers=layers, aggregator=agg).to(DEV)
    g = torch.Generator().manual_seed(W + D)
    table = torch.randn(N + 1, D, generator=g) * 0.5
    table[0] = 0
    walks = torch.randint(0, N + 1, (n, W, T), generator=g)
    walks[0, 0, 3:] = 0                                             # PAD steps
    walks[1] = walks[2]                                             # repeated ids: their rows' gradients add up
    hp = {'n_triangular_walks': W, 'random_walk_len': T, 'node_embed_size': D, 'fused_forward': True}
    go = torch.randn(n, D, generator=g)

    class Emb:                                                      # (node_matrix: only .weight is read)
        pass
    E = table.to(DEV).requires_grad_(True)
    emb = Emb()
    emb.weight = E
    tapped = ops.tap_table(E)
    wd = walks.to(DEV)
    ops.presort_ids(wd, N)
    X = aps.aggregate_structure_anchor_patch(hp, None, m, emb, wd, wd, None, torch.device(DEV), table=tapped)
    (X * go.to(DEV)).sum().backward()
    r = copy.deepcopy(m).cpu().double()
    Er = table.double().requires_grad_(True)
    xr = Er[walks.view(n * W, T)]
    out, _ = r.lstm(xr)
    Xr = r.linear(out[:, -1, :] if agg == 'last' else out.sum(dim=1)).view(n, W, -1).sum(1)
    (Xr * go.double()).sum().backward()
    assert_close(X, Xr, 'anchor embeddings')
    assert float(E.grad[0].abs().max()) == 0.0                     # PAD row
    gref = Er.grad.clone()
    gref[0] = 0
    assert_close(E.grad, gref, 'd table', norm_tol=1e-5)
    named = dict(m.named_parameters())
    for nm, q in r.named_parameters():
        assert_close(named[nm].grad, q.grad, 'd ' + nm, norm_tol=2e-5)
    biases = [k for k in named if 'bias' in k and k.startswith('lstm.')]
    spans = sorted((named[k].grad.data_ptr(), named[k].grad.data_ptr() + named[k].grad.numel() * 4) for k in biases)
    assert all(a[1] <= b[0] for a, b in zip(spans[:-1], spans[1:])), 'bias gradients overlap in memory'
    with torch.no_grad():
        X2 = aps.aggregate_structure_anchor_patch(dict(hp, fused_forward=False), None, m, emb, wd, wd, None, torch.device(DEV), table=E.detach())
    assert_close(X2, X.detach(), 'unfused route', norm_tol=1e-5)
