"""-m gpu: every BASELINE.json configuration other than the benchmark shard (configs[3], see
test_gpu_fullsize.py) at its own size, stage by stage against the oracle on sampled rows:

  density_n  configs[0]  DENSITY recipe ~1k nodes, neighbourhood channel only, 5 layers (H1)
  ppi_bp     configs[1]  PPI-BP stand-in, all three channels, batch of 64, ~7 components (H2)
  hpo_metab  configs[2]  HPO-METAB stand-in, 4 layers, 360 structure patches (DTW stressed)
  em_user    configs[4]  EM-USER stand-in: k = 2 border, ~50 components of up to 60 nodes, sparse
                         prepare (multi-component P-internal BFS), general DTW kernel (rows > 32),
                         fp16-stored table

The stand-ins (subgnn_amd/standins.py) are written in the reference's file formats and read back by
the drop-in constructor; the oracle reads the same edge list with its own networkx-order reader
(pinned by golden g1).  Integer stages are compared bit for bit, the training step (logits, loss,
gradients) within 1e-4 relative against oracle/float_half.py fed with the product's prepared state.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, oracle_inputs as _oracle_inputs
from oracle import cbind, float_half as FH, graph as OG, integer_half as IH, tape as OT
from oracle.integer_half import bfs_hops_numpy as _bfs_hops_numpy

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SLOTS = ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B')


@pytest.fixture(scope='module', params=['density_n', 'ppi_bp', 'hpo_metab', 'em_user'])
def cfg(request, tmp_path_factory):
    from subgnn_amd import standins
    name = request.param
    root = tmp_path_factory.mktemp(name)
    # dropout off: the training step is then a deterministic function the oracle can restate
    # (ff_attn off here: the EM-USER preset's attention read-out on half operands has its own test below, with the tolerance
    # half-rounded scores need -- test_em_user_with_ff_attn_half_mfma_scores)
    model, d, n_edges = standins.build_model(root, name, {'lin_dropout': 0.0, 'lstm_dropout': 0.0, 'ff_attn': False}, torch.device(DEV))
    G = OG.read_edgelist(os.path.join(d, 'edge_list.txt'))
    rowptr, col = G.csr()
    with open(os.path.join(d, 'degree_sequence.txt')) as f:
        dd = json.load(f)
    full = np.zeros(len(rowptr) - 1, dtype=np.int32)
    for k, v in dd.items():
        full[int(k) + 1] = int(v)
    out = dict(name=name, model=model, dir=d, G=G, rowptr=rowptr, col=col, full_degree=full,
               sparse=standins.PRESETS[name]['sparse'], ego=os.path.exists(os.path.join(d, 'ego_graphs.txt')),
               rng=np.random.default_rng(11))
    yield out
    out.clear()
    torch.cuda.empty_cache()


def _comp(cc_row):
    return [int(v) for v in cc_row if v != 0]


def _sample_rows(cc, rng, n):
    """(s, c) of real component rows, the largest one always included."""
    real = np.argwhere(cc[:, :, 0] != 0)
    pick = real[rng.choice(len(real), min(n, len(real)), replace=False)]
    sizes = (cc != 0).sum(2)
    big = np.unravel_index(np.argmax(sizes), sizes.shape)
    return [(int(big[0]), int(big[1]))] + [(int(p[0]), int(p[1])) for p in pick]


def test_graph_order_and_components(cfg):
    m, G = cfg['model'], cfg['G']
    g = m.networkx_graph
    # the product's CSR (vectorised numpy reader) is the oracle's networkx-order graph (python dict reader)
    assert np.array_equal(g.rowptr.cpu().numpy(), cfg['rowptr']) and np.array_equal(g.col.cpu().numpy()[:g.nnz], cfg['col'])
    assert np.array_equal(g.node_order.cpu().numpy(), np.asarray(G.node_order, dtype=np.int32))
    for sp in ('train', 'val'):
        cc = getattr(m, sp + '_cc_ids').cpu().numpy()
        subs = getattr(m, sp + '_sub_G')
        assert cc.shape[0] == len(subs)
        for s in cfg['rng'].choice(len(subs), min(40, len(subs)), replace=False):
            comps = IH.connected_components(G, subs[s])
            assert len(comps) <= cc.shape[1]
            for c in range(cc.shape[1]):
                assert _comp(cc[s, c]) == (comps[c] if c < len(comps) else [])
        # the padded shape is the tight one
        n_cc = (cc[:, :, 0] != 0).sum(1)
        assert n_cc.max() == cc.shape[1] and (cc != 0).sum(2).max() == cc.shape[2]
    if cfg['name'] == 'em_user':
        assert m.train_cc_ids.shape[1] >= 20 and m.train_cc_ids.shape[2] > 32     # many components, long rows
    if cfg['name'] == 'ppi_bp':
        assert m.train_cc_ids.shape[1] >= 5


def _border_sizes_numpy(rowptr, col, cc, ego):
    """|border| of every component row for the 1-hop cases, vectorised (the oracle's python sets
    would take minutes on 4k rows of a degree-440 graph) -> (entries other than id 0, all entries).
    In ego-dict mode the shifted ids can contain 0, which the padded matrix cannot tell from PAD
    (neither can the reference) but which still counts towards the padded width."""
    S, C, _ = cc.shape
    out = np.zeros((S, C), dtype=np.int64)
    out0 = np.zeros((S, C), dtype=np.int64)
    for s in range(S):
        for c in range(C):
            comp = cc[s, c][cc[s, c] != 0]
            if len(comp) == 0:
                continue
            nb = np.concatenate([col[rowptr[v]:rowptr[v + 1]] for v in comp]).astype(np.int64)
            nb = np.unique(nb - 1 if ego else nb)
            nb = np.setdiff1d(nb, comp)
            out[s, c], out0[s, c] = len(nb[nb != 0]), len(nb)
    return out, out0


def _oracle_nanchors(ids, width, n_slots, seed, stream, r):
    real = np.sort(np.asarray([v for v in ids if v != 0], dtype=np.int64))
    n = len(real)
    out = []
    for i in range(n_slots):
        k = OT.nanchor_pick(seed, stream, r * n_slots + i, n, n < width)
        out.append(0 if k < 0 else int(real[k]))
    return out


def test_border_sets_and_neighbourhood_anchors(cfg):
    m, G, rng = cfg['model'], cfg['G'], cfg['rng']
    hp = m.hparams
    if not hp['use_neighborhood']:
        pytest.skip('neighbourhood channel off')
    seed, k = int(hp['seed']), hp['neigh_sample_border_size']
    cc = m.train_cc_ids.cpu().numpy()
    S, C, L = cc.shape
    rows = _sample_rows(cc, rng, 8 if cfg['sparse'] else 30)
    if not cfg['sparse']:
        nb = m.train_N_border.cpu().numpy()
        Lb = nb.shape[2]
        if k == 1 or cfg['ego']:
            sizes, sizes0 = _border_sizes_numpy(cfg['rowptr'], cfg['col'], cc, cfg['ego'])
            assert np.array_equal((nb != 0).sum(2), sizes) and Lb == sizes0.max()
        for s, c in rows:
            want = IH.component_border_set(G, cc[s, c], k, ego_dict_mode=cfg['ego'])
            assert [int(v) for v in nb[s, c] if v != 0] == sorted(v for v in want if v != 0)
        for l in range(hp['n_layers']):
            ni, nbo = m.anchors_neigh_int['train'][l].cpu().numpy(), m.anchors_neigh_border['train'][l].cpu().numpy()
            for s, c in rows:
                r = s * C + c
                assert ni[s, c].tolist() == _oracle_nanchors(cc[s, c], L, hp['n_anchor_patches_N_in'], seed,
                                                             OT.stream_id(OT.STREAM_N_INT, 'train', l), r)
                assert nbo[s, c].tolist() == _oracle_nanchors(nb[s, c], Lb, hp['n_anchor_patches_N_out'], seed,
                                                              OT.stream_id(OT.STREAM_N_BOR, 'train', l), r)
        return
    # sparse path: the border is never materialised; anchors are rank queries on the BFS bitmap and their
    # similarity is the hop level
    from subgnn_amd import ops
    sets = ops.Ragged.from_padded(m.train_cc_ids.reshape(S * C, L))
    counts = ops.khop_border(m.networkx_graph, sets, k).lengths.cpu().numpy()
    width = int(counts.max())
    sims = m.train_neigh_pos_similarities
    for l in range(hp['n_layers']):
        ni, nbo = m.anchors_neigh_int['train'][l].cpu().numpy(), m.anchors_neigh_border['train'][l].cpu().numpy()
        w = sims[('N', 'out', l)].cpu().numpy()
        assert isinstance(sims[('N', 'in', l)], ops.ZeroSims)
        for s, c in rows:
            r = s * C + c
            levels = IH.border_hop_levels(G, cc[s, c], k)
            assert counts[r] == len(levels)
            assert ni[s, c].tolist() == _oracle_nanchors(cc[s, c], L, hp['n_anchor_patches_N_in'], seed,
                                                         OT.stream_id(OT.STREAM_N_INT, 'train', l), r)
            want = _oracle_nanchors(sorted(levels), width, hp['n_anchor_patches_N_out'], seed,
                                    OT.stream_id(OT.STREAM_N_BOR, 'train', l), r)
            assert nbo[s, c].tolist() == want
            assert w[s, c].tolist() == [float(levels[a]) if a else 0.0 for a in want]
        pad = cc[:, :, 0] == 0
        assert (nbo[pad] == 0).all() and (w[pad] == 0).all()
    if k == 2:
        assert set(np.unique(w).tolist()) == {0.0, 1.0, 2.0}


def test_position_anchors_and_similarities(cfg):
    m, G, rng = cfg['model'], cfg['G'], cfg['rng']
    hp = m.hparams
    cc = m.train_cc_ids.cpu().numpy()
    S, C, L = cc.shape
    n = G.max_id()
    seed = int(hp['seed'])
    if hp['use_position']:
        for l in range(hp['n_layers']):
            assert m.anchors_pos_ext[l].cpu().numpy().tolist() == \
                IH.position_anchors_border(G, hp['n_anchor_patches_pos_out'], seed, l).tolist()
            pin = m.anchors_pos_int['train'][l].cpu().numpy()
            st = OT.stream_id(OT.STREAM_P_INT, 'train', l)
            for s in rng.choice(S, min(40, S), replace=False):
                sg = m.train_sub_G[s]
                assert pin[s].tolist() == [sg[OT.choice_index(seed, st, int(s), j, len(sg))]
                                           for j in range(hp['n_anchor_patches_pos_in'])]
    if not cfg['sparse']:
        if not (hp['use_position'] or hp['use_neighborhood']):
            return
        apsp = np.load(os.path.join(cfg['dir'], 'shortest_path_matrix.npy'), mmap_mode='r')
        for src in rng.choice(n, 3, replace=False) + 1:                       # the GPU metric precompute itself
            d = _bfs_hops_numpy(cfg['rowptr'], cfg['col'], int(src), n)[1:].astype(np.float64)
            d[d == 255] = 0
            assert np.array_equal(np.asarray(apsp[src - 1]), d)
        slab = m.train_neigh_pos_similarities
        for s, c in _sample_rows(cc, rng, 12):
            comp = np.asarray(_comp(cc[s, c]))
            want = np.min(np.asarray(apsp[comp - 1, :]), axis=0).astype(np.float32)
            assert np.array_equal(slab[s, c].cpu().numpy(), want)
        pad = torch.from_numpy(cc[:, :, 0] == 0).to(slab.device)
        if bool(pad.any()):
            assert float(slab[pad].abs().max()) == 0
        return
    if not hp['use_position']:
        return
    sims = m.train_neigh_pos_similarities
    real = cc[:, :, 0] != 0
    for l in range(hp['n_layers']):
        ext = m.anchors_pos_ext[l].cpu().numpy()
        w = sims[('P', 'out', l)].cpu().numpy()
        for a in rng.choice(len(ext), 5, replace=False):
            d = _bfs_hops_numpy(cfg['rowptr'], cfg['col'], int(ext[a]), n).astype(np.float32)
            d[d == 255] = 0
            want = np.zeros((S, C), dtype=np.float32)
            for s, c in np.argwhere(real):
                want[s, c] = d[cc[s, c][cc[s, c] != 0]].min()
            assert np.array_equal(w[:, :, a], want)
        pin = m.anchors_pos_int['train'][l].cpu().numpy()
        wi = sims[('P', 'in', l)]
        if C == 1:
            continue
        wi = wi.cpu().numpy()
        assert wi.shape == (S, C, pin.shape[1])
        for s in rng.choice(S, 3, replace=False):
            for a in rng.choice(pin.shape[1], 3, replace=False):
                d = _bfs_hops_numpy(cfg['rowptr'], cfg['col'], int(pin[s, a]), n).astype(np.float32)
                d[d == 255] = 0
                for c in range(C):
                    comp = cc[s, c][cc[s, c] != 0]
                    assert wi[s, c, a] == (d[comp].min() if len(comp) else 0.0)
        # an anchor that lies in component c is at distance 0 from it
        own = (torch.from_numpy(pin).unsqueeze(1).unsqueeze(-1) == torch.from_numpy(cc).unsqueeze(2)).any(-1).numpy()
        assert (wi[own] == 0).all()


def test_structure_patches_walks_and_picks(cfg):
    m, G, rng = cfg['model'], cfg['G'], cfg['rng']
    hp = m.hparams
    if not hp['use_structure']:
        pytest.skip('structure channel off')
    seed = int(hp['seed'])
    sa = m.structure_anchors.cpu().numpy()
    P = hp['max_sim_epochs'] * hp['n_anchor_patches_structure'] * hp['n_layers']
    assert sa.shape[0] == P and (sa != 0).sum(1).max() == sa.shape[1]
    iw, bw = m.int_structure_anchor_random_walks.cpu().numpy(), m.bor_structure_anchor_random_walks.cpu().numpy()
    W, Tn = hp['n_triangular_walks'], hp['random_walk_len']
    assert iw.shape == (P, W, Tn) and bw.shape == (P, W, Tn)
    for p in rng.choice(P, 10, replace=False):
        walk = IH.triangular_walk(G, hp['sample_walk_len'], hp['rw_beta'],
                                  IH._Draws(seed, OT.stream_id(OT.STREAM_STRUCT_PATCH), int(p)), 'graph')
        assert _comp(sa[p]) == walk
        nodes = IH.patch_unique_nodes(sa[p])
        inb = IH.patch_in_border_nodes(G, nodes)
        for w in range(W):
            a = IH.triangular_walk(G, Tn, hp['rw_beta'], IH._Draws(seed, OT.stream_id(OT.STREAM_WALK_INT), int(p) * W + w),
                                   'inside', nodes)
            assert _comp(iw[p, w]) == a and (iw[p, w][len(a):] == 0).all()
            b = IH.triangular_walk(G, Tn, hp['rw_beta'], IH._Draws(seed, OT.stream_id(OT.STREAM_WALK_BOR), int(p) * W + w),
                                   'border', nodes, inb)
            assert _comp(bw[p, w]) == b and (bw[p, w][len(b):] == 0).all()
    for l in range(hp['n_layers']):
        patches, idx, irw, brw = m.anchors_structure[l]
        idx = [int(i) for i in (idx.tolist() if torch.is_tensor(idx) else idx)]
        assert idx == IH.structure_anchor_indices(P, hp['n_anchor_patches_structure'], seed, l)
        assert np.array_equal(patches.cpu().numpy(), sa[idx]) and np.array_equal(irw.cpu().numpy(), iw[idx])
        assert np.array_equal(brw.cpu().numpy(), bw[idx])


def test_degree_sequences_and_dtw(cfg):
    m, rng = cfg['model'], cfg['rng']
    hp = m.hparams
    if not hp['use_structure']:
        pytest.skip('structure channel off')
    from subgnn_amd import gamma, ops
    rowptr, col, full = cfg['rowptr'], cfg['col'], cfg['full_degree']
    g = m.networkx_graph
    assert g.full_degree is not None and np.array_equal(g.full_degree.cpu().numpy(), full)
    cc = m.train_cc_ids.cpu().numpy()
    S, C, L = cc.shape
    sa = m.structure_anchors.cpu().numpy()
    rows = _sample_rows(cc, rng, 24)
    pcols = rng.choice(sa.shape[0], 12, replace=False)
    xp, xf = cbind.ragged([_comp(cc[s, c]) for s, c in rows])
    yp, yf = cbind.ragged([_comp(sa[p]) for p in pcols])
    xi, xe = cbind.degree_sequence(rowptr, col, full, xp, xf, True)
    yi, ye = cbind.degree_sequence(rowptr, col, full, yp, yf, True)
    # the degree-sequence kernel itself on the sampled sets (both the wave and, for rows > 64, the block form)
    sets = ops.Ragged.from_lists([_comp(cc[s, c]) for s, c in rows], g.device)
    for internal, want in ((True, xi), (False, xe)):
        _, vals = gamma.degree_sequences(g, sets, internal, use_degree_dict=True)
        assert np.array_equal(vals.cpu().numpy()[:len(want)], want)
    rsel = torch.tensor([s * C + c for s, c in rows], device=g.device)
    csel = torch.from_numpy(pcols).to(g.device)
    for sims, x, y in ((m.train_int_struc_similarities, xi, yi), (m.train_bor_struc_similarities, xe, ye)):
        assert tuple(sims.shape) == (S, C, sa.shape[0])
        got = sims.reshape(S * C, -1)[rsel][:, csel].cpu().numpy()
        want = cbind.fastdtw_sim(xp, x, yp, y, m.hparams['dtw_tie_order'])
        assert np.array_equal(got, want)
        pad = torch.from_numpy(cc[:, :, 0] == 0).to(sims.device)
        if bool(pad.any()):
            assert float(sims[pad].abs().max()) == 0
        assert float(sims[~pad].min()) > 0 and float(sims.max()) <= 1
    if cfg['name'] == 'em_user':
        assert L > 32            # the general DTW kernel (register kernel covers x rows <= 32)


def test_training_step_against_oracle(cfg):
    """One batch of the configuration's own batch size: logits, loss and every gradient against the
    dense torch-CPU restatement fed with the same prepared state."""
    m, rng = cfg['model'], cfg['rng']
    hp = m.hparams
    B = min(hp['batch_size'], len(m.train_sub_G))
    idx = torch.from_numpy(np.sort(rng.choice(len(m.train_sub_G), B, replace=False)))
    m.train()
    m.zero_grad(set_to_none=True)
    batch = m.make_batch('train', idx)
    out = m.training_step(batch, 0)
    logits = m._forward_batch('train', batch)
    m.backward(None, out['loss'], None, 0)
    params, anchors, ob, ccp = _oracle_inputs(m, batch, idx)
    ref_logits = FH.forward(params, hp, 'train', ob, anchors, ccp)
    labels = batch['label'].cpu()
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, labels)
    ref_loss.backward()
    assert logits.shape == (B, m.num_classes)
    assert_close(logits, ref_logits, 'logits')
    assert_close(out['loss'], ref_loss, 'loss')
    checked = 0
    for k, p in m.named_parameters():
        ref = params[k].grad
        if p.grad is None:
            assert ref is None or float(ref.abs().max()) == 0, k
            continue
        if ref is None:
            assert float(p.grad.abs().max()) == 0, k
            continue
        assert_close(p.grad, ref, 'grad ' + k)
        checked += 1
    assert checked >= 8
    assert float(m.node_embeddings.weight.grad.abs().max()) > 0 and float(m.node_embeddings.weight.grad[0].abs().max()) == 0
    # and the optimizer step the caller takes next moves the parameters
    opt = m.configure_optimizers()
    before = m.lin.weight.detach().clone()
    torch.nn.utils.clip_grad_norm_(m.parameters(), hp['grad_clip'])
    opt.step()
    assert not torch.equal(before, m.lin.weight.detach())


def test_em_user_with_ff_attn_half_mfma_scores(tmp_path_factory):
    """BASELINE configs[4] as worded -- "EM-USER (large components, border channel heavy), fp16 embeddings with MFMA
    attention scores": the EM-USER stand-in with ``ff_attn`` on and the fp16-stored table (hid_dim 579, >= 20 components
    per subgraph).  (1) a training step at the configuration's batch size (32 subgraphs: under 2048 component rows the
    score contraction is the library GEMM on half-rounded operands + the fused epilogue); (2) the whole train split as
    one batch (>= 5k component rows: the hand-written v_mfma_f32_32x32x16_f16 kernel -- counted).  Logits and loss against
    the oracle with the same operand rounding within 1e-4 element-wise; gradients against the fp32 restatement within
    half precision."""
    from subgnn_amd import standins, _lib, hotpath
    root = tmp_path_factory.mktemp('em_user_attn')
    m, d, _ = standins.build_model(root, 'em_user', {'lin_dropout': 0.0, 'lstm_dropout': 0.0, 'ff_attn': True},
                                   torch.device(DEV))
    hp = m.hparams
    assert hp['ff_attn'] and m.attention.half_operands and m.hid_dim <= 640
    C = m.train_cc_ids.shape[1]
    assert C >= 20 and m.train_cc_ids.shape[2] > 32                        # many components, long rows (the stand-in's pieces merge to >= 20)
    lib = _lib.load()
    calls = {'f16': 0, 'epi': 0}
    real_f16, real_epi = lib.sgnn_attn_scores_fwd_f16, lib.sgnn_attn_scores_epilogue

    def f16(*a):
        calls['f16'] += 1
        return real_f16(*a)

    def epi(*a):
        calls['epi'] += 1
        return real_epi(*a)
    lib.sgnn_attn_scores_fwd_f16, lib.sgnn_attn_scores_epilogue = f16, epi
    try:
        rng = np.random.default_rng(5)
        # (1) the configuration's batch
        B = hp['batch_size']
        idx = torch.from_numpy(np.sort(rng.choice(len(m.train_sub_G), B, replace=False)))
        m.train()
        m.zero_grad(set_to_none=True)
        batch = m.make_batch('train', idx)
        out = m.training_step(batch, 0)
        m.backward(None, out['loss'], None, 0)
        assert calls == {'f16': 0, 'epi': 1} and B * batch['cc_ids'].shape[1] < 2048
        logits = m._forward_batch('train', batch)
        params, anchors, ob, ccp = _oracle_inputs(m, batch, idx)
        ref_logits = FH.forward(params, hp, 'train', ob, anchors, ccp)
        ref_loss = torch.nn.functional.cross_entropy(ref_logits, batch['label'].cpu())
        assert_close(logits, ref_logits, 'logits (batch of 32, half-rounded score operands)')
        assert_close(out['loss'], ref_loss, 'loss')
        hp32 = dict(hp, embedding_dtype='fp32')
        params32, _, _, ccp32 = _oracle_inputs(m, batch, idx)
        l32 = torch.nn.functional.cross_entropy(FH.forward(params32, hp32, 'train', ob, anchors, ccp32), batch['label'].cpu())
        l32.backward()
        for k in ('attention._u_matrix', 'attention._w_matrix', 'attention._v_vector', 'attn_vector', 'lin.weight'):
            assert_close(dict(m.named_parameters())[k].grad, params32[k].grad, 'grad ' + k, 5e-2)      # (the softmax weights come from half-rounded scores)
        # (2) the whole split as one batch: the matrix-core kernel
        calls.update(f16=0, epi=0)
        full = hotpath.full_split_batch(m, 'train')
        rows = full['cc_ids'].shape[0] * full['cc_ids'].shape[1]
        assert rows >= 2048
        m.eval()
        with torch.no_grad():
            got = m._forward_batch('train', full)
        assert calls == {'f16': 1, 'epi': 0}
        all_idx = torch.arange(full['cc_ids'].shape[0])
        params, anchors, ob, ccp = _oracle_inputs(m, full, all_idx)
        with torch.no_grad():
            want = FH.forward(params, hp, 'train', ob, anchors, ccp)
        assert_close(got, want, 'logits (whole split, v_mfma_f32_32x32x16_f16 scores)')
    finally:
        lib.sgnn_attn_scores_fwd_f16, lib.sgnn_attn_scores_epilogue = real_f16, real_epi
        torch.cuda.empty_cache()
