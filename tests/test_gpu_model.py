"""-m gpu: the drop-in SubGNN module against the reference goldens.

  * prepare_data stages produced by the HIP kernels, compared with the reference's outputs
    (as sets where the reference's order is CPython-set order);
  * full forward / loss / backward parity (golden g11) given the reference's own prepared state
    (layered parity: same stage inputs), through BOTH the fused path and the reference-shaped
    get_anchor_patches + SG_MPN path.
"""
import json

import numpy as np
import pytest
import torch

from helpers import T, assert_close, write_dataset_from_golden, G11_VARIANTS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(golden, tmp_path, hp_over=None):
    from subgnn_amd import config
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    name = write_dataset_from_golden(golden, tmp_path)
    config.PROJECT_ROOT = tmp_path
    hp = dict(golden.hp)
    hp['seed'] = golden.seed
    if hp_over:
        hp.update(hp_over)
    torch.manual_seed(0)
    return SubGNN(hp, **dataset_paths(name))


def _rows_by_set(cc):
    """(S,C,L) -> {(s, frozenset(component)): (s, c)}"""
    out = {}
    for s in range(cc.shape[0]):
        for c in range(cc.shape[1]):
            st = frozenset(int(v) for v in cc[s, c] if v != 0)
            if st:
                out[(s, st)] = (s, c)
    return out


def test_prepare_data_stages(golden, tmp_path):
    m = _model(golden, tmp_path)
    m.prepare_data()
    hp = golden.hp
    for sp in ('train', 'val'):
        mine = getattr(m, sp + '_cc_ids').cpu().numpy()
        ref = golden['g2_cc_ids_' + sp]
        a, b = _rows_by_set(mine), _rows_by_set(ref)
        assert set(a) == set(b)                                              # g2
        nb_m, nb_r = getattr(m, sp + '_N_border').cpu().numpy(), golden['g3_border_' + sp]
        np_m, np_r = getattr(m, sp + '_neigh_pos_similarities').cpu().numpy(), golden['g4_np_sim_' + sp]
        is_m, is_r = getattr(m, sp + '_int_struc_similarities').cpu().numpy(), golden['g7_int_struc_sim_' + sp]
        bs_m, bs_r = getattr(m, sp + '_bor_struc_similarities').cpu().numpy(), golden['g7_bor_struc_sim_' + sp]
        for key, (s, c) in a.items():
            s2, c2 = b[key]
            assert sorted(v for v in nb_m[s, c] if v != 0) == sorted(v for v in nb_r[s2, c2] if v != 0)   # g3
            assert np.array_equal(np_m[s, c], np_r[s2, c2])                  # g4
            assert np.array_equal(is_m[s, c], is_r[s2, c2])                  # g7 (provisional pin)
            assert np.array_equal(bs_m[s, c], bs_r[s2, c2])
        # padded component rows carry PAD similarities
        pad = mine[:, :, 0] == 0
        assert (np_m[pad] == 0).all() and (is_m[pad] == 0).all()
        for l in range(hp['n_layers']):
            assert np.array_equal(m.anchors_pos_int[sp][l].cpu().numpy(), golden['g8_P_int_%s_%d' % (sp, l)])
    assert np.array_equal(m.structure_anchors.cpu().numpy(), golden['g5_structure_anchors'])   # g5, bit exact
    for l in range(hp['n_layers']):
        assert np.array_equal(m.anchors_pos_ext[l].cpu().numpy(), golden['g8_P_ext_%d' % l])
        assert m.anchors_structure[l][1] == [int(i) for i in golden['g8_S_idx_%d' % l]]
    # walks: every walk stays inside its patch (inside) / starts on the patch border (border)
    sa = m.structure_anchors.cpu().numpy()
    iw = m.int_structure_anchor_random_walks.cpu().numpy()
    for p in range(sa.shape[0]):
        assert set(iw[p].reshape(-1).tolist()) - {0} <= set(sa[p].tolist())
    # one training step runs end to end and yields finite numbers
    batch = next(iter(m.train_dataloader()))
    out = m.training_step(batch, 0)
    m.backward(None, out['loss'], None, 0)
    assert torch.isfinite(out['loss'])
    assert m.node_embeddings.weight.grad is not None and float(m.node_embeddings.weight.grad[0].abs().max()) == 0


def _inject(m, g, t, hp):
    d = lambda x: T(x).to(DEV)
    L = hp['n_layers']
    m.train_cc_ids = d(g[t + 'cc_ids_train'])
    m.train_N_border = d(g[t + 'border_train'])
    m.train_neigh_pos_similarities = d(g[t + 'np_sim_train'])
    m.train_int_struc_similarities = d(g[t + 'int_sim_train'])
    m.train_bor_struc_similarities = d(g[t + 'bor_sim_train'])
    m.anchors_neigh_int = {'train': {l: d(g[t + 'N_int_train_%d' % l]) for l in range(L)}}
    m.anchors_neigh_border = {'train': {l: d(g[t + 'N_bor_train_%d' % l]) for l in range(L)}}
    m.anchors_pos_int = {'train': {l: d(g[t + 'P_int_train_%d' % l]) for l in range(L)}}
    m.anchors_pos_ext = {l: d(g[t + 'P_ext_%d' % l]) for l in range(L)}
    m.anchors_structure = {l: (d(g[t + 'S_patches_%d' % l]), [int(i) for i in g[t + 'S_idx_%d' % l]],
                               d(g[t + 'S_int_rw_%d' % l]), d(g[t + 'S_bor_rw_%d' % l])) for l in range(L)}
    m._build_sim_cols()
    m.init_all_embeddings(split='train', trainable=hp['trainable_cc'])
    if hp['trainable_cc']:
        with torch.no_grad():
            for nm in ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B'):
                getattr(m, 'train_%s_cc_embed' % nm).copy_(d(g[t + 'cc_param/' + nm]))


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('variant', G11_VARIANTS)
def test_forward_backward_parity_g11(tiny, tmp_path, variant, fused):
    g = tiny
    t = 'g11_%s/' % variant
    hp = json.loads(str(g[t + 'hparams']))
    hp['fused_forward'] = fused
    m = _model(g, tmp_path, hp)
    hp = m.hparams
    sd = {k[len(t) + 3:]: T(g[k]) for k in g.files if k.startswith(t + 'sd/')}
    own = {k: v for k, v in sd.items() if not k.startswith('train_')}
    missing, unexpected = m.load_state_dict(own, strict=False)
    assert not unexpected and all(k.startswith('train_') for k in missing)
    _inject(m, g, t, hp)
    m.train()
    batch = m.make_batch('train', g[t + 'idx'])
    res = m.training_step(batch, 0)
    logits = m._forward_batch('train', batch)
    m.zero_grad()
    m.backward(None, res['loss'], None, 0)
    assert_close(logits, g[t + 'logits'], 'logits')
    assert_close(res['loss'], g[t + 'loss'], 'loss')
    n = 0
    params = dict(m.named_parameters())
    for k in g.files:
        if not k.startswith(t + 'grad/'):
            continue
        nm = k[len(t) + 5:]
        if nm.startswith('train_'):
            nm2 = nm
        ref = g[k]
        p = params[nm]
        if p.grad is None:
            assert np.abs(ref).max() == 0, nm
        else:
            assert_close(p.grad, ref, 'grad ' + nm)
            n += 1
    assert n > 10


@pytest.mark.parametrize('variant', ['sum', 'max_trainable'])
def test_fp16_stored_table_equals_fp32_on_rounded_values(tiny, tmp_path, variant):
    """hparams['embedding_dtype'] = 'fp16' (BASELINE.json configs[4]): the fused kernels read an IEEE
    half copy of the table and accumulate in fp32.  With a master table whose values are exactly
    representable in half, the fp16-storage path must reproduce the fp32 path -- logits, loss and
    every gradient (which still flows to the fp32 master) -- and the half copy must follow the master
    when the optimizer changes it."""
    g = tiny
    t = 'g11_%s/' % variant
    res = {}
    for mode in ('fp32', 'fp16'):
        hp = json.loads(str(g[t + 'hparams']))
        hp['embedding_dtype'] = mode
        m = _model(g, tmp_path, hp)
        hp = m.hparams
        sd = {k[len(t) + 3:]: T(g[k]) for k in g.files if k.startswith(t + 'sd/')}
        m.load_state_dict({k: v for k, v in sd.items() if not k.startswith('train_')}, strict=False)
        with torch.no_grad():
            m.node_embeddings.weight.copy_(m.node_embeddings.weight.half().float())     # half-representable master
        _inject(m, g, t, hp)
        m.train()
        batch = m.make_batch('train', g[t + 'idx'])
        out = m.training_step(batch, 0)
        m.zero_grad()
        m.backward(None, out['loss'], None, 0)
        res[mode] = (out['loss'].detach().clone(), m._forward_batch('train', batch).detach(),
                     {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        if mode == 'fp16':
            h0 = m._half_table()
            assert h0.dtype == torch.float16 and h0 is m._half_table()                  # one persistent buffer
            with torch.no_grad():
                m.node_embeddings.weight.add_(1.0)
                h1 = m._half_table()                                                    # refreshed in place
                assert h1 is h0 and torch.equal(h1.float(), m.node_embeddings.weight.half().float())
                # a change nothing can see (fused Adam and graph replays are handled by the dirty mark a
                # training read leaves; this is a raw .data edit)
                m.node_embeddings.weight.data.sub_(0.5)
                assert not torch.equal(m._half_table().float(), m.node_embeddings.weight.half().float())
                m.invalidate_half_table()
                assert torch.equal(m._half_table().float(), m.node_embeddings.weight.half().float())
            # fused Adam does not move the parameter's version counter: the copy must follow all the same
            opt = m.configure_optimizers()
            out = m.training_step(batch, 0)
            opt.zero_grad()
            m.backward(None, out['loss'], None, 0)
            before = m.node_embeddings.weight.detach().clone()
            opt.step()
            assert not torch.equal(before, m.node_embeddings.weight.detach())
            with torch.no_grad():
                assert torch.equal(m._half_table().float(), m.node_embeddings.weight.half().float())
    assert_close(res['fp16'][0], res['fp32'][0].cpu().numpy(), 'loss')
    assert_close(res['fp16'][1], res['fp32'][1].cpu().numpy(), 'logits')
    assert res['fp16'][2].keys() == res['fp32'][2].keys()
    for k in res['fp32'][2]:
        assert_close(res['fp16'][2][k], res['fp32'][2][k].cpu().numpy(), 'grad ' + k)
    assert float(res['fp16'][2]['node_embeddings.weight'].abs().max()) > 0


def test_precompute_graph_metrics(golden, tmp_path):
    """GPU graph-metric precompute (the SNAP script's outputs) against the fixture's own files."""
    import json as _json
    import os
    from subgnn_amd import precompute_graph_metrics as pgm
    name = write_dataset_from_golden(golden, tmp_path, with_ego=True)
    d = tmp_path / name
    ref_deg = _json.load(open(d / 'degree_sequence.txt'))
    ref_ego = _json.load(open(d / 'ego_graphs.txt'))
    for f in ('degree_sequence.txt', 'ego_graphs.txt', 'shortest_path_matrix.npy'):
        os.remove(d / f)
    pgm.calculate_stats(d, torch.device(DEV))
    assert np.array_equal(np.load(d / 'shortest_path_matrix.npy'), golden['apsp'])
    assert _json.load(open(d / 'degree_sequence.txt')) == ref_deg
    got = _json.load(open(d / 'ego_graphs.txt'))
    assert {k: sorted(v) for k, v in got.items()} == {k: sorted(v) for k, v in ref_ego.items()}


def _golden_prepared(m, g, sp='train'):
    """The reference's own prepared tensors (its component order is CPython-set order, so batches of
    the product's own prepare_data cannot be compared row by row) put into the model."""
    d = lambda x: T(x).to(DEV)
    hp = m.hparams
    L = hp['n_layers']
    m.train_cc_ids = d(g['g2_cc_ids_train'])
    m.train_N_border = d(g['g3_border_train'])
    m.train_neigh_pos_similarities = d(g['g4_np_sim_train'])
    m.train_int_struc_similarities = d(g['g7_int_struc_sim_train'])
    m.train_bor_struc_similarities = d(g['g7_bor_struc_sim_train'])
    m.anchors_neigh_int = {'train': {l: d(g['g8_N_int_train_%d' % l]) for l in range(L)}}
    m.anchors_neigh_border = {'train': {l: d(g['g8_N_bor_train_%d' % l]) for l in range(L)}}
    m.anchors_pos_int = {'train': {l: d(g['g8_P_int_train_%d' % l]) for l in range(L)}}
    m.anchors_pos_ext = {l: d(g['g8_P_ext_%d' % l]) for l in range(L)}
    m.anchors_structure = {l: (d(g['g8_S_patches_%d' % l]), [int(i) for i in g['g8_S_idx_%d' % l]],
                               d(g['g8_S_int_rw_%d' % l]), d(g['g8_S_bor_rw_%d' % l])) for l in range(L)}
    m.__dict__.pop('_resident', None)


def test_pad_collate_and_make_batch_g12(golden, tmp_path):
    """g12 at the product level: SubgraphDataset items through SubGNN._pad_collate, and the
    device-resident make_batch the loaders use, both give the batch the reference's _pad_collate
    built from the same prepared tensors (S.py:1068-1114)."""
    m = _model(golden, tmp_path)
    _golden_prepared(m, golden)
    idxs = [int(i) for i in golden['g12_idx']]
    ds = m._dataset('train')
    b1 = m._pad_collate([ds[i] for i in idxs])
    b2 = m.make_batch('train', idxs)
    for k in ('cc_ids', 'N_border', 'NP_sim', 'I_S_sim', 'B_S_sim', 'subgraph_idx', 'label'):
        assert np.array_equal(b1[k].cpu().numpy(), golden['g12_' + k]), k
        assert np.array_equal(b2[k].cpu().numpy(), golden['g12_' + k]), k
    assert np.array_equal(b1['subgraph_ids'].cpu().numpy(), golden['g12_subgraph_ids'])
    w = golden['g12_subgraph_ids'].shape[1]                     # make_batch keeps the split's padded width
    s2 = b2['subgraph_ids'].cpu().numpy()
    assert np.array_equal(s2[:, :w], golden['g12_subgraph_ids']) and (s2[:, w:] == 0).all()


@pytest.mark.parametrize('ch,inside', [(c, i) for c in ('N', 'P', 'S') for i in (True, False)])
def test_get_anchor_patches_g9(tiny, tmp_path, ch, inside):
    """g9 at the product level: anchor_patch_samplers.get_anchor_patches (aps:333-399) with the
    reference's parameters and sampled anchors -> the reference's (patches, mask, embeds)."""
    from subgnn_amd import anchor_patch_samplers as aps
    g = tiny
    m = _model(g, tmp_path)
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith('sd/')}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing
    _golden_prepared(m, g)
    cc_ids = T(g['g12_cc_ids']).to(DEV)
    mask = (cc_ids != 0)[:, :, 0]
    channel = {'N': 'neighborhood', 'P': 'position', 'S': 'structure'}[ch]
    m.eval()
    with torch.no_grad():
        ap, am, ae = aps.get_anchor_patches('train', m.hparams, m.networkx_graph, m.node_embeddings,
                                            T(g['g12_subgraph_idx']).to(DEV), cc_ids, mask, m.lstm, m.anchors_neigh_int,
                                            m.anchors_neigh_border, m.anchors_pos_int, m.anchors_pos_ext,
                                            m.anchors_structure, 1, channel, inside, m.device)
    tag = 'g9_%s_%s_' % (ch, 'in' if inside else 'out')
    assert np.array_equal(ap.cpu().numpy(), g[tag + 'patches'])
    assert np.array_equal(am.cpu().numpy(), g[tag + 'mask'])
    assert_close(ae, g[tag + 'embeds'], 'anchor embeds')


def test_ff_attn_with_fp16_takes_the_half_mfma_scores(tiny, tmp_path):
    """BASELINE configs[4] wiring: embedding_dtype = 'fp16' + ff_attn -> the attention scores of the read-out run
    on v_mfma_f32_32x32x16_f16 (AdditiveAttention.half_operands); logits stay within half precision of the
    exact model and every parameter still receives a gradient."""
    g = tiny
    t = 'g11_ff_attn/'
    outs = {}
    for mode in ('fp32', 'fp16'):
        hp = json.loads(str(g[t + 'hparams']))
        hp['embedding_dtype'] = mode
        m = _model(g, tmp_path, hp)
        sd = {k[len(t) + 3:]: T(g[k]) for k in g.files if k.startswith(t + 'sd/')}
        m.load_state_dict({k: v for k, v in sd.items() if not k.startswith('train_')}, strict=False)
        _inject(m, g, t, m.hparams)
        assert m.attention.half_operands == (mode == 'fp16')
        m.train()
        batch = m.make_batch('train', g[t + 'idx'])
        out = m.training_step(batch, 0)
        m.zero_grad()
        m.backward(None, out['loss'], None, 0)
        outs[mode] = (m._forward_batch('train', batch).detach(), m.attention._u_matrix.grad.clone())
    assert_close(outs['fp32'][0], g[t + 'logits'], 'logits fp32')
    assert_close(outs['fp16'][0], outs['fp32'][0].cpu().numpy(), 'logits fp16 vs fp32', 2e-2)
    assert float(outs['fp16'][1].abs().max()) > 0
