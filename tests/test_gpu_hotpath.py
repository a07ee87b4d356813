"""-m gpu: the sparse precompute (hotpath.prepare_sparse) gives the model the same numbers the
dense reference-shaped prepare_data gathers from its slabs, and a DP-sharded pass reproduces
the single-rank pass."""
import numpy as np
import pytest
import torch

from helpers import assert_close, write_dataset_from_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _models(golden, tmp_path, over=None):
    from subgnn_amd import config
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    name = write_dataset_from_golden(golden, tmp_path, with_ego=False)   # true k-hop border in both paths
    config.PROJECT_ROOT = tmp_path
    out = []
    for _ in range(2):
        hp = dict(golden.hp)
        hp.update({'seed': golden.seed, 'neigh_sample_border_size': 2, 'lin_dropout': 0.0})
        if over:
            hp.update(over)
        torch.manual_seed(0)
        out.append(SubGNN(hp, **dataset_paths(name)))
    return out


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_sparse_prepare_equals_dense_prepare(name, tmp_path):
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden(name)                 # no ego dict: true k-hop border in both paths
    dense, sparse = _models(golden, tmp_path)
    dense.prepare_data()
    hotpath.prepare_sparse(sparse, 'train')
    _compare_dense_and_sparse(dense, sparse)
    # resample_anchor_patches (SubGNN.py:453-460) on both: the dense model re-draws from its kept border sets and slabs, the
    # sparse one runs a new sparse pass keyed by the resample epoch (round 6: it used to fail on the border sets it never had) --
    # fresh draws, the same on both paths, with their similarities
    before = {l: sparse.anchors_neigh_border['train'][l].clone() for l in range(dense.hparams['n_layers'])}
    p_before = {l: sparse.anchors_pos_ext[l].clone() for l in range(dense.hparams['n_layers'])}
    for m in (dense, sparse):
        m.__dict__['_resample_epoch'] = 1
        m._prepare_anchors_only()
    assert any(not torch.equal(before[l], sparse.anchors_neigh_border['train'][l]) for l in before)
    assert any(not torch.equal(p_before[l], sparse.anchors_pos_ext[l]) for l in p_before)
    _compare_dense_and_sparse(dense, sparse, structure_picks=True)


def _compare_dense_and_sparse(dense, sparse, structure_picks=False):
    hp = dense.hparams
    assert torch.equal(dense.train_cc_ids, sparse.train_cc_ids)
    slab = dense.train_neigh_pos_similarities
    S, C, _ = slab.shape
    if structure_picks and hp['use_structure']:
        for l in range(hp['n_layers']):
            a, b = dense.anchors_structure[l], sparse.anchors_structure[l]
            assert [int(v) for v in a[1]] == [int(v) for v in b[1]]              # the re-picked patch numbers
    for l in range(hp['n_layers']):
        # anchors: identical draws (same tape; ragged sampler == padded sampler)
        assert torch.equal(dense.anchors_neigh_int['train'][l], sparse.anchors_neigh_int['train'][l])
        assert torch.equal(dense.anchors_neigh_border['train'][l], sparse.anchors_neigh_border['train'][l])
        assert torch.equal(dense.anchors_pos_int['train'][l], sparse.anchors_pos_int['train'][l])
        assert torch.equal(dense.anchors_pos_ext[l], sparse.anchors_pos_ext[l])
        real = (dense.train_cc_ids[:, :, 0] != 0)
        for key, ids in ((('N', 'in', l), dense.anchors_neigh_int['train'][l]),
                         (('N', 'out', l), dense.anchors_neigh_border['train'][l]),
                         (('P', 'in', l), dense.anchors_pos_int['train'][l].unsqueeze(1).expand(S, C, -1)),
                         (('P', 'out', l), dense.anchors_pos_ext[l].view(1, 1, -1).expand(S, C, -1))):
            edge = (ids != 0) & real.unsqueeze(-1)
            want = torch.gather(slab, 2, (ids - 1).clamp(min=0)) * edge
            got = sparse.train_neigh_pos_similarities[key]
            if key[:2] == ('P', 'out'):
                assert float((got * (~real).unsqueeze(-1)).abs().max()) == 0.0    # padded components: rows stay 0 unmasked
            got = (got.dense() if hasattr(got, 'dense') else got) * edge          # ops.ZeroSims: known-zero weights
            assert torch.equal(got, want), key
    # (the per-pass path keeps the walks' full width; the dense prepare trims trailing all-PAD columns like the reference)
    w = dense.structure_anchors.shape[1]
    assert torch.equal(dense.structure_anchors, sparse.structure_anchors[:, :w]) and not bool(sparse.structure_anchors[:, w:].any())
    assert torch.equal(dense.train_int_struc_similarities, sparse.train_int_struc_similarities)
    assert torch.equal(dense.train_bor_struc_similarities, sparse.train_bor_struc_similarities)
    # same logits through the per-edge similarity dict as through the dense slab
    idx = torch.arange(6)
    dense.eval(); sparse.eval()
    sparse.load_state_dict(dense.state_dict())
    with torch.no_grad():
        a = dense._forward_batch('train', dense.make_batch('train', idx))
        b = sparse._forward_batch('train', sparse.make_batch('train', idx))
    assert_close(b, a, 'logits sparse vs dense', norm_tol=1e-5)


@pytest.mark.parametrize('graph_step', [False, True])
def test_trainer_resamples_a_sparse_prepared_model_like_a_dense_one(graph_step, tmp_path):
    """train_config.Trainer over two epochs with resample_anchor_patches on a model prepared by hotpath.prepare_sparse (graphs
    whose N x N structures cannot exist): every epoch end draws epoch-keyed anchors with a new sparse pass -- and trains to the
    losses of the dense-prepared twin, whose resample re-draws from its kept border sets (same tape, same similarities)."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    from subgnn_amd.train_config import Trainer
    golden = load_golden('density')
    dense, sparse = _models(golden, tmp_path, {'resample_anchor_patches': True, 'lstm_dropout': 0.0})
    sparse.load_state_dict(dense.state_dict())
    dense.prepare_data()
    for sp in ('train', 'val'):
        hotpath.prepare_sparse(sparse, sp)
    hist = []
    for m, prepared in ((dense, True), (sparse, True)):
        # (graph_step: the recorded training step; the sparse model's resample installs new tensors, so it records again per epoch)
        tr = Trainer(2, m.hparams.get('grad_clip', 0.0), log=lambda *a, **k: None, hip_graph_step=graph_step)
        torch.manual_seed(5)                               # (the loaders' shuffles)
        tr.fit(m, prepared=prepared)
        assert m.__dict__['_resample_epoch'] == 2
        hist.append([(h['train_loss'], h['val_loss']) for h in tr.history])
    for (a, b), (c, d) in zip(*hist):
        assert abs(a - c) <= 1e-5 * max(1.0, abs(a)) and abs(b - d) <= 1e-5 * max(1.0, abs(b)), hist


def test_sparse_prepared_model_fits_and_tests_like_the_dense_one(tmp_path):
    """The whole driver loop on a sparse-prepared model -- one epoch of train_config.Trainer, then the TEST split prepared by
    hotpath.prepare_sparse(model, 'test') and Trainer.test -- gives the dense-prepared twin's test loss and metrics
    (prepare_test_data, SubGNN.py:994-1022, and test_epoch_end, :466-520)."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    from subgnn_amd.train_config import Trainer
    golden = load_golden('density')
    dense, sparse = _models(golden, tmp_path, {'lstm_dropout': 0.0})
    sparse.load_state_dict(dense.state_dict())
    logs = []
    for m, is_sparse in ((dense, False), (sparse, True)):
        if is_sparse:
            for sp in ('train', 'val'):
                hotpath.prepare_sparse(m, sp)
        else:
            m.prepare_data()
        tr = Trainer(1, m.hparams.get('grad_clip', 0.0), log=lambda *a, **k: None, hip_graph_step=False)
        torch.manual_seed(5)
        tr.fit(m, prepared=True)
        if is_sparse:
            hotpath.prepare_sparse(m, 'test')
        else:
            m.prepare_test_data()
        out = tr.test(m)
        logs.append({k: float(v) for k, v in out['log'].items() if any(t in k for t in ('loss', 'f1', 'acc'))})
    assert logs[0].keys() == logs[1].keys() and 'test_loss' in logs[0]
    for k in logs[0]:
        assert abs(logs[0][k] - logs[1][k]) <= 1e-5 * max(1.0, abs(logs[0][k])), (k, logs)


def test_full_split_step_runs_and_is_deterministic(tmp_path):
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden('density')
    (m, _) = _models(golden, tmp_path)
    hotpath.prepare_sparse(m, 'train')
    batch = hotpath.full_split_batch(m, 'train')
    m.eval()
    with torch.no_grad():
        a = m._forward_batch('train', batch)
        hotpath.prepare_sparse(m, 'train')          # same seed -> same tape -> same draws
        b = m._forward_batch('train', hotpath.full_split_batch(m, 'train'))
    assert torch.equal(a, b)
    assert a.shape[0] == len(m.train_sub_G)


@pytest.mark.parametrize('name,world', [('tiny', 3), ('density', 4)])
def test_sharded_passes_reproduce_the_single_rank_pass(name, world, tmp_path):
    """Data parallelism over subgraph shards (dist.Shard): each rank's pass over ITS block of the split --
    draws keyed by the global subgraph number, padded widths reduced over ranks -- yields exactly the rows
    the single-rank pass yields for those subgraphs: component tensors, every anchor tensor, every
    similarity.  One process plays the ranks in turn (uneven blocks included); the reductions over ranks
    are replaced by the known global values."""
    from conftest import load_golden
    from subgnn_amd import hotpath, ops
    from subgnn_amd import dist as sdist
    golden = load_golden(name)
    full, part = _models(golden, tmp_path)
    hotpath.prepare_sparse(full, 'train')
    hp = full.hparams
    S, C, L = full.train_cc_ids.shape
    sets = ops.Ragged.from_padded(full.train_cc_ids.reshape(S * C, L))
    width = ops.khop_border(full.networkx_graph, sets, hp['neigh_sample_border_size']).lengths.max().view(1)
    dims = torch.tensor([C, L], device=width.device)

    class PlayedShard(sdist.Shard):
        def reduce_max(self, t):
            return torch.maximum(t, (dims if t.numel() == 2 else width).to(t.dtype))
    all_subs, all_labels = list(full.train_sub_G), full.train_sub_G_label
    for r in range(world):
        sh = PlayedShard(S, r, world, collectives=False)
        a, b = sh.start, sh.stop
        part.train_sub_G, part.train_sub_G_label = all_subs[a:b], all_labels[a:b]
        part.__dict__.pop('_subs_train', None)
        for k in ('_degseq_order', '_dtw_group_rows'):
            part.__dict__.pop(k, None)
        hotpath.prepare_sparse(part, 'train', shard=sh)
        assert torch.equal(part.train_cc_ids, full.train_cc_ids[a:b])
        for l in range(hp['n_layers']):
            assert torch.equal(part.anchors_neigh_int['train'][l], full.anchors_neigh_int['train'][l][a:b])
            assert torch.equal(part.anchors_neigh_border['train'][l], full.anchors_neigh_border['train'][l][a:b])
            assert torch.equal(part.anchors_pos_int['train'][l], full.anchors_pos_int['train'][l][a:b])
            assert torch.equal(part.anchors_pos_ext[l], full.anchors_pos_ext[l])
            for key in (('N', 'out', l), ('P', 'out', l), ('P', 'in', l)):
                x, y = part.train_neigh_pos_similarities[key], full.train_neigh_pos_similarities[key]
                if isinstance(y, ops.ZeroSims):
                    assert isinstance(x, ops.ZeroSims) and x.shape[1:] == y.shape[1:]
                else:
                    assert torch.equal(x, y[a:b]), key
        assert torch.equal(part.structure_anchors, full.structure_anchors)
        assert torch.equal(part.train_int_struc_similarities, full.train_int_struc_similarities[a:b])
        assert torch.equal(part.train_bor_struc_similarities, full.train_bor_struc_similarities[a:b])


def test_bfs_level_hint_gives_the_same_similarities_and_a_short_hint_is_repaired(tmp_path):
    """The second pass enqueues (levels the first pass needed + margin + the empty level that proves the end) BFS levels
    instead of max_bfs_hops: same similarities.  A hint that is too small is noticed BEFORE the pass is consumed
    (install_pass): the search is repeated with the full cap and the similarities replaced -- training never sees the
    truncated ones; only a graph deeper than max_bfs_hops itself is an error."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden('density')
    (m, _) = _models(golden, tmp_path)
    hotpath.prepare_sparse(m, 'train')
    L = m.hparams['n_layers']
    first = [m.train_neigh_pos_similarities[('P', 'out', l)].clone() for l in range(L)]
    hint = dict(m._bfs_level_hint)
    assert set(hint) == {('P_out', 'train', l) for l in range(L)} and all(1 <= v < 32 for v in hint.values())
    st = hotpath.prepare_pass(m, 'train')                             # hinted
    assert len(st.bfs_checks) == L
    assert all(enq == hint[key] + hotpath.BFS_LEVEL_MARGIN + 1 for key, _, _, _, enq, _ in st.bfs_checks)
    hotpath.install_pass(m, st)
    assert not st.bfs_checks and not m.__dict__.get('_bfs_redone')
    for l in range(L):
        assert torch.equal(m.train_neigh_pos_similarities[('P', 'out', l)], first[l])
    # a search exactly (margin) levels deeper than the hint still passes without a repeat
    for key in hint:
        m._bfs_level_hint[key] = hint[key] - hotpath.BFS_LEVEL_MARGIN
    hotpath.prepare_sparse(m, 'train')
    assert not m.__dict__.get('_bfs_redone') and m._bfs_level_hint == hint
    # a hint that is too small: the truncated similarities never reach the model
    for key in hint:
        m._bfs_level_hint[key] = -hotpath.BFS_LEVEL_MARGIN            # enqueue exactly one level
    st = hotpath.prepare_pass(m, 'train')
    short = [st.attrs['train_neigh_pos_similarities'][('P', 'out', l)].clone() for l in range(L)]
    assert any(not torch.equal(short[l], first[l]) for l in range(L))  # (what the search produced before the check)
    hotpath.install_pass(m, st)
    assert m._bfs_redone == L and m._bfs_level_hint == hint            # repeated with the full cap, hint repaired
    for l in range(L):
        assert torch.equal(m.train_neigh_pos_similarities[('P', 'out', l)], first[l])
    hotpath.prepare_sparse(m, 'train')
    assert m._bfs_redone == L
    for l in range(L):
        assert torch.equal(m.train_neigh_pos_similarities[('P', 'out', l)], first[l])
    # the cap itself is checked too: on the first search, and when a repeated search still runs out
    m.hparams['max_bfs_hops'] = 1
    for key in hint:
        m._bfs_level_hint[key] = 0                                    # hint + margin + 1 > cap: capped at 1 level
    with pytest.raises(RuntimeError, match='max_bfs_hops'):
        hotpath.prepare_sparse(m, 'train')
    m._bfs_level_hint.clear()
    with pytest.raises(RuntimeError, match='max_bfs_hops'):
        hotpath.prepare_sparse(m, 'train')


def _pass_outputs(m, split='train'):
    L = m.hparams['n_layers']
    out = {'cc_ids': getattr(m, split + '_cc_ids'), 'structure_anchors': m.structure_anchors,
           'int_walks': m.int_structure_anchor_random_walks, 'bor_walks': m.bor_structure_anchor_random_walks,
           'int_struc': getattr(m, split + '_int_struc_similarities'), 'bor_struc': getattr(m, split + '_bor_struc_similarities')}
    sims = getattr(m, split + '_neigh_pos_similarities')
    for l in range(L):
        out['N_in%d' % l] = m.anchors_neigh_int[split][l]
        out['N_out%d' % l] = m.anchors_neigh_border[split][l]
        out['P_in%d' % l] = m.anchors_pos_int[split][l]
        out['P_out%d' % l] = m.anchors_pos_ext[l]
        for key in (('N', 'out', l), ('P', 'out', l), ('P', 'in', l)):
            v = sims[key]
            out['sim_%s_%s%d' % key] = v.dense() if hasattr(v, 'dense') else v
        patches, _, iw, bw = m.anchors_structure[l]
        out['S_patches%d' % l], out['S_iw%d' % l], out['S_bw%d' % l] = patches, iw, bw
    return {k: v.clone() for k, v in out.items()}


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_two_stream_pass_equals_the_one_stream_pass(name, tmp_path):
    """hparams['overlap_streams']: the side stream (structure patches, position BFS, walks) and the main stream
    (border BFS, draws, degree sequences) produce what the single-stream pass produces -- also on repeated passes,
    when the caching allocator hands blocks freed on one stream to the other -- and a training step runs on it."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden(name)
    (one, two) = _models(golden, tmp_path)
    one.hparams['overlap_streams'] = False
    two.hparams['overlap_streams'] = True
    hotpath.prepare_sparse(one, 'train')
    want = _pass_outputs(one)
    for rep in range(4):
        hotpath.prepare_sparse(two, 'train')
        assert two._side_stream is not None
        got = _pass_outputs(two)
        for k in want:
            assert torch.equal(got[k], want[k]), (rep, k)
        junk = [torch.randn(1 << 18, device=DEV) for _ in range(8)]      # churn the allocator between passes
        del junk
    two.load_state_dict(one.state_dict())
    one.train(); two.train()
    la = one.training_step(hotpath.full_split_batch(one, 'train'), 0)['loss']
    lb = two.training_step(hotpath.full_split_batch(two, 'train'), 0)['loss']
    la.backward(); lb.backward()
    assert float(la) == float(lb)
    assert torch.equal(one.node_embeddings.weight.grad, two.node_embeddings.weight.grad)


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_pipelined_passes_train_like_sequential_passes(name, tmp_path):
    """hotpath.PassPipeline: preparing pass k + 1 on a side stream while pass k trains gives the losses and the
    parameters of prepare-then-train, step for step (a pass's draws depend on the seed and the resample counter
    only; nothing the preparation reads is written by the training half)."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden(name)
    (seq, pip) = _models(golden, tmp_path)
    pip.load_state_dict(seq.state_dict())
    seq.train(); pip.train()

    def train(m, opt):
        out = m.training_step(hotpath.full_split_batch(m, 'train'), 0)
        out['loss'].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return float(out['loss'])
    o_seq, o_pip = seq.configure_optimizers(), pip.configure_optimizers()
    want = []
    for k in range(4):
        hotpath.prepare_sparse(seq, 'train')
        want.append(train(seq, o_seq))
    pipe = hotpath.PassPipeline(pip, 'train')
    pipe.start()
    got = []
    for k in range(4):
        pipe.install()
        pipe.start()
        got.append(train(pip, o_pip))
    torch.cuda.synchronize()
    assert got == want
    for (n1, a), (_, b) in zip(seq.named_parameters(), pip.named_parameters()):
        assert torch.equal(a, b), n1
    with pytest.raises(RuntimeError):
        hotpath.PassPipeline(pip, 'train').install()


@pytest.mark.parametrize('name,big_bytes,pipelined', [('tiny', 1 << 30, True), ('tiny', 1024, False), ('density', 1024, True)])
def test_training_half_replayed_from_a_hipgraph_trains_like_the_eager_step(name, big_bytes, pipelined, tmp_path):
    """hotpath.CapturedTraining: the training half (training_step -> backward -> ClipAdam) recorded into a hipGraph on
    the second pass and replayed, every later pass copied into the recording's tensors (install_pass_static) -- losses
    and parameters equal the eager prepare-then-train schedule bit for bit over 5 passes (clip + Adam with device-side
    step counts; without dropout: a replayed graph draws its masks from its own Philox offsets)."""
    from conftest import load_golden
    from subgnn_amd import hotpath, optim
    golden = load_golden(name)
    (seq, cap) = _models(golden, tmp_path, {'lin_dropout': 0.0, 'lstm_dropout': 0.0})
    cap.load_state_dict(seq.state_dict())
    seq.train(); cap.train()
    lr, clip = 0.01, 0.5
    o_seq = optim.ClipAdam(seq.parameters(), lr, max_norm=clip, big_bytes=big_bytes)
    o_cap = optim.ClipAdam(cap.parameters(), lr, max_norm=clip, big_bytes=big_bytes, capturable=True)
    assert (len(o_cap.big) > 0) == (big_bytes < (1 << 30))
    want = []
    for k in range(5):
        hotpath.prepare_sparse(seq, 'train')
        out = seq.training_step(hotpath.full_split_batch(seq, 'train'), 0)
        out['loss'].backward()
        o_seq.step()
        o_seq.zero_grad(set_to_none=True)
        want.append(float(out['loss']))
    trainer = hotpath.CapturedTraining(cap, o_cap, 'train', warmup=1)
    got = []
    if pipelined:
        pipe = hotpath.PassPipeline(cap, 'train')
        pipe.start()
    for k in range(5):
        if pipelined:
            pipe.install(installer=trainer.install)
            ev = torch.cuda.Event()
            ev.record()
            loss, acc = trainer.step()
            pipe.start(after=ev)
        else:
            trainer.install(hotpath.prepare_pass(cap, 'train'))
            loss, acc = trainer.step()
        got.append(float(loss))
    torch.cuda.synchronize()
    assert trainer.recordings == 1 and trainer.graph is not None, trainer.last_changed
    assert got == want
    for (n1, a), (_, b) in zip(seq.named_parameters(), cap.named_parameters()):
        assert torch.equal(a, b), n1
    with pytest.raises(ValueError):
        hotpath.CapturedTraining(cap, optim.ClipAdam(cap.parameters(), lr), 'train')      # a host step count cannot be replayed


@pytest.mark.parametrize('name,big_bytes', [('density', 1024), ('density', 1 << 30)])
def test_both_halves_replayed_from_hipgraphs_train_like_the_eager_passes(name, big_bytes, tmp_path):
    """hotpath.GraphedPasses: the sampling + similarity half AND the training half of a pass recorded into hipGraphs, two
    alternating slots (slot B's preparation replays on a second stream while slot A trains), nothing installed or copied
    between the halves -- losses and parameters equal the eager prepare-then-train schedule bit for bit over 9 passes
    (two eager warm-up passes, one recording per slot, five replayed passes)."""
    from conftest import load_golden
    from subgnn_amd import hotpath, optim
    golden = load_golden(name)
    (seq, cap) = _models(golden, tmp_path, {'lin_dropout': 0.0, 'lstm_dropout': 0.0})
    cap.load_state_dict(seq.state_dict())
    seq.train(); cap.train()
    lr, clip = 0.01, 0.5
    o_seq = optim.ClipAdam(seq.parameters(), lr, max_norm=clip, big_bytes=big_bytes)
    o_cap = optim.ClipAdam(cap.parameters(), lr, max_norm=clip, big_bytes=big_bytes, capturable=True)
    want = []
    for k in range(9):
        hotpath.prepare_sparse(seq, 'train')
        out = seq.training_step(hotpath.full_split_batch(seq, 'train'), 0)
        out['loss'].backward()
        o_seq.step()
        o_seq.zero_grad(set_to_none=True)
        want.append(float(out['loss']))
    passes = hotpath.GraphedPasses(cap, o_cap, 'train', warmup=2)
    got = []
    for k in range(9):
        loss, acc = passes.step()
        got.append(float(loss))
    torch.cuda.synchronize()
    assert passes.recordings == 2 and all(s is not None for s in passes.slots)
    assert got == want
    for (n1, a), (_, b) in zip(seq.named_parameters(), cap.named_parameters()):
        assert torch.equal(a, b), n1
    with pytest.raises(ValueError):
        hotpath.GraphedPasses(cap, optim.ClipAdam(cap.parameters(), lr), 'train')


@pytest.mark.parametrize('tie', [1, 2])
def test_dtw_tie_order_hparam_reaches_both_paths(tie, tmp_path):
    """hparams['dtw_tie_order'] selects fastdtw's predecessor rule end to end: the dense prepare_data and the sparse
    hot path both hand it to the DTW launch, and the structure similarities equal the oracle's under that rule."""
    from conftest import load_golden
    from oracle import cbind
    from subgnn_amd import hotpath, gamma, ops
    golden = load_golden('tiny')
    dense, sparse = _models(golden, tmp_path, {'dtw_tie_order': tie})
    dense.prepare_data()
    hotpath.prepare_sparse(sparse, 'train')
    assert torch.equal(dense.train_int_struc_similarities, sparse.train_int_struc_similarities)
    assert torch.equal(dense.train_bor_struc_similarities, sparse.train_bor_struc_similarities)
    S, C, L = dense.train_cc_ids.shape
    g = dense.networkx_graph
    use_dict = g.full_degree is not None
    for internal, got in ((True, dense.train_int_struc_similarities), (False, dense.train_bor_struc_similarities)):
        c_sets, c_seq = gamma.degree_sequences(g, dense.train_cc_ids.view(S * C, L), internal, use_dict)
        a_sets, a_seq = gamma.degree_sequences(g, dense.structure_anchors, internal, use_dict)
        ref = cbind.fastdtw_sim(c_sets.ptr.cpu().numpy(), c_seq.cpu().numpy()[:int(c_sets.ptr[-1])],
                                a_sets.ptr.cpu().numpy(), a_seq.cpu().numpy()[:int(a_sets.ptr[-1])], tie)
        assert np.array_equal(got.view(S * C, -1).cpu().numpy(), ref)
    with pytest.raises(ValueError):
        _models(golden, tmp_path, {'dtw_tie_order': 3})


def test_streamed_p_internal_similarities_equal_the_hop_table_form(tmp_path, monkeypatch):
    """Multi-component subgraphs: above the hop-table budget the distinct P-internal anchors are streamed through the
    fused BFS + set-min in blocks of sources (round 2 raised NotImplementedError there).  Forced here with a zero
    budget and a block of 64 sources: same similarities as the hop-table form, same logits."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden('tiny')
    table, streamed = _models(golden, tmp_path)
    assert hotpath.prepare_pass(table, 'train').attrs['train_cc_ids'].shape[1] > 1          # several components per subgraph
    hotpath.prepare_sparse(table, 'train')
    monkeypatch.setattr(hotpath, 'MAX_PINT_BYTES', 0)
    real = hotpath._pint_sims_streamed
    calls = []

    def small_blocks(*a, **k):
        calls.append(1)
        return real(*a, chunk_bytes=4 * 64 * a[4] * a[5], **k)            # -> blocks of 64 sources
    monkeypatch.setattr(hotpath, '_pint_sims_streamed', small_blocks)
    hotpath.prepare_sparse(streamed, 'train')
    L = table.hparams['n_layers']
    assert len(calls) == L
    for l in range(L):
        a, b = table.train_neigh_pos_similarities[('P', 'in', l)], streamed.train_neigh_pos_similarities[('P', 'in', l)]
        assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_a_subgraph_of_thousands_of_nodes_goes_through_the_whole_pass():
    """No 2048-node limit (the reference pads to any size): a split with one 3000-node subgraph in several components
    next to ordinary ones is prepared by the sparse path (components, border draws, position similarities, degree
    sequences, DTW against rows of thousands of entries) and trained on for a step; the huge components and their
    structure similarities equal the oracle's."""
    import networkx as nx
    from oracle import cbind, graph as OG, integer_half as IH
    from subgnn_amd import hotpath, ops
    from subgnn_amd.SubGNN import SubGNN
    n = 7000
    Gx = nx.barabasi_albert_graph(n, 3, seed=4)
    G = OG.from_edge_pairs(list(Gx.edges()))
    rowptr, col = G.csr()
    rng = np.random.default_rng(1)
    big = (rng.choice(np.arange(1, n + 1), 3000, replace=False)).tolist()
    subs = [big] + [[int(v) for v in rng.choice(np.arange(1, n + 1), 15, replace=False)] for _ in range(11)]
    labels = torch.tensor([i % 3 for i in range(len(subs))])
    g = ops.DeviceGraph(rowptr, col, np.asarray(G.node_order, dtype=np.int32), DEV)
    hp = {"use_neighborhood": True, "use_structure": True, "use_position": True, "seed": 3, "node_embed_size": 16,
          "structure_patch_type": "triangular_random_walk", "lstm_aggregator": "last", "n_processes": 1,
          "resample_anchor_patches": False, "freeze_node_embeds": False, "use_mpn_projection": True,
          "compute_similarities": True, "sample_walk_len": 12, "n_triangular_walks": 3, "random_walk_len": 5, "rw_beta": 0.65,
          "batch_size": 4, "learning_rate": 1e-3, "grad_clip": 1.0, "n_layers": 1, "neigh_sample_border_size": 1,
          "n_anchor_patches_pos_out": 9, "n_anchor_patches_pos_in": 5, "n_anchor_patches_N_in": 4, "n_anchor_patches_N_out": 6,
          "n_anchor_patches_structure": 5, "linear_hidden_dim_1": 16, "linear_hidden_dim_2": 8, "lin_dropout": 0.0,
          "lstm_dropout": 0.0, "lstm_n_layers": 1, "cc_aggregator": "sum", "trainable_cc": False, "max_sim_epochs": 2,
          "embedding_type": "gin"}
    torch.manual_seed(0)
    emb = torch.randn(n, 16, device=DEV)
    m = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []},
                           {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3)
    hotpath.prepare_sparse(m, 'train')
    cc = m.train_cc_ids.cpu().numpy()
    ref = IH.connected_components(G, big)
    assert len(ref) > 5 and max(len(c) for c in ref) > 2048                       # a giant component and many small ones
    assert [[int(v) for v in row if v != 0] for row in cc[0] if row[0] != 0] == ref
    # structure similarities of the giant component's row against the C oracle
    giant = int(np.argmax([len(c) for c in ref]))
    a_sets = ops.Ragged.from_padded(m.structure_anchors)
    for internal, got in ((True, m.train_int_struc_similarities), (False, m.train_bor_struc_similarities)):
        from subgnn_amd import gamma
        _, a_seq = gamma.degree_sequences(g, m.structure_anchors, internal, g.full_degree is not None)
        cp, cf = cbind.ragged([ref[giant]])
        ci, ce = cbind.degree_sequence(rowptr, col, None, cp, cf, True)
        want = cbind.fastdtw_sim(cp, ci if internal else ce, a_sets.ptr.cpu().numpy(),
                                 a_seq.cpu().numpy()[:int(a_sets.ptr[-1])], m.hparams['dtw_tie_order'])
        assert np.array_equal(got[0, giant].cpu().numpy(), want[0])
    m.train()
    batch = hotpath.full_split_batch(m, 'train')
    out = m.training_step(batch, 0)
    m.backward(None, out['loss'], None, 0)
    assert torch.isfinite(out['loss']) and float(m.node_embeddings.weight.grad.abs().max()) > 0


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_position_search_queued_beside_the_dtw_prepares_the_same_pass(name, tmp_path):
    """hparams['bfs_beside_dtw'] only moves the position channel's block (shared anchors, P-internal draws, multi-source BFS)
    behind an event recorded right before the DTW launches: every tensor of the prepared pass is the same, over two passes
    (the second one uses the kept per-split state and the hinted search depth)."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden(name)
    a, b = _models(golden, tmp_path)
    b.hparams['bfs_beside_dtw'] = True

    def snapshot(m):
        out = {}

        def walk(prefix, o):
            if isinstance(o, torch.Tensor):
                out[prefix] = o.detach().clone()
            elif isinstance(o, dict):
                for k, v in o.items():
                    walk('%s[%r]' % (prefix, k), v)
            elif isinstance(o, (list, tuple)):
                for i, v in enumerate(o):
                    walk('%s[%d]' % (prefix, i), v)
        for nm in ('train_cc_ids', 'train_neigh_pos_similarities', 'train_int_struc_similarities', 'train_bor_struc_similarities',
                   'anchors_neigh_int', 'anchors_neigh_border', 'anchors_pos_int', 'anchors_pos_ext', 'anchors_structure',
                   'structure_anchors'):
            walk(nm, getattr(m, nm, None))
        return out
    for _ in range(2):
        hotpath.prepare_sparse(a, 'train')
        hotpath.prepare_sparse(b, 'train')
        torch.cuda.synchronize()
        sa, sb = snapshot(a), snapshot(b)
        assert sa.keys() == sb.keys() and len(sa) > 10
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_pool_reuse_pass_equals_a_full_pass(name, tmp_path):
    """a pass that REUSES the structure-patch pool of an earlier pass (no patches, walks, degree sequences, DTW:
    hotpath.prepare_pass(pool=...), PassPipeline(pool_epochs=max_sim_epochs)) hands the model bit for bit what a full pass does --
    every similarity column the forward consumes, the re-picked patches and walks, and hence the logits."""
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden(name)
    full, reuse = _models(golden, tmp_path)
    first = hotpath.prepare_pass(reuse, 'train')
    hotpath.install_pass(reuse, first)
    st = hotpath.prepare_pass(reuse, 'train', pool=hotpath.pool_of(first))
    assert st.pool_reused and not first.pool_reused                   # nothing of the structure pool was rebuilt
    hotpath.install_pass(reuse, st)
    hotpath.prepare_sparse(full, 'train')
    assert torch.equal(full.structure_anchors, reuse.structure_anchors)
    for sim in ('train_int_struc_similarities', 'train_bor_struc_similarities'):
        assert torch.equal(getattr(full, sim), getattr(reuse, sim))
    for l in range(full.hparams['n_layers']):
        for a, b in zip(full.anchors_structure[l], reuse.anchors_structure[l]):
            assert torch.equal(torch.as_tensor(a), torch.as_tensor(b).to(torch.as_tensor(a).device))
        assert torch.equal(full._sim_col_cache[l], reuse._sim_col_cache[l])
    reuse.load_state_dict(full.state_dict())
    full.eval(); reuse.eval()
    with torch.no_grad():
        a = full._forward_batch('train', hotpath.full_split_batch(full, 'train'))
        b = reuse._forward_batch('train', hotpath.full_split_batch(reuse, 'train'))
    assert torch.equal(a, b)
    # the pipelined form: passes 0 and 2 rebuild the pool, pass 1 re-picks from it
    pipe = hotpath.PassPipeline(reuse, 'train', pool_epochs=2)
    kinds = []
    for _ in range(3):
        pipe.start()
        kinds.append(pipe.pending[-1][0].pool_reused)
        pipe.install()
        with torch.no_grad():
            assert torch.equal(reuse._forward_batch('train', hotpath.full_split_batch(reuse, 'train')), a)
    torch.cuda.synchronize()
    assert kinds == [False, True, False]


@pytest.mark.parametrize('channels', ['N', 'P', 'S', 'NP', 'NS', 'PS', 'NPS'])
def test_every_channel_combination_trains_on_both_prepare_paths(channels, tmp_path):
    """use_neighborhood / use_position / use_structure in every combination (the reference's datasets switch them freely:
    config_files/*/): the dense prepare_data and the sparse prepare give the same logits, the training step matches the oracle
    (logits, loss, every gradient: 1e-4) and a recorded step replays the eager one bit for bit."""
    from conftest import load_golden
    from helpers import oracle_inputs
    from oracle import float_half as FH
    from subgnn_amd import hotpath, optim
    from subgnn_amd.graph_step import CapturedTrainStep
    golden = load_golden('density')
    over = {'use_neighborhood': 'N' in channels, 'use_position': 'P' in channels, 'use_structure': 'S' in channels,
            'lstm_dropout': 0.0}
    dense, sparse = _models(golden, tmp_path, over)
    sparse.load_state_dict(dense.state_dict())
    dense.prepare_data()
    hotpath.prepare_sparse(sparse, 'train')
    hp = dense.hparams
    idx = torch.arange(min(hp['batch_size'], len(dense.train_sub_G)))
    dense.eval(); sparse.eval()
    with torch.no_grad():
        a = dense._forward_batch('train', dense.make_batch('train', idx))
        b = sparse._forward_batch('train', sparse.make_batch('train', idx))
    assert_close(b, a, 'logits sparse vs dense (%s)' % channels, norm_tol=1e-5)
    # the training step against the oracle
    m = dense
    m.train()
    m.zero_grad(set_to_none=True)
    batch = m.make_batch('train', idx)
    out = m.training_step(batch, 0)
    m.backward(None, out['loss'], None, 0)
    params, anchors, ob, ccp = oracle_inputs(m, batch, idx)
    ref_logits = FH.forward(params, hp, 'train', ob, anchors, ccp)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, batch['label'].cpu())
    ref_loss.backward()
    assert_close(out['loss'], ref_loss, 'loss (%s)' % channels)
    checked = 0
    for k, p in m.named_parameters():
        ref = params[k].grad
        if p.grad is None or ref is None:
            assert (p.grad is None or float(p.grad.abs().max()) == 0) and (ref is None or float(ref.abs().max()) == 0), k
            continue
        assert_close(p.grad, ref, 'grad %s (%s)' % (k, channels))
        checked += 1
    assert checked >= 4
    # a recorded step is the eager step
    m.zero_grad(set_to_none=True)
    twin = sparse
    twin.load_state_dict(m.state_dict())
    twin.train()
    o1 = optim.ClipAdam(m.parameters(), 0.01, max_norm=0.5, capturable=True)
    o2 = optim.ClipAdam(twin.parameters(), 0.01, max_norm=0.5, capturable=True)
    cap = CapturedTrainStep(twin, o2, idx.numel(), 0.0, warmup=1)
    for _ in range(3):
        e = m.training_step(m.make_batch('train', idx), 0)
        o1.zero_grad(set_to_none=True)
        m.backward(None, e['loss'], o1, 0)
        o1.step()
        r = cap.replay(idx)[0]
        le = float(e['loss'].detach())
        assert abs(le - float(r)) <= 1e-5 * max(1.0, abs(le)), channels


@pytest.mark.parametrize('over', [
    {'use_mpn_projection': False},
    {'lstm_aggregator': 'sum'},
    {'lstm_n_layers': 2, 'lstm_aggregator': 'sum'},
    {'freeze_node_embeds': True},
    {'n_layers': 3},
    {'n_layers': 1, 'cc_aggregator': 'max', 'trainable_cc': True},
    {'structure_patch_type': 'ego_graph', 'structure_anchor_patch_radius': 1},
    {'linear_hidden_dim_1': 40, 'linear_hidden_dim_2': 12, 'node_embed_size': 8},
], ids=lambda o: '+'.join('%s=%s' % kv for kv in o.items()))
def test_less_common_hyperparameters_train_like_the_oracle(over, tmp_path):
    """Options outside the five g11 variants -- no MPN projection, the LSTM aggregators and depths, frozen node embeddings, 3
    layers, max component aggregation with trainable components, ego-graph structure patches, other head widths -- on the
    density fixture: one training step (logits, loss, every gradient) against the oracle fed the product's prepared state."""
    from conftest import load_golden
    from helpers import oracle_inputs
    from oracle import float_half as FH
    golden = load_golden('density')
    o = dict(over)
    o['lstm_dropout'] = 0.0
    m, _ = _models(golden, tmp_path, o)
    m.prepare_data()
    hp = m.hparams
    idx = torch.arange(min(hp['batch_size'], len(m.train_sub_G)))
    m.train()
    m.zero_grad(set_to_none=True)
    batch = m.make_batch('train', idx)
    out = m.training_step(batch, 0)
    m.backward(None, out['loss'], None, 0)
    params, anchors, ob, ccp = oracle_inputs(m, batch, idx)
    ref_logits = FH.forward(params, hp, 'train', ob, anchors, ccp)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, batch['label'].cpu())
    ref_loss.backward()
    assert_close(out['loss'], ref_loss, 'loss %r' % (over,))
    checked = 0
    for k, p in m.named_parameters():
        ref = params[k].grad
        if k == 'node_embeddings.weight' and over.get('freeze_node_embeds'):
            continue                                        # (frozen in the product; the oracle differentiates everything)
        if p.grad is None or ref is None:
            assert (p.grad is None or float(p.grad.abs().max()) == 0) and (ref is None or float(ref.abs().max()) == 0), k
            continue
        assert_close(p.grad, ref, 'grad %s %r' % (k, over))
        checked += 1
    assert checked >= 6
    if over.get('freeze_node_embeds'):
        assert m.node_embeddings.weight.grad is None and not m.node_embeddings.weight.requires_grad      # S.py:166-167


def test_cached_similarity_files_are_read_back(tmp_path):
    """compute_similarities = False (the reference's .npy cache, S.py:726-742, 852-873, 893-978): a second model on the same
    dataset directory loads what the first one computed and saved -- border sets, shortest-path and structure similarities,
    structure patches and walks -- and arrives at the same prepared state and the same logits."""
    from conftest import load_golden
    golden = load_golden('density')
    first, second = _models(golden, tmp_path, {'lstm_dropout': 0.0})
    first.prepare_data()
    second.hparams['compute_similarities'] = False
    second.load_state_dict(first.state_dict())
    second.prepare_data()
    for k in ('train_cc_ids', 'val_cc_ids', 'train_N_border', 'train_neigh_pos_similarities', 'val_neigh_pos_similarities',
              'structure_anchors', 'int_structure_anchor_random_walks', 'bor_structure_anchor_random_walks',
              'train_int_struc_similarities', 'train_bor_struc_similarities', 'val_int_struc_similarities'):
        a, b = getattr(first, k), getattr(second, k)
        assert torch.equal(a.to(b.dtype) if a.dtype != b.dtype else a, b), k
    idx = torch.arange(6)
    first.eval(); second.eval()
    with torch.no_grad():
        la = first._forward_batch('train', first.make_batch('train', idx))
        lb = second._forward_batch('train', second.make_batch('train', idx))
    assert torch.equal(la, lb)


def test_a_disconnected_graph_prepares_alike_on_both_paths(tmp_path):
    """A base graph with a second connected component, and subgraphs that take nodes from both: components of one subgraph that
    cannot reach each other, position anchors nobody reaches.  The reference's shortest-path matrix holds 0 for such pairs and its
    row-min runs over those zeros (precompute_graph_metrics.py:20-25, SubGNN.py:772); the dense path (matrix from the GPU metric
    precompute) and the sparse path (multi-source BFS with the same convention) have to agree on every anchor, every similarity
    and the logits."""
    import os
    from conftest import load_golden
    from subgnn_amd import config, hotpath
    from subgnn_amd import precompute_graph_metrics as pgm
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    golden = load_golden('density')
    name = write_dataset_from_golden(golden, tmp_path, with_ego=False)
    d = os.path.join(str(tmp_path), name)
    n0 = int(golden['g1_rowptr'].shape[0]) - 2                     # nodes 0 .. n0 - 1 in file ids
    extra = list(range(n0, n0 + 10))                               # a 10-node path, attached to nothing
    with open(os.path.join(d, 'edge_list.txt'), 'a') as f:
        for a, b in zip(extra[:-1], extra[1:]):
            f.write('%d %d\n' % (a, b))
    emb = torch.load(os.path.join(d, 'gin_embeddings.pth'))
    torch.save(torch.cat([emb, torch.randn(10, emb.shape[1], generator=torch.Generator().manual_seed(1))], 0),
               os.path.join(d, 'gin_embeddings.pth'))
    rows = open(os.path.join(d, 'subgraphs.pth')).read().splitlines()
    out = []
    for i, r in enumerate(rows):
        c = r.split('\t')
        if i % 4 == 0:                                             # every fourth subgraph reaches into the island
            c[0] = c[0] + '-%d-%d' % (extra[(i // 4) % 9], extra[(i // 4) % 9 + 1])
        out.append('\t'.join(c))
    open(os.path.join(d, 'subgraphs.pth'), 'w').write('\n'.join(out) + '\n')
    for fn in ('shortest_path_matrix.npy', 'degree_sequence.txt'):
        os.remove(os.path.join(d, fn))
    pgm.calculate_stats(d, shortest_paths=True, ego=False)
    config.PROJECT_ROOT = tmp_path
    models = []
    for _ in range(2):
        hp = dict(golden.hp)
        hp.update({'seed': golden.seed, 'neigh_sample_border_size': 2, 'lin_dropout': 0.0, 'lstm_dropout': 0.0})
        torch.manual_seed(0)
        models.append(SubGNN(hp, **dataset_paths(name)))
    dense, sparse = models
    dense.prepare_data()
    hotpath.prepare_sparse(sparse, 'train')
    assert dense.train_cc_ids.shape[1] >= 2                        # subgraphs with more than one component exist
    _compare_dense_and_sparse(dense, sparse)
