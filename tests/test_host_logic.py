"""CPU tests of the host-side logic of the product package (no kernels are called)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import graph as OG, integer_half as IH
from helpers import write_dataset_from_golden


def test_networkx_order_csr_matches_reference_graph(golden):
    from subgnn_amd.graph import networkx_order_csr
    rp, col, order = networkx_order_csr(golden['edge_list'])
    assert np.array_equal(rp, golden['g1_rowptr'])
    assert np.array_equal(col, golden['g1_col'])
    assert np.array_equal(order, golden['g1_node_order'])


def test_read_subgraphs_and_state_dict_keys(tmp_path, tiny):
    """The dataset reader and the module's parameter names (checkpoint compatibility)."""
    from subgnn_amd import config
    from subgnn_amd.subgraph_utils import read_subgraphs
    name = write_dataset_from_golden(tiny, tmp_path)
    tr, trl, va, val, te, tel = read_subgraphs(os.path.join(str(tmp_path), name, 'subgraphs.pth'))
    assert [[v + 1 for v in s] for s in tr] == tiny.ragged('subgraphs_train', 0)
    assert np.array_equal(trl.numpy(), tiny['labels_train'])
    assert len(va) == tiny['subgraphs_val'].shape[0] and len(te) == tiny['subgraphs_test'].shape[0]
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    config.PROJECT_ROOT = tmp_path
    hp = dict(tiny.hp)
    if torch.cuda.is_available():
        pytest.skip('construction-only check is for the CPU box')
    m = SubGNN(hp, **dataset_paths(name))
    ref_keys = {k[3:] for k in tiny.files if k.startswith('sd/')}
    assert set(m.state_dict().keys()) == ref_keys
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tiny['sd/' + k].shape, k
    assert m.hid_dim == tiny['sd/lin.weight'].shape[1]


def test_trim_zero_columns(golden):
    from subgnn_amd.subgraph_utils import trim_zero_columns
    idx = golden['g12_idx']
    x = torch.from_numpy(golden['g2_cc_ids_train'][idx])
    assert np.array_equal(trim_zero_columns(x).numpy(), golden['g12_cc_ids'])


def test_tape_constants_match_oracle():
    from subgnn_amd import tape
    from oracle import tape as OT
    for k in dir(OT):
        if k.startswith('STREAM_'):
            assert getattr(tape, k) == getattr(OT, k)
    assert tape.stream_id(5, 'val', 3) == OT.stream_id(5, 'val', 3)
    src = open(os.path.join(os.path.dirname(tape.__file__), 'csrc', 'common.h')).read()
    for const in (OT.K_STREAM, OT.K_ITEM, OT.K_DRAW, OT.M1, OT.M2):
        assert ('0x%016X' % const) in src.upper().replace('ULL', '').replace('0X', '0x') or ('%X' % const) in src.upper()


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(__file__), '..', 'subgnn_amd')
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f


def test_density_dataset_writer_round_trips(tmp_path):
    """The synthetic DENSITY-style writer produces files the loaders read back."""
    from subgnn_amd.prepare_dataset import write_density_dataset
    from subgnn_amd.subgraph_utils import read_subgraphs
    from subgnn_amd.graph import parse_edge_list, networkx_order_csr
    d = write_density_dataset(tmp_path / 'density', n_nodes=300, m=4, n_subgraphs=40, subgraph_nodes=10, embed_dim=8)
    tr, trl, va, val, te, tel = read_subgraphs(d / 'subgraphs.pth')
    assert len(tr) == 32 and len(va) == 4 and len(te) == 4
    assert set(trl.tolist()) <= {0, 1, 2} and all(1 <= len(s) <= 10 for s in tr)      # edits can cut a member off
    edges = parse_edge_list(d / 'edge_list.txt')
    rp, col, order = networkx_order_csr(edges)
    assert 250 <= len(order) <= 300 and rp[-1] == 2 * len(edges)                      # largest component only
    assert torch.load(d / 'gin_embeddings.pth').shape == (len(order), 8)


# ---- synthetic dataset recipes (prepare_dataset.py: same random stream as the reference) -----------------

RECIPE_CASES = ('density', 'cut_ratio', 'coreness', 'cc', 'density_b')


def _recipe_args(kw):
    return dict(n=kw['n'], m=kw['m'], p=kw['p'], q=kw['q'], n_subgraphs=kw['n_subgraphs'],
                n_subgraph_nodes=kw['n_subgraph_nodes'], n_bins=kw['n_bins'], n_components=kw['n_connected_components'])


@pytest.mark.parametrize('case', RECIPE_CASES)
def test_dataset_recipes_replay_the_reference_stream(case):
    """All four recipes against the reference's own runs (tests/golden/recipes.npz: prepare_dataset.py imported and run
    with the global ``random`` seeded): the EDITED graph edge for edge in ``graph.edges()`` order, the node order, the
    subgraph lists, the labels, the 80/10/10 split and the position of the random stream afterwards are identical --
    every ``random.sample`` the reference makes is made here on the same population in the same order."""
    import json
    import warnings
    from subgnn_amd import prepare_dataset as pd
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'recipes.npz'))
    kw = json.loads(str(z[case + '/kwargs']))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        st = pd.generate(kw['desired_property'], seed=int(z['seed']), **_recipe_args(kw))
        mask = pd.split_mask(len(st['labels']), st['rng'])
    G = st['graph']
    assert np.array_equal(np.array(list(G.edges()), dtype=np.int64).reshape(-1, 2), z[case + '/edges'])
    assert np.array_equal(np.array(list(G.nodes()), dtype=np.int64), z[case + '/nodes'])
    want_subs = [[v for v in row if v >= 0] for row in z[case + '/subgraphs'].tolist()]
    assert [list(s) for s in st['subgraphs']] == want_subs
    assert list(st['labels']) == [str(l) for l in z[case + '/labels']]
    assert mask == z[case + '/mask'].tolist()
    assert st['rng'].random() == float(z[case + '/next_random'])          # nothing more, nothing less was drawn
    # the relabelled lists (what write_dataset writes when the edits cut a node off) name the same nodes of the final graph
    assert len(st['relabelled_subgraphs']) == len(want_subs)
    assert all(0 <= v < G.number_of_nodes() for s in st['relabelled_subgraphs'] for v in s)
    if st['identity']:
        assert st['relabelled_subgraphs'] == want_subs


@pytest.mark.parametrize('prop,over', [
    ('density', dict(n=600, n_subgraphs=40)),
    ('cut_ratio', dict(n=600, n_subgraphs=30)),
    ('coreness', dict(n=500, n_subgraphs=6)),
    ('cc', dict(n=300, n_subgraphs=40)),
])
def test_dataset_recipes(prop, over, tmp_path):
    """Each recipe writes a loadable dataset whose labels are the property's bins: densities / cut
    ratios edited towards their targets, component counts as stapled, letters in ascending bin order."""
    import warnings
    import networkx as nx
    from subgnn_amd import prepare_dataset as pd
    from subgnn_amd.subgraph_utils import read_subgraphs
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out, info = pd.write_dataset(tmp_path / prop, prop, seed=5, embed_dim=8, **over)
    G = nx.read_edgelist(str(out / 'edge_list.txt'), nodetype=int)
    assert nx.is_connected(G) and sorted(G.nodes) == list(range(G.number_of_nodes()))
    tr, trl, va, val, te, tel = read_subgraphs(out / 'subgraphs.pth')
    n = len(tr) + len(va) + len(te)
    assert n == info['n_subgraphs'] and abs(len(tr) - 0.8 * n) <= 1 and abs(len(va) - len(te)) <= 1
    assert all(0 <= v < G.number_of_nodes() for s in tr + va + te for v in s)          # ids that exist in the written graph
    assert torch.load(out / 'gin_embeddings.pth').shape == (G.number_of_nodes(), 8)
    vals, labs = np.array(info['values'], dtype=float), np.array(info['labels'])
    assert set(labs) == {chr(65 + i) for i in range(len(set(labs)))} and len(set(labs)) >= 2
    for a, b in zip(sorted(set(labs))[:-1], sorted(set(labs))[1:]):          # bins ascend with the letters
        assert vals[labs == a].max() <= vals[labs == b].min()
    if prop == 'density':
        hit = [min(abs(v - t) for t in pd.DENSITY_RANGE) < pd.DENSITY_EPSILON for v in vals]
        assert np.mean(hit) > 0.05     # later edits and the largest-component cut disturb earlier subgraphs (as upstream: 10 %)
    if prop == 'cut_ratio':
        assert vals.min() > 0 and vals.max() < 0.05
    if prop == 'cc':
        assert set(int(v) for v in vals) <= set(pd.CC_RANGE) and (vals == 1).any() and (vals >= 5).any()
        assert all((l == 'A') == (v == 1) for l, v in zip(labs, vals))
    if prop == 'coreness':
        assert vals.min() >= 1.0


def test_equal_count_bins_and_letters():
    from subgnn_amd import prepare_dataset as pd
    vals = [0.1, 0.5, 0.2, 0.9, 0.3, 0.7]
    cuts = pd.equal_count_bins(vals, 3)
    assert np.allclose(cuts, [0.2, 0.5])
    assert pd.letters(np.digitize(vals, bins=cuts)) == ['A', 'C', 'B', 'C', 'B', 'C']


# ---- extra.npz: caller pins, metrics (8f-4), dataset recipe (8f-3) -------------------------

def _extra():
    from conftest import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, 'extra.npz'), allow_pickle=False)


def test_caller_merged_hparams_match_the_reference(tmp_path):
    """train_config.py:44-86: the same config.json through the reference's read_json +
    get_hyperparams_optuna (fixed trial: first choice / lower bound) and through this repository's."""
    import json
    from subgnn_amd import train_config as TC
    z = _extra()
    cfg = tmp_path / 'config.json'
    cfg.write_text(str(z['caller_config_text']))
    rc = TC.read_json(cfg)
    assert list(rc.keys()) == json.loads(str(z['caller_run_config_keys']))
    hp = TC.get_hyperparams(rc, TC.FixedTrial())
    assert hp == json.loads(str(z['caller_merged_hparams']))
    assert list(hp.keys()) == json.loads(str(z['caller_merged_hparams_order']))


def test_metrics_match_the_reference():
    """subgraph_utils.calc_f1 / calc_accuracy (su:94-124), single- and multi-label, and the epoch metrics of
    validation_epoch_end (S.py:408-464) recomputed from the reference's own per-batch outputs."""
    import json
    import types
    from subgnn_amd import subgraph_utils as su
    from subgnn_amd.SubGNN import SubGNN
    z = _extra()
    logits, labels, ml = torch.from_numpy(z['m_logits']), torch.from_numpy(z['m_labels']), torch.from_numpy(z['ml_labels'])
    for avg in ('macro', 'micro'):
        assert np.allclose(su.calc_f1(logits, labels, avg).numpy(), z['m_f1_' + avg])
        assert np.allclose(su.calc_f1(logits, ml, avg, multilabel_binarizer=object()).numpy(), z['ml_f1_' + avg])
    assert np.allclose(su.calc_accuracy(logits, labels).numpy(), z['m_acc'])
    assert np.allclose(su.calc_accuracy(logits, ml, multilabel_binarizer=object()).numpy(), z['ml_acc'])
    outs = []
    for i in range(int(z['val_n_batches'])):
        pre = 'val_out/%d/' % i
        outs.append({k[len(pre):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(pre)})
    me = types.SimpleNamespace(multilabel=False, multilabel_binarizer=None)
    logs = SubGNN._epoch_metrics(me, outs, 'val')
    want = json.loads(str(z['val_metric_values']))
    assert list(logs.keys()) == json.loads(str(z['val_metric_keys']))
    for k, v in want.items():
        assert abs(float(logs[k]) - v) < 1e-6 or (np.isnan(v) and np.isnan(float(logs[k]))), k


def test_density_recipe_against_the_reference_run():
    """The first fixture of the DENSITY recipe (tests/golden/extra.npz, round 2: the reference's own run with its seeds):
    property function, equal-count bins, letters, base graph -- and, since round 3, the whole run replayed: edited graph,
    subgraphs, labels, split (the other recipes and a second density case: test_dataset_recipes_replay_the_reference_stream)."""
    import json
    import warnings
    import networkx as nx
    from subgnn_amd import prepare_dataset as pd
    z = _extra()
    P = json.loads(str(z['recipe_params']))
    subs = [[int(v) for v in row if v != -1] for row in z['recipe_subgraphs']]
    G = nx.Graph()
    G.add_nodes_from(int(v) for v in z['recipe_final_nodes'])
    G.add_edges_from((int(u), int(v)) for u, v in z['recipe_final_edges'])
    dens = [pd.density(G, s) for s in subs]
    assert np.allclose(dens, z['recipe_density'], rtol=0, atol=0)
    labels = pd.letters(np.digitize(dens, bins=pd.equal_count_bins(dens, P['n_bins'])))
    assert labels == [str(l) for l in z['recipe_labels']]
    base = nx.barabasi_albert_graph(P['n'], P['m'], seed=P['seed'])
    assert sorted(tuple(sorted(e)) for e in base.edges()) == [tuple(e) for e in z['recipe_base_edges'].tolist()]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        st = pd.generate('density', seed=P['seed'], n=P['n'], m=P['m'], n_subgraphs=P['n_subgraphs'],
                         n_subgraph_nodes=P['n_subgraph_nodes'], n_bins=P['n_bins'])
        mask = pd.split_mask(len(st['labels']), st['rng'])
    assert [list(s) for s in st['subgraphs']] == subs
    assert np.array_equal(np.array(list(st['graph'].edges()), dtype=np.int64), z['recipe_final_edges'])
    assert np.array_equal(np.array(list(st['graph'].nodes()), dtype=np.int64), z['recipe_final_nodes'])
    assert list(st['labels']) == [str(l) for l in z['recipe_labels']]
    assert np.array_equal(np.array(st['values']), z['recipe_density'])
    assert mask == z['recipe_mask'].tolist()


def test_block_rows_prefers_a_divisor_near_the_wanted_block():
    """ops._block_rows: the row-block size of a split contraction divides the row count when a divisor exists between
    want / 2 and 2 want (no remainder GEMM), and is the divisor nearest to ``want``."""
    from subgnn_amd import ops
    assert ops._block_rows(50000) == 1000 and 50000 % 1000 == 0
    assert ops._block_rows(16800) in (1050, 1120, 1200, 840, 800) and 16800 % ops._block_rows(16800) == 0
    assert abs(ops._block_rows(16800) - 1024) == min(abs(d - 1024) for d in range(512, 2049) if 16800 % d == 0)
    assert ops._block_rows(1024 * 7) == 1024
    p = 100003                                   # a prime: no divisor in range
    assert ops._block_rows(p) == 1024


def test_copy_into_keeps_addresses_and_reports_what_it_had_to_replace():
    """hotpath._copy_into (install_pass_static): tensors of equal shape / dtype are copied INTO the kept tensors (the
    addresses a recorded training half reads stay valid), attributes hung on tensors follow, a shape change is replaced
    and reported, storage seen twice is copied once."""
    from subgnn_amd import hotpath
    kept_a, kept_b = torch.zeros(4, 3), torch.zeros(5, dtype=torch.int64)
    kept_a._sgnn_ids32 = torch.zeros(12, dtype=torch.int32)
    dst = {'a': kept_a, 'nest': {0: (kept_b, None, 7)}, 'gone': torch.zeros(2)}
    new_a = torch.arange(12.).view(4, 3)
    new_a._sgnn_ids32 = torch.arange(12, dtype=torch.int32)
    new_b = torch.arange(5)
    src = {'a': new_a, 'nest': {0: (new_b, None, 8)}, 'gone': torch.zeros(3), 'alias': new_a}
    replaced = []
    out = hotpath._copy_into(dst, src, 'root', replaced)
    assert out is dst and out['a'] is kept_a and torch.equal(kept_a, new_a)
    assert out['a']._sgnn_ids32.data_ptr() == kept_a._sgnn_ids32.data_ptr() and torch.equal(kept_a._sgnn_ids32, new_a._sgnn_ids32)
    assert out['nest'][0][0] is kept_b and torch.equal(kept_b, new_b) and out['nest'][0][2] == 8
    assert out['gone'].shape == (3,) and any("gone" in r for r in replaced)
    assert out['alias'] is new_a and any('alias' in r for r in replaced)      # nothing kept under that name yet: replaced
    assert len(replaced) == 2


def test_cpu_baseline_worker_pool_equals_serial():
    """oracle/cpu_baseline.py deals its Python stages to worker processes (forked before the GPU is initialised in bench.py):
    the pooled map keeps the order and the values of the serial one."""
    from subgnn_amd import synthetic
    from oracle import cpu_baseline as cb
    n = 3000
    rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, 4, seed=1), n)
    subs = synthetic.bfs_subgraphs(rowptr, col, 40, 8, seed=2)
    cb.stop_pool()
    cb._W['G'] = cb.CSRGraph(rowptr, col)
    serial = [cb._w_components(s) for s in subs]
    assert cb._pmap(cb._w_components, subs) == serial            # no pool: in this process
    try:
        assert cb.start_pool(rowptr, col, procs=3)[1] == 3
        assert cb._pmap(cb._w_components, subs) == serial
        rows = [np.asarray(s, dtype=np.int64) for s in subs]
        got = cb._pmap(cb._w_border, rows)
        want = [cb._w_border(r) for r in rows]
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    finally:
        cb.stop_pool()
    assert cb._POOL is None


def test_bench_projection_from_a_measured_step():
    """bench.projection: 8-GPU estimates from the N = 1 step, marked as not measured; the weak form gains what the exposed
    reduce-scatter costs and the owner-computes Adam saves."""
    import types
    import bench
    table = torch.nn.Parameter(torch.zeros(1_000_001, 64))
    model = types.SimpleNamespace(node_embeddings=types.SimpleNamespace(weight=table),
                                  parameters=lambda: [table, torch.nn.Parameter(torch.zeros(10, 10))])
    res = {'ms_per_step': 10.0, 'stages_ms': {'optimizer': 0.8}}
    p = bench.projection(types.SimpleNamespace(), res, model, 50_000)
    assert p['not_a_measurement'] and p['n_gpus'] == 8
    assert p['collective_bytes_per_step']['table_gradient_reduce_scatter'] == 1_000_001 * 64 * 4
    assert p['collective_bytes_per_step']['small_gradients_all_reduce'] == 400
    d, r = p['weak_direct'], p['weak_ring']
    assert d['ms_per_step'] < r['ms_per_step'] and 7.0 < r['speedup_vs_1gpu'] <= d['speedup_vs_1gpu'] <= 8.8
    assert abs(d['ms_per_step'] - (10.0 + p['reduce_scatter_ms']['direct_7_links'] + 0.1 - 0.7)) < 0.02
    assert abs(p['reduce_scatter_ms']['ring'] - 7 * p['reduce_scatter_ms']['direct_7_links']) < 0.01


def test_trainer_optimizer_swap_only_for_plain_adam_on_the_device():
    """optim.accelerate (what train_config.Trainer steps with): anything but a fresh plain torch.optim.Adam over CUDA parameters
    is handed back untouched -- CPU parameters, another optimizer class, weight decay, a second parameter group, existing state."""
    import torch
    from subgnn_amd import optim
    p = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(3))]
    for opt in (torch.optim.Adam(p, lr=0.1), torch.optim.SGD(p, lr=0.1), torch.optim.Adam(p, lr=0.1, weight_decay=0.01),
                torch.optim.Adam([{'params': p[:1]}, {'params': p[1:]}], lr=0.1), torch.optim.AdamW(p, lr=0.1)):
        assert optim.accelerate(opt, 0.5, capturable=True) is opt
    used = torch.optim.Adam(p, lr=0.1)
    p[0].grad = torch.ones_like(p[0])
    used.step()
    assert optim.accelerate(used, 0.5) is used                  # (state exists: the swap would drop it)


def test_batched_row_contraction_matches_einsum():
    import torch
    from subgnn_amd import ops
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(2, 37, 5, generator=g), torch.randn(2, 37, 3, generator=g)
    assert torch.allclose(ops.contract_rows_batched(a, b), torch.einsum('grm,grn->gmn', a, b), atol=1e-5)


def test_kernel_count_is_none_without_a_device():
    import torch
    from subgnn_amd import standins
    if not torch.cuda.is_available():
        assert standins.count_kernels(lambda: None) is None


def test_f1_from_predictions_is_sklearns_f1_score():
    """su.calc_f1 evaluates micro / macro F1 from the confusion counts (subgraph_utils.f1_from_predictions) instead of calling
    sklearn per validation batch: same values, including the classes-without-predictions case sklearn warns about."""
    import warnings
    from sklearn.metrics import f1_score
    from subgnn_amd.subgraph_utils import f1_from_predictions, calc_f1
    rng = np.random.default_rng(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for trial in range(120):
            n, K = int(rng.integers(1, 80)), int(rng.integers(2, 8))
            yt = rng.integers(0, K, n)
            yp = rng.integers(0, K, n) if trial % 3 else rng.integers(0, 2, n)
            Yt = rng.integers(0, 2, (n, K))
            Yp = rng.integers(0, 2, (n, K)) if trial % 4 else np.zeros((n, K), dtype=np.int64)
            for avg in ('micro', 'macro'):
                assert abs(f1_from_predictions(yt, yp, avg) - f1_score(yt, yp, average=avg)) < 1e-12
                assert abs(f1_from_predictions(Yt, Yp, avg) - f1_score(Yt, Yp, average=avg)) < 1e-12
        logits = torch.randn(40, 5, generator=torch.Generator().manual_seed(1))
        labels = torch.randint(0, 5, (40,), generator=torch.Generator().manual_seed(2))
        for avg in ('micro', 'macro'):
            got = calc_f1(logits, labels, avg)
            assert got.dtype == torch.float32 and abs(float(got) - f1_score(labels, logits.argmax(1), average=avg)) < 1e-6


def test_roc_auc_is_sklearns_roc_auc_score():
    """The epoch-end AUROCs (S.py:408-444: one sklearn call for the average and one per class) come from
    subgraph_utils.roc_auc: same values as this image's sklearn -- ties in the scores, binary / one-vs-rest / multilabel,
    nan for a one-class binary problem, ValueError where the class count differs from the score columns (which
    _epoch_metrics turns into nan)."""
    import warnings
    from sklearn.metrics import roc_auc_score
    from subgnn_amd.subgraph_utils import roc_auc
    rng = np.random.default_rng(5)

    def both(f, *a, **k):
        try:
            return f(*a, **k)
        except ValueError:
            return 'ValueError'

    def same(a, b):
        if isinstance(a, str) or isinstance(b, str):
            return a == b
        return (np.isnan(a) and np.isnan(b)) or abs(a - b) < 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for trial in range(150):
            n, K = int(rng.integers(2, 120)), int(rng.integers(3, 7))    # (two score columns are sklearn's binary case: 1-d scores)
            score = rng.normal(size=n).astype(np.float32)
            if trial % 3 == 0:
                score = np.round(score * 2) / 2                                        # many tied scores
            yb = rng.integers(0, 2, n) if trial % 7 else np.zeros(n, dtype=np.int64)  # (one class only: nan)
            a, b = both(roc_auc, yb, score), both(roc_auc_score, yb, score)
            assert same(a, b), (trial, a, b)
            logits = rng.normal(size=(n, K)).astype(np.float32)
            if trial % 4 == 0:
                logits = np.round(logits)
            e = np.exp(logits - logits.max(1, keepdims=True))
            prob = (e / e.sum(1, keepdims=True))
            ym = rng.integers(0, K, n)                                                 # (small n: a class may be absent: ValueError)
            a, b = both(roc_auc, ym, prob, multi_class='ovr'), both(roc_auc_score, ym, prob, multi_class='ovr')
            assert same(a, b), (trial, a, b)
            Y = rng.integers(0, 2, (n, K))
            sig = 1.0 / (1.0 + np.exp(-logits))
            a, b = both(roc_auc, Y, sig, multi_class='ovr'), both(roc_auc_score, Y, sig, multi_class='ovr')
            assert same(a, b), (trial, a, b)
