"""-m gpu: the build-owned counterpart of train_config.py drives the drop-in module end to end from a
reference-format config.json (with // comments and an optuna block), and training reduces the loss."""
import json

import numpy as np
import os

import pytest
import torch

from helpers import write_dataset_from_golden

pytestmark = pytest.mark.gpu

CONFIG = '''{
    "data": {"task": "ds"},
    //"no_gpu": true,
    "tb": {"tb_logging": false, "dir": "tensorboard", "name": "x"},
    "optuna": {"opt_n_trials": 1, "opt_n_cores": 1, "monitor_metric": "val_micro_f1", "opt_direction": "maximize",
               "sampler": "random", "pruning": false},
    "hyperparams_fix": %s,
    "hyperparams_optuna": {
        "batch_size": {"type": "suggest_categorical", "args": [[8, 16]]},
        "learning_rate": {"type": "suggest_float", "args": [5e-3, 1e-2], "kwargs": {"log": true}},
        "grad_clip": {"type": "suggest_float", "args": [0.5, 1.0]},
        "n_layers": {"type": "suggest_int", "args": [1, 2]}
    }
}'''


def test_train_driver_end_to_end(tiny, tmp_path):
    from subgnn_amd import config, train_config
    write_dataset_from_golden(tiny, tmp_path, 'ds')
    fix = dict(tiny.hp)
    for k in ('batch_size', 'learning_rate', 'n_layers'):
        fix.pop(k, None)
    fix.update({'max_epochs': 6, 'seed': 3, 'lin_dropout': 0.0, 'compute_similarities': True})
    cfg = tmp_path / 'config.json'
    cfg.write_text(CONFIG % json.dumps(fix))
    config.PROJECT_ROOT = tmp_path
    rc = train_config.read_json(cfg)
    assert rc['optuna']['monitor_metric'] == 'val_micro_f1' and 'no_gpu' not in rc
    best, model, trainer = train_config.train_model(rc, results_dir=tmp_path / 'results', log=lambda *a: None)
    assert model.hparams['batch_size'] == 8 and model.hparams['n_layers'] == 1
    assert len(model.metric_scores) == 6 and 0.0 <= best <= 1.0
    assert {'val_loss', 'val_micro_f1', 'val_macro_f1', 'val_acc', 'avg_val_acc', 'avg_macro_f1', 'val_auroc'} <= set(model.metric_scores[-1])
    h = trainer.history
    assert all(torch.isfinite(torch.tensor(e['train_loss'])) for e in h)
    assert h[-1]['train_loss'] < h[0]['train_loss']                 # it learns the 10 training subgraphs
    assert os.path.exists(tmp_path / 'results' / 'final_metric_scores.json')
    assert os.path.exists(tmp_path / 'results' / 'hyperparams.json')
    # the similarity cache was written with the reference's file names
    sim = tmp_path / 'ds' / 'similarities'
    names = set(os.listdir(sim))
    assert '0_train_similarities.npy' in names and '2_0_train_border_set.npy' in names
    assert any(n.startswith('int_struc_12_triangular_random_walk_2_0_train') for n in names)
    res = trainer.test(model)
    assert 'test_micro_f1' in model.test_results and torch.isfinite(res['avg_test_loss'])


def test_seed_sweep_driver(tiny, tmp_path):
    """SubGNN/test.py: one fresh model per seed, test metrics aggregated into experiment_results.json."""
    from subgnn_amd import config, train_config
    from subgnn_amd import test as sweep
    write_dataset_from_golden(tiny, tmp_path, 'ds')
    fix = dict(tiny.hp)
    for k in ('batch_size', 'learning_rate', 'n_layers'):
        fix.pop(k, None)
    fix.update({'max_epochs': 2, 'lin_dropout': 0.0, 'compute_similarities': True})
    cfg = tmp_path / 'config.json'
    cfg.write_text(CONFIG % json.dumps(fix))
    config.PROJECT_ROOT = tmp_path
    exp = sweep.run_seeds(train_config.read_json(cfg), n_seeds=3, results_dir=tmp_path / 'sweep', log=lambda *a: None)
    assert len(exp['test_acc']) == 3 and len(exp['test_micro_f1']) == 3 and len(exp['test_auroc']) == 3
    assert abs(exp['test_acc_mean'] - sum(exp['test_acc']) / 3) < 1e-12 and exp['test_micro_f1_sd'] >= 0
    saved = json.loads((tmp_path / 'sweep' / 'experiment_results.json').read_text())
    assert saved['test_micro_f1'] == exp['test_micro_f1']
    assert os.path.exists(tmp_path / 'sweep' / 'version_2' / 'final_metric_scores.json')


def test_component_recipe_through_the_model(tiny, tmp_path):
    """A generated COMPONENT dataset (stapled components) read, prepared and trained by the drop-in
    module: the components the HIP union-find finds per subgraph are the ones the generator stapled."""
    import networkx as nx
    from subgnn_amd import config, train_config, prepare_dataset as pd, precompute_graph_metrics as pgm
    from subgnn_amd.subgraph_utils import read_subgraphs
    out, info = pd.write_dataset(tmp_path / 'ds', 'cc', seed=9, embed_dim=16, n=250, n_subgraphs=24, n_subgraph_nodes=6)
    pgm.calculate_stats(out)
    fix = dict(tiny.hp)
    for k in ('batch_size', 'learning_rate', 'n_layers'):
        fix.pop(k, None)
    fix.update({'max_epochs': 1, 'seed': 1, 'lin_dropout': 0.0, 'compute_similarities': True, 'node_embed_size': 16})
    cfg = tmp_path / 'config.json'
    cfg.write_text(CONFIG % json.dumps(fix))
    config.PROJECT_ROOT = tmp_path
    best, model, trainer = train_config.train_model(train_config.read_json(cfg), log=lambda *a: None)
    G = nx.read_edgelist(str(out / 'edge_list.txt'), nodetype=int)
    tr = read_subgraphs(out / 'subgraphs.pth')[0]
    got = (model.train_cc_ids[:, :, 0] != 0).sum(dim=1).cpu().tolist()
    assert got == [nx.number_connected_components(G.subgraph(s)) for s in tr]
    assert model.num_classes == 2 and torch.isfinite(torch.tensor(trainer.history[0]['train_loss']))



def test_one_trainer_step_matches_the_reference(tiny, tmp_path):
    """SURVEY 8(b) caller pin: the parameters after ONE trainer step (training_step -> backward ->
    clip_grad_norm_ -> Adam.step, the PL 0.7.x order) on g11's 'sum' case, against the imported reference's."""
    import os
    from conftest import GOLDEN_DIR
    from helpers import T, assert_close
    from test_gpu_model import _model, _inject
    from subgnn_amd import train_config as TC
    z = np.load(os.path.join(GOLDEN_DIR, 'extra.npz'), allow_pickle=False)
    g, t = tiny, 'g11_sum/'
    hp = json.loads(str(g[t + 'hparams']))
    m = _model(g, tmp_path, hp)
    sd = {k[len(t) + 3:]: T(g[k]) for k in g.files if k.startswith(t + 'sd/')}
    m.load_state_dict({k: v for k, v in sd.items() if not k.startswith('train_')}, strict=False)
    _inject(m, g, t, m.hparams)
    m.train()
    trainer = TC.Trainer(1, gradient_clip_val=float(z['step_clip']))
    opt = m.configure_optimizers()
    loss = trainer._eager_step(m, opt, m.make_batch('train', g[t + 'idx']), 0)
    assert_close(loss, z['step_loss'], 'loss')
    post = m.state_dict()
    n = 0
    for k in z.files:
        if k.startswith('step_post/'):
            name = k[len('step_post/'):]
            if name in post and post[name].dtype == torch.float32:
                assert_close(post[name], z[k], 'post-step ' + name, 1e-4)
                n += 1
    assert n > 20


def test_a_multilabel_dataset_runs_through_fit_and_test(tmp_path):
    """Several labels per subgraph (HPO-NEURO's shape: su:24-92 reads ``label-label``, the loss becomes BCE with logits and the
    metrics sigmoid-thresholded, SubGNN.py:133, su:90-124): the density fixture with every third label turned into a label pair,
    through prepare_data, two epochs of the Trainer, prepare_test_data and Trainer.test -- eagerly and with recorded training /
    validation steps, which have to agree (the recorded validation loss to float32 summation order)."""
    from conftest import load_golden
    from subgnn_amd import config
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    from subgnn_amd.train_config import Trainer
    golden = load_golden('density')
    name = write_dataset_from_golden(golden, tmp_path, with_ego=False)
    f = os.path.join(str(tmp_path), name, 'subgraphs.pth')
    rows = open(f).read().splitlines()
    out = []
    for i, r in enumerate(rows):
        c = r.split('\t')
        lab = int(c[1])
        c[1] = '%d-%d' % (lab, (lab + 1) % 3) if i % 3 == 0 else str(lab)
        out.append('\t'.join(c))
    open(f, 'w').write('\n'.join(out) + '\n')
    config.PROJECT_ROOT = tmp_path
    runs = {}
    for graph_step in (False, True):
        hp = dict(golden.hp)
        hp.update({'seed': golden.seed, 'neigh_sample_border_size': 2, 'lin_dropout': 0.0, 'lstm_dropout': 0.0})
        torch.manual_seed(0)
        m = SubGNN(hp, **dataset_paths(name))
        assert m.multilabel and isinstance(m.loss, torch.nn.BCEWithLogitsLoss)
        tr = Trainer(2, hp.get('grad_clip', 0.0), log=lambda *a, **k: None, hip_graph_step=graph_step)
        torch.manual_seed(5)
        tr.fit(m)
        m.prepare_test_data()
        t = tr.test(m)
        runs[graph_step] = [v for h in tr.history for v in (h['train_loss'], h['val_loss'])] + \
            [float(t['log'][k]) for k in ('test_loss', 'test_micro_f1', 'test_macro_f1', 'test_auroc')]
        assert all(np.isfinite(runs[graph_step]))
        assert runs[graph_step][2] < runs[graph_step][0]                  # the training loss went down
    for a, b in zip(runs[False], runs[True]):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), runs
