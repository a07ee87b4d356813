#!/usr/bin/env python3
"""More fixtures produced by IMPORTING the reference (build container only), next to make_goldens.py
(whose three files stay byte-identical: this script writes its own ``extra.npz``).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_extra.py

  ego     structure anchor patches with structure_patch_type == 'ego_graph'
          (SubGNN/anchor_patch_samplers.py:226-228), radius 1 and 2, and the walks over them;
  caller  the pins of SURVEY.md section 8(b): the merged hyper-parameter dict
          SubGNN/train_config.py:81-86 builds from a config.json for fixed trial values (the config text is
          written HERE, in the reference's schema -- it is not a reference file), the parameters after ONE
          trainer step on g11's 'sum' case (training_step -> backward -> clip_grad_norm_ -> Adam.step, the
          PL 0.7.x hook order), and model.metric_scores[-1] after one validation epoch together with
          the per-batch outputs it was computed from;
  metrics SubGNN/subgraph_utils.py:94-124 calc_f1 / calc_accuracy on fixed logits, single- and multi-label;
  recipe  prepare_dataset/prepare_dataset.py on the DENSITY recipe at a small size: base graph, the graph
          after the density edits, subgraphs, property values, bins, labels, split mask.

Stand-ins: tests/golden/_standins (README there); ``train_node_emb`` (PyG pre-training) is an empty module.
No reference source text is stored: arrays and the JSON of hyper-parameter dicts only.
"""
import io
import json
import os
import random as pyrandom
import sys
import tempfile
import types
import warnings
from contextlib import redirect_stdout
from pathlib import Path

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
warnings.simplefilter('ignore')

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens as MG            # noqa: E402  (installs the tape shims into the imported reference)

import numpy as np                   # noqa: E402
import networkx as nx                # noqa: E402
import torch                         # noqa: E402

S, aps, su, refconfig = MG.S, MG.aps, MG.su, MG.refconfig
T = MG.T

CALLER_CONFIG = """{
    // written for the fixture, in the schema of config_files/README.md
    "data": {"task": "tiny"},
    "tb": {"dir_full": "tb", "name": "fixture"},
    "optuna": {"opt_n_trials": 1, "opt_n_cores": 1, "monitor_metric": "val_micro_f1", "opt_direction": "maximize",
               "sampler": "grid", "pruning": false},
    "hyperparams_fix": {
        "max_epochs": 2, "seed": 7, "use_neighborhood": true, "use_structure": true, "use_position": true,
        "node_embed_size": 8, "structure_patch_type": "triangular_random_walk", "lstm_aggregator": "last",
        "n_processes": 2, "resample_anchor_patches": false, "freeze_node_embeds": false, "use_mpn_projection": true,
        "compute_similarities": true, "sample_walk_len": 12, "n_triangular_walks": 3, "random_walk_len": 6,
        "rw_beta": 0.65, "neigh_sample_border_size": 2, "n_anchor_patches_pos_out": 7, "n_anchor_patches_pos_in": 5,
        "n_anchor_patches_N_in": 4, "n_anchor_patches_N_out": 6, "n_anchor_patches_structure": 5,
        "linear_hidden_dim_1": 16, "linear_hidden_dim_2": 8, "lstm_dropout": 0.0, "lstm_n_layers": 1,
        "cc_aggregator": "sum", "trainable_cc": false, "max_sim_epochs": 2, "embedding_type": "gin"
    },
    "hyperparams_optuna": {
        "batch_size": {"type": "suggest_categorical", "args": [[6, 4]]},
        "learning_rate": {"type": "suggest_float", "args": [0.001, 0.01], "kwargs": {"log": true}},
        "n_layers": {"type": "suggest_int", "args": [2, 3]},
        "lin_dropout": {"type": "suggest_float", "args": [0.0, 0.5]},
        "grad_clip": {"type": "suggest_float", "args": [0.5, 1.0]}
    }
}
"""


class FixedTrial:
    """The first categorical choice, the lower bound of every range."""

    def suggest_categorical(self, name, choices):
        return choices[0]

    def suggest_float(self, name, low, high, **kw):
        return low

    def suggest_int(self, name, low, high, **kw):
        return low


def ego_goldens(root, out):
    for radius in (1, 2):
        hp = dict(MG.BASE_HP)
        hp.update({'structure_patch_type': 'ego_graph', 'structure_anchor_patch_radius': radius, 'seed': 0})
        model = MG.build_model('tiny', hp)
        G = model.networkx_graph
        MG.CTX.seed = MG.SEED
        sa = aps.sample_structure_anchor_patches(hp, G, model.device, hp['max_sim_epochs'])
        t = 'ego_r%d_' % radius
        out[t + 'structure_anchors'] = MG.t2n(sa)
        out[t + 'hparams'] = np.array(json.dumps(hp))
        out[t + 'int_rw'] = MG.t2n(aps.perform_random_walks(hp, G, sa, True))
        out[t + 'views_int'] = MG.ragged_pad(MG.CTX.record.get('walk_views_bor_int', []))
        out[t + 'bor_rw'] = MG.t2n(aps.perform_random_walks(hp, G, sa, False))
        out[t + 'in_border'] = MG.ragged_pad([list(x) for x in MG.CTX.record['walk_in_border_bor']])
        out[t + 'views_bor'] = MG.ragged_pad([list(x) for x in MG.CTX.record['walk_views_bor_bor']])


def caller_goldens(root, out):
    sys.path.insert(0, str(MG.REF / 'SubGNN'))
    import train_config as TC                       # reference module; optuna / commentjson / PL are stand-ins
    cfg = Path(root) / 'caller_config.json'
    cfg.write_text(CALLER_CONFIG)
    rc = TC.read_json(str(cfg))
    hyp = TC.get_hyperparams_optuna(rc, FixedTrial())
    out['caller_config_text'] = np.array(CALLER_CONFIG)
    out['caller_run_config_keys'] = np.array(json.dumps(list(rc.keys())))
    out['caller_merged_hparams'] = np.array(json.dumps(hyp))
    out['caller_merged_hparams_order'] = np.array(json.dumps(list(hyp.keys())))
    # ---- one trainer step on g11's 'sum' case (same build as make_goldens.forward_goldens) ----------------
    hp = dict(MG.BASE_HP)
    model = MG.build_model('tiny', hp, seed=3)
    MG.CTX.seed = MG.SEED
    model.prepare_data()
    for sp in ('train', 'val'):
        for l in range(hp['n_layers']):
            model.anchors_neigh_int[sp][l] = model.anchors_neigh_int[sp][l].contiguous()
            model.anchors_neigh_border[sp][l] = model.anchors_neigh_border[sp][l].contiguous()
    ds = S.SubgraphDataset(model.train_sub_G, model.train_sub_G_label, model.train_cc_ids, model.train_N_border,
                           model.train_neigh_pos_similarities, model.train_int_struc_similarities,
                           model.train_bor_struc_similarities, model.multilabel, model.multilabel_binarizer)
    idxs = [7, 2, 5, 0, 9, 3]
    batch = model._pad_collate([ds[i] for i in idxs])
    model.train()
    opt = model.configure_optimizers()
    opt.zero_grad()
    res = model.training_step(batch, 0)
    model.backward(None, res['loss'], opt, 0)
    clip = 0.5
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    opt.step()
    ref = np.load(HERE / 'tiny.npz')
    assert np.allclose(MG.t2n(res['loss']), ref['g11_sum/loss']), 'not the g11 sum case any more'
    out['step_clip'] = np.array(clip)
    out['step_loss'] = MG.t2n(res['loss'])
    out['step_grad_norm'] = MG.t2n(total)
    for k, v in model.state_dict().items():
        out['step_post/' + k] = MG.t2n(v)
    # ---- one validation epoch with the stepped model ----------------------------------------------------
    model.eval()
    outputs = []
    with torch.no_grad():
        for i, b in enumerate(model.val_dataloader()):
            outputs.append(model.validation_step(b, i))
    for i, o in enumerate(outputs):
        for k, v in o.items():
            out['val_out/%d/%s' % (i, k)] = MG.t2n(torch.as_tensor(v))
    out['val_n_batches'] = np.array(len(outputs))
    buf = io.StringIO()
    with redirect_stdout(buf):
        ret = model.validation_epoch_end(outputs)
    scores = model.metric_scores[-1]
    out['val_metric_keys'] = np.array(json.dumps(list(scores.keys())))
    out['val_metric_values'] = np.array(json.dumps({k: float(v) for k, v in scores.items()}))
    out['val_epoch_end_keys'] = np.array(json.dumps(sorted(ret.keys())))
    out['val_labels_all'] = MG.t2n(model.val_sub_G_label)


def metric_goldens(out):
    from sklearn.preprocessing import MultiLabelBinarizer
    g = torch.Generator().manual_seed(12)
    logits = torch.randn(40, 4, generator=g)
    labels = torch.randint(0, 4, (40,), generator=g)
    out['m_logits'] = MG.t2n(logits)
    out['m_labels'] = MG.t2n(labels)
    for avg in ('macro', 'micro'):
        out['m_f1_' + avg] = MG.t2n(su.calc_f1(logits, labels, avg_type=avg))
    out['m_acc'] = MG.t2n(su.calc_accuracy(logits, labels))
    mlb = MultiLabelBinarizer().fit([[0, 1, 2, 3]])
    ml = (torch.rand(40, 4, generator=g) < 0.4).long()
    out['ml_labels'] = MG.t2n(ml)
    for avg in ('macro', 'micro'):
        out['ml_f1_' + avg] = MG.t2n(su.calc_f1(logits, ml, avg_type=avg, multilabel_binarizer=mlb))
    out['ml_acc'] = MG.t2n(su.calc_accuracy(logits, ml, multilabel_binarizer=mlb))


def recipe_goldens(root, out):
    fake = types.ModuleType('config')
    fake.PROJECT_ROOT = Path(root) / 'recipe_root'
    fake.PAD_VALUE = 0
    saved = sys.modules.get('config')
    sys.modules['config'] = fake                                  # config_prepare_dataset creates DATASET_DIR at import
    sys.modules['train_node_emb'] = types.ModuleType('train_node_emb')
    sys.path.insert(0, str(MG.REF / 'prepare_dataset'))
    try:
        import prepare_dataset as PD
        import config_prepare_dataset as C
        n, m, ns, k, bins = 300, 4, 40, 12, 3
        pyrandom.seed(C.RANDOM_SEED)
        np.random.seed(C.RANDOM_SEED)
        buf = io.StringIO()
        with redirect_stdout(buf):
            sg = PD.SyntheticGraph(base_graph_type='barabasi_albert', subgraph_type='bfs', n_subgraphs=ns,
                                   n_connected_components=1, n_subgraph_nodes=k, features_type='one_hot', n=n, p=0.5,
                                   q=0, m=m, n_bins=bins, subgraph_generator='complete',
                                   modify_graph_for_properties=True, desired_property='density')
            mask = PD.generate_mask(len(sg.subgraph_labels))
        base = nx.barabasi_albert_graph(n, m, seed=C.RANDOM_SEED)
        out['recipe_params'] = np.array(json.dumps(dict(n=n, m=m, n_subgraphs=ns, n_subgraph_nodes=k, n_bins=bins,
                                                        seed=C.RANDOM_SEED, density_range=C.DENSITY_RANGE,
                                                        density_epsilon=C.DENSITY_EPSILON, max_tries=C.MAX_TRIES)))
        out['recipe_base_edges'] = np.array(sorted(tuple(sorted(e)) for e in base.edges()), dtype=np.int64)
        out['recipe_final_edges'] = np.array(list(sg.graph.edges()), dtype=np.int64)
        out['recipe_final_nodes'] = np.array(list(sg.graph.nodes()), dtype=np.int64)
        out['recipe_subgraphs'] = MG.ragged_pad([list(s) for s in sg.subgraphs])
        out['recipe_labels'] = np.array([str(l) for l in sg.subgraph_labels])
        out['recipe_density'] = np.array([nx.density(sg.graph.subgraph(s)) for s in sg.subgraphs], dtype=np.float64)
        out['recipe_mask'] = np.array(mask, dtype=np.int64)
    finally:
        if saved is not None:
            sys.modules['config'] = saved


def main():
    root = tempfile.mkdtemp(prefix='subgnn_golden_extra_')
    refconfig.PROJECT_ROOT = Path(root)
    rng2 = np.random.default_rng(77)
    edges, subgraphs, labels, splits = MG.make_tiny(rng2)
    MG.write_dataset(root, 'tiny', edges, subgraphs, labels, splits, 8, False, rng2)
    out = {'seed': np.array(MG.SEED)}
    ego_goldens(root, out)
    caller_goldens(root, out)
    metric_goldens(out)
    recipe_goldens(root, out)
    np.savez_compressed(HERE / 'extra.npz', **out)
    print('extra written:', len(out), 'arrays,', (HERE / 'extra.npz').stat().st_size // 1024, 'KiB')


if __name__ == '__main__':
    main()
