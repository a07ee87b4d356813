#!/usr/bin/env python3
"""Stand-in benchmarks for BASELINE.json configs[2] and configs[4] (the real datasets are not
available offline; graph statistics as quoted in SURVEY.md section 8, every line labelled stand-in):

  (density_n and ppi_bp: configs[0] and configs[1], same driver; presets in subgnn_amd/standins.py)
  hpo_metab  "HPO-METAB, all three channels, 1 x MI355X, structure-channel DTW gamma kernel
             stressed": BA n=14 587, m=222 (~3.2 M edges, mean degree ~440), 2 400 subgraphs of ~14
             nodes in 1-2 components; best_model_hyperparameters/hpo_metab/hyperparams.json with the
             N and P channels switched on too, max_sim_epochs 5 => 5*18*4 = 360 structure patches.
             Dense reference-shaped prepare_data from the reference's file formats.
  em_user    "EM-USER (large components, border channel heavy)": BA n=57 333, m=80 (~4.6 M edges),
             324 subgraphs of ~155 nodes in ~52 components (two large ones + singletons);
             best_model_hyperparameters/em_user/hyperparams.json (k = 2 border, N 16/32, sum,
             trainable_cc, batch 32) with all three channels on.  The dense (N, N) float64 hop matrix
             would be 26 GB, so this one uses hotpath.prepare_sparse.

Prints one JSON line: prepare_data seconds (and its stage split for the sparse path), ms per training
step (fwd + bwd + clip + Adam) eager and replayed from a hipGraph.

    python tools/bench_standin.py --config density_n|ppi_bp|hpo_metab|em_user [--steps 30] [--warmup 5]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from subgnn_amd.standins import PRESETS, write_standin          # noqa: E402  (the presets live with the package: tests use them too)


def time_steps(model, opt, hp, steps, warmup, graph):
    B = hp['batch_size']

    def index_batches():
        while True:
            for idx in model.train_dataloader().index_batches():
                if idx.numel() == B:
                    yield idx
    it = index_batches()
    if graph:
        from subgnn_amd.graph_step import CapturedTrainStep
        cap = CapturedTrainStep(model, opt, B, hp['grad_clip'])

        def step():
            return cap.replay(next(it))[0]
    else:
        def step():
            out = model.training_step(model.make_batch('train', next(it)), 0)
            opt.zero_grad(set_to_none=True)
            model.backward(None, out['loss'], opt, 0)
            torch.nn.utils.clip_grad_norm_(model.parameters(), hp['grad_clip'])
            opt.step()
            return out['loss']
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, float(loss.detach())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=sorted(PRESETS), required=True)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--atomics', action='store_true', help="hparams['deterministic'] = False: float atomics in the backward pass")
    args = ap.parse_args()
    from subgnn_amd import config, hotpath, precompute_graph_metrics as pgm
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    P = PRESETS[args.config]
    hp = dict(P['hp'])
    hp['deterministic'] = not args.atomics
    root = tempfile.mkdtemp(prefix=args.config + '_')
    name = args.config + '_standin'
    t0 = time.time()
    d, n_edges = write_standin(root, args.config)
    t_write = time.time() - t0
    t0 = time.time()
    pgm.calculate_stats(d, shortest_paths=not P['sparse'], ego=not P['sparse'])
    t_metrics = time.time() - t0
    config.PROJECT_ROOT = root
    torch.manual_seed(3)
    model = SubGNN(dict(hp), **dataset_paths(name))
    stages = None
    torch.cuda.synchronize()
    t0 = time.time()
    if P['sparse']:
        timer = hotpath.StageTimer(True)
        for sp in ('val', 'train'):       # val first: it also pays the one-time code-object loads
            hotpath.prepare_sparse(model, sp, timer if sp == 'train' else None)
        torch.cuda.synchronize()
        stages = {k: round(v, 3) for k, v in timer.summary().items()}
    else:
        model.prepare_data()
        torch.cuda.synchronize()
    t_prep = time.time() - t0
    opt = model.configure_optimizers()
    model.train()
    ms_eager, loss = time_steps(model, opt, model.hparams, args.steps, args.warmup, graph=False)
    ms_graph, loss_g = time_steps(model, opt, model.hparams, args.steps, args.warmup, graph=True)
    B = hp['batch_size']
    print(json.dumps({
        'metric': 'subgraphs/sec fwd+bwd (all 3 channels on)', 'unit': 'subgraphs/s', 'n_gpus': 1,
        'value': B * 1e3 / ms_graph, 'ms_per_step': ms_graph, 'hip_graph_step': True,
        'eager': {'value': B * 1e3 / ms_eager, 'ms_per_step': ms_eager},
        'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s stand-in (BA n=%d m=%d, %d edges, %d subgraphs), %s prepare, batch of %d, training '
                               'step = fwd + bwd + clip + Adam' % (args.config, P['n'], P.get('m', 5), n_edges, P['n_sub'],
                                                                   'sparse' if P['sparse'] else 'dense reference-shaped', B),
                   'cc_ids_shape': list(model.train_cc_ids.shape), 'n_layers': hp['n_layers'],
                   'structure_patches': int(model.structure_anchors.shape[0]) if model.structure_anchors is not None else 0},
        'deterministic_backward': not args.atomics, 'prepare_data_s': round(t_prep, 2), 'prepare_stages_ms_train_split': stages,
        'dataset_write_s': round(t_write, 2), 'graph_metrics_s': round(t_metrics, 2),
        'loss': loss, 'loss_graph': loss_g}))


if __name__ == '__main__':
    main()
