#!/usr/bin/env python3
"""Secondary benchmark: BASELINE.json configs[1] -- "PPI-BP, all three channels, 1 x MI355X, batch of
64 subgraphs" -- the reference's per-step regime (train.py:109-148 hyper-parameters, dense
reference-shaped prepare_data from files on disk, B = 64 training steps fwd + bwd + Adam).

The real PPI-BP data is not available offline; the stand-in follows the statistics quoted in
SURVEY.md section 8 (17 080 nodes / ~317 k edges / 1 591 subgraphs of ~10 nodes in ~7 components,
6 classes) and is labelled as a stand-in.  Everything goes through the file formats of the reference
(edge_list.txt, subgraphs.pth, *_embeddings.pth, shortest_path_matrix.npy, degree_sequence.txt,
ego_graphs.txt) and the drop-in SubGNN constructor.  This regime is launch-latency bound (tens of
small kernels per step), which is why bench.py's headline workload is the large-shard one.

    python tools/bench_ppi_bp.py [--steps 50] [--warmup 5]
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

H2 = {   # reference SubGNN/train.py:109-148 get_hyperparams (+ the two keys the config files add)
    "max_epochs": 200, "use_neighborhood": True, "use_structure": True, "use_position": True, "seed": 3,
    "node_embed_size": 128, "structure_patch_type": "triangular_random_walk", "lstm_aggregator": "last",
    "n_processes": 4, "resample_anchor_patches": False, "freeze_node_embeds": False, "use_mpn_projection": True,
    "print_train_times": False, "compute_similarities": True, "sample_walk_len": 50, "n_triangular_walks": 5,
    "random_walk_len": 10, "rw_beta": 0.65, "set2set": False, "ff_attn": False, "batch_size": 64,
    "learning_rate": 0.00025420762516423353, "grad_clip": 0.2160947806012501, "n_layers": 1,
    "neigh_sample_border_size": 1, "n_anchor_patches_pos_out": 123, "n_anchor_patches_pos_in": 34,
    "n_anchor_patches_N_in": 19, "n_anchor_patches_N_out": 69, "n_anchor_patches_structure": 37,
    "linear_hidden_dim_1": 64, "linear_hidden_dim_2": 32, "lstm_dropout": 0.21923625197416907, "lstm_n_layers": 2,
    "lin_dropout": 0.04617609616314509, "cc_aggregator": "max", "trainable_cc": True, "auto_lr_find": True,
    "max_sim_epochs": 5, "embedding_type": "gin",
}


def write_standin(root, n=17080, m=19, n_sub=1591, seed=7):
    from subgnn_amd import synthetic
    d = os.path.join(root, 'ppi_bp_standin')
    os.makedirs(os.path.join(d, 'similarities'), exist_ok=True)
    edges = synthetic.barabasi_albert_edges(n, m, seed)
    rowptr, col = synthetic.sorted_csr(edges, n)
    und = np.unique(np.sort(edges, axis=1), axis=0)
    with open(os.path.join(d, 'edge_list.txt'), 'w') as f:
        f.write(''.join('%d %d\n' % (u, v) for u, v in und))
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(n_sub):
        nodes = []
        for size in rng.permutation([1, 1, 1, 1, 1, 2, 3])[:int(rng.integers(5, 8))]:
            piece = synthetic.bfs_subgraphs(rowptr, col, 1, int(size), int(rng.integers(1 << 30)))[0]
            nodes.extend(piece)
        nodes = list(dict.fromkeys(nodes))
        sp = 'train' if i < int(0.8 * n_sub) else ('val' if i < int(0.9 * n_sub) else 'test')
        lines.append('-'.join(str(v - 1) for v in nodes) + '\t' + str(i % 6) + '\t' + sp + '\t\n')
    with open(os.path.join(d, 'subgraphs.pth'), 'w') as f:
        f.write(''.join(lines))
    torch.save(torch.randn(n, 128, generator=torch.Generator().manual_seed(seed)), os.path.join(d, 'gin_embeddings.pth'))
    return d, len(und)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--graph', action='store_true', help='replay the step from a hipGraph (graph_step.py)')
    args = ap.parse_args()
    from subgnn_amd import config, precompute_graph_metrics as pgm
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    root = tempfile.mkdtemp(prefix='ppi_bp_')
    t0 = time.time()
    d, n_edges = write_standin(root)
    pgm.calculate_stats(d)
    t_data = time.time() - t0
    config.PROJECT_ROOT = root
    torch.manual_seed(3)
    model = SubGNN(dict(H2), **dataset_paths('ppi_bp_standin'))
    t0 = time.time()
    model.prepare_data()
    torch.cuda.synchronize()
    t_prep = time.time() - t0
    opt = model.configure_optimizers()
    model.train()
    B = H2['batch_size']

    def batches():
        while True:
            for b in model.train_dataloader():
                yield b
    it = batches()

    def step():
        batch = next(it)
        out = model.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        model.backward(None, out['loss'], opt, 0)
        torch.nn.utils.clip_grad_norm_(model.parameters(), H2['grad_clip'])
        opt.step()
        return out['loss']
    if args.graph:
        from subgnn_amd.graph_step import CapturedTrainStep
        cap = CapturedTrainStep(model, opt, B, H2['grad_clip'])

        def index_batches():
            while True:
                for idx in model.train_dataloader().index_batches():
                    if idx.numel() == B:
                        yield idx
        iti = index_batches()

        def step():
            return cap.replay(next(iti))[0]
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({
        'metric': 'subgraphs/sec fwd+bwd (all 3 channels on)', 'value': B * args.steps / el, 'unit': 'subgraphs/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * el / args.steps,
        'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'PPI-BP stand-in (BA n=17080 m=19, %d edges, 1591 subgraphs of ~10 nodes in ~7 components), '
                               'train.py:109-148 hyper-parameters (D=128, N 19/69, P 34/123, S 37, 2-layer LSTM, '
                               'trainable_cc), batch of 64, training step = fwd + bwd + clip + Adam' % n_edges,
                   'cc_ids_shape': list(model.train_cc_ids.shape)},
        'prepare_data_s': round(t_prep, 2), 'dataset_write_and_graph_metrics_s': round(t_data, 2),
        'loss': float(loss.detach()), 'hip_graph_step': bool(args.graph)}))


if __name__ == '__main__':
    main()
