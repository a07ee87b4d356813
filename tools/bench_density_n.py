#!/usr/bin/env python3
"""BASELINE.json configs[0]: "DENSITY synthetic (prepare_dataset), ~1k-node base graph, neighborhood
channel only, CPU reference path" -- the one configuration the reference's own Python can run.

The same dataset (DENSITY recipe of subgnn_amd/prepare_dataset.py at the scale of
prepare_dataset/config_prepare_dataset.py:15-31: BA n=1000, m=5, 250 BFS subgraphs of 20 nodes, seed
42; embeddings N(0,1), D=32; APSP / degree dict / ego graphs written with scipy + networkx so that no
GPU is needed) and the same hyper-parameters (best_model_hyperparameters/density/
N_density_hyperparams.json, restated below; ``compute_similarities`` on) go through

  --mode reference   the IMPORTED reference on the CPU (build container only: needs /root/reference and
                     the stand-ins of tests/golden/_standins; RNG left as the reference has it) --
                     timed: SubGNN.prepare_data(), then training steps (fwd + loss.backward + Adam);
  --mode gpu         this repository on the MI355X -- same stages.

Each prints one JSON line; profiles/ keeps both (the reference line is measured on the build
container's 8 cores -- it cannot run on the GPU box).
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

H1 = {  # reference best_model_hyperparameters/density/N_density_hyperparams.json
    "max_epochs": 200, "use_neighborhood": True, "use_structure": False, "use_position": False, "seed": 0,
    "node_embed_size": 32, "structure_patch_type": "triangular_random_walk", "lstm_aggregator": "last",
    "n_processes": 4, "resample_anchor_patches": False, "freeze_node_embeds": False, "use_mpn_projection": True,
    "print_train_times": False, "compute_similarities": True, "batch_size": 64,
    "learning_rate": 0.00025922124890367574, "grad_clip": 0.4827462116072751, "n_layers": 5,
    "neigh_sample_border_size": 2, "n_anchor_patches_pos_out": 99, "n_anchor_patches_pos_in": 53,
    "n_anchor_patches_N_in": 20, "n_anchor_patches_N_out": 37, "n_anchor_patches_structure": 28,
    "linear_hidden_dim_1": 64, "linear_hidden_dim_2": 32, "n_triangular_walks": 6, "random_walk_len": 20,
    "sample_walk_len": 20, "rw_beta": 0.31289948259603506, "lstm_dropout": 0.00382614656521465,
    "lin_dropout": 0.09405144951216626, "lstm_n_layers": 2, "cc_aggregator": "sum", "trainable_cc": False,
    "max_sim_epochs": 5, "embedding_type": "gin",
}


def write_config1(root):
    """The dataset directory, written without a GPU (identical in both modes)."""
    import networkx as nx
    import numpy as np
    import scipy.sparse as sp
    from scipy.sparse.csgraph import shortest_path
    from subgnn_amd import prepare_dataset as pd
    out, info = pd.write_dataset(Path(root) / 'density_n', 'density', seed=42, embed_dim=32, n=1000, n_subgraphs=250)
    G = nx.read_edgelist(str(out / 'edge_list.txt'), nodetype=int)
    n = G.number_of_nodes()
    A = nx.to_scipy_sparse_array(G, nodelist=range(n), format='csr')
    d = shortest_path(sp.csr_matrix(A), method='D', unweighted=True)
    d[~np.isfinite(d)] = 0
    np.save(out / 'shortest_path_matrix.npy', d.astype(np.float64))
    with open(out / 'degree_sequence.txt', 'w') as f:
        json.dump({str(v): G.degree(v) for v in range(n)}, f)
    with open(out / 'ego_graphs.txt', 'w') as f:
        json.dump({str(v): [int(w) for w in G.neighbors(v)] for v in range(n)}, f)
    return out, info


def paths(name):
    return dict(graph_path=name + '/edge_list.txt', subgraph_path=name + '/subgraphs.pth',
                embedding_path=name + '/gin_embeddings.pth', similarities_path=name + '/similarities/',
                shortest_paths_path=name + '/shortest_path_matrix.npy',
                degree_dict_path=name + '/degree_sequence.txt', ego_graph_path=name + '/ego_graphs.txt')


def run_reference(root, steps):
    """Imports the reference with the stand-ins; no RNG shims (timing, not parity)."""
    here = REPO / 'tests' / 'golden'
    ref = Path('/root/reference')
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    sys.dont_write_bytecode = True
    sys.path[:0] = [str(here / '_standins'), str(ref / 'SubGNN'), str(ref)]
    import networkx as nx
    import scipy.sparse
    import torch
    import config as refconfig
    import subgraph_utils as su
    import SubGNN as S

    class _NxShim:                                     # networkx 2.4 adjacency_matrix semantics (su:136-143)
        def adjacency_matrix(self, G, *a, **k):
            return scipy.sparse.csr_matrix(nx.adjacency_matrix(G, *a, **k))

        def __getattr__(self, k):
            return getattr(nx, k)
    su.nx = _NxShim()
    refconfig.PROJECT_ROOT = Path(root)
    torch.manual_seed(0)
    model = S.SubGNN(dict(H1), **paths('density_n'))
    t0 = time.perf_counter()
    model.prepare_data()
    t_prep = time.perf_counter() - t0
    for key in list(vars(model)):                       # torch >= 2: stored N-anchor tensors must be contiguous
        pass
    for d in (model.anchors_neigh_int, model.anchors_neigh_border):
        for sp in d:
            for l in d[sp]:
                d[sp][l] = d[sp][l].contiguous()
    opt = model.configure_optimizers()
    model.train()
    loader = model.train_dataloader()
    batches = list(loader)
    B = batches[0]['cc_ids'].shape[0]

    def step(i):
        out = model.training_step(batches[i % len(batches)], i)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
    step(0)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i + 1)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return dict(kind='reference (imported, CPU)', cores=os.cpu_count(), torch_threads=torch.get_num_threads(),
                prepare_data_s=round(t_prep, 2), ms_per_step=ms, subgraphs_per_s=B * 1e3 / ms, batch=B, steps=steps)


def run_gpu(root, steps):
    import torch
    from subgnn_amd import config
    from subgnn_amd.SubGNN import SubGNN
    from subgnn_amd.graph_step import CapturedTrainStep
    config.PROJECT_ROOT = Path(root)
    torch.manual_seed(0)
    model = SubGNN(dict(H1), **paths('density_n'))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.prepare_data()
    torch.cuda.synchronize()
    t_prep = time.perf_counter() - t0
    opt = model.configure_optimizers()
    model.train()
    B = H1['batch_size']

    def idx_batches():
        while True:
            for idx in model.train_dataloader().index_batches():
                if idx.numel() == B:
                    yield idx
    it = idx_batches()
    res = {}
    for label, graph in (('eager', False), ('hip_graph', True)):
        if graph:
            cap = CapturedTrainStep(model, opt, B, H1['grad_clip'])
            step = lambda: cap.replay(next(it))
        else:
            def step():
                out = model.training_step(model.make_batch('train', next(it)), 0)
                opt.zero_grad(set_to_none=True)
                model.backward(None, out['loss'], opt, 0)
                torch.nn.utils.clip_grad_norm_(model.parameters(), H1['grad_clip'])
                opt.step()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        res[label] = dict(ms_per_step=ms, subgraphs_per_s=B * 1e3 / ms)
    return dict(kind='this repository (MI355X)', prepare_data_s=round(t_prep, 2), batch=B, steps=steps, **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=['reference', 'gpu'], required=True)
    ap.add_argument('--steps', type=int, default=20)
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix='config1_')
    t0 = time.perf_counter()
    out, info = write_config1(root)
    t_write = time.perf_counter() - t0
    res = run_reference(root, args.steps) if args.mode == 'reference' else run_gpu(root, args.steps)
    res.update(workload='BASELINE configs[0]: DENSITY recipe, BA n=1000 m=5 (%d nodes, %d edges after editing), %d BFS '
                        'subgraphs of 20 nodes, neighbourhood channel only, N_density_hyperparams (5 layers, N 20/37, '
                        'D=32, batch 64)' % (info['n_nodes'], info['n_edges'], info['n_subgraphs']),
               dataset_write_s=round(t_write, 2))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
