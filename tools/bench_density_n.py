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
  --mode port        the oracle (oracle/integer_half.py + float_half.py: the CPU restatement bench.py's cpu_baseline times
                     at the 1M-node size) on the SAME dataset, same stages, same batch size, on the same cores -- the
                     ratio reference / port calibrates the port back to "reference CPU path" (BASELINE.md section 3 iii);
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


ALL = {  # reference best_model_hyperparameters/density/all_density_hyperparams.json (+ max_sim_epochs / embedding_type, which
    # the json lacks and SubGNN.py:893 / train_config.py:214 read) with D = 32 (the embeddings written for config 1)
    "use_neighborhood": True, "use_structure": True, "use_position": True, "seed": 0, "max_epochs": 200,
    "node_embed_size": 32, "structure_patch_type": "triangular_random_walk", "lstm_aggregator": "last",
    "n_processes": 4, "resample_anchor_patches": False, "freeze_node_embeds": False, "print_train_times": False,
    "use_mpn_projection": True, "compute_similarities": True, "sample_walk_len": 50,
    "n_triangular_walks": 5, "random_walk_len": 10, "rw_beta": 0.65, "batch_size": 64,
    "learning_rate": 0.0002951850045886519, "grad_clip": 0.1929946246623414, "n_layers": 1,
    "neigh_sample_border_size": 1, "n_anchor_patches_pos_out": 183, "n_anchor_patches_pos_in": 57,
    "n_anchor_patches_N_in": 10, "n_anchor_patches_N_out": 43, "n_anchor_patches_structure": 42,
    "linear_hidden_dim_1": 64, "linear_hidden_dim_2": 64, "lin_dropout": 0.2522849803237359,
    "lstm_dropout": 0.0, "lstm_n_layers": 1, "cc_aggregator": "max", "trainable_cc": False,
    "max_sim_epochs": 5, "embedding_type": "gin", "structure_similarity_fn": "dtw",
}
HP = {'n_density': H1, 'all_density': ALL}


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


def run_reference(root, steps, H1=H1):
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
    for d in (model.anchors_neigh_int, model.anchors_neigh_border) if H1['use_neighborhood'] else ():
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


def run_port(root, steps, H1=H1):
    """The oracle restatement on the dataset the reference leg reads: prepare_data's neighbourhood stages (components,
    border sets with the ego-dict semantics, N anchors for 5 layers, dense shortest-path similarities), then training
    steps of 64 subgraphs: oracle/float_half.forward + cross entropy + backward + Adam.  TEST INFRASTRUCTURE timed as a
    baseline; nothing of the product runs here."""
    import numpy as np
    import torch
    from oracle import graph as OG, integer_half as IH, float_half as FH, tape as T
    d = Path(root) / 'density_n'
    G = OG.read_edgelist(str(d / 'edge_list.txt'))
    subs, labels = {'train': [], 'val': [], 'test': []}, {'train': [], 'val': [], 'test': []}
    names = {}
    with open(d / 'subgraphs.pth') as f:
        for line in f:
            nodes, lab, split = line.rstrip('\n').split('\t')[:3]
            subs[split].append([int(v) + 1 for v in nodes.split('-')])
            labels[split].append(names.setdefault(lab, len(names)))
    emb = torch.load(d / 'gin_embeddings.pth')
    apsp = np.load(d / 'shortest_path_matrix.npy')
    hp = dict(H1)
    D, L, seed = emb.shape[1], hp['n_layers'], hp['seed']
    t0 = time.perf_counter()
    prep = {}
    for split in ('train', 'val', 'test'):
        if not subs[split]:
            continue
        cc_ids = IH.pad_cc_ids([IH.connected_components(G, s) for s in subs[split]])
        S, C, Lc = cc_ids.shape
        border = IH.pad_border_sets([[IH.component_border_set(G, cc_ids[s, c], hp['neigh_sample_border_size'], ego_dict_mode=True)
                                      if cc_ids[s, c, 0] != 0 else set() for c in range(C)] for s in range(S)])
        n_int = {l: IH.sample_neighborhood_anchors(cc_ids, hp['n_anchor_patches_N_in'], seed, T.stream_id(T.STREAM_N_INT, split, l))
                 for l in range(L)}
        n_bor = {l: IH.sample_neighborhood_anchors(border, hp['n_anchor_patches_N_out'], seed, T.stream_id(T.STREAM_N_BOR, split, l))
                 for l in range(L)}
        sims = IH.shortest_path_similarities(apsp, cc_ids)
        extra = None
        if hp['use_position'] or hp['use_structure']:
            from oracle import cbind
            if split == 'train':                    # shared draws: once (aps:306-328; SubGNN.py:1036-1050)
                rowptr, col = G.csr(sort=True)
                full = np.zeros(len(rowptr) - 1, dtype=np.int32)
                for v in G.node_order:
                    full[v] = G.degree(v)
                n_p = hp['max_sim_epochs'] * hp['n_anchor_patches_structure'] * L
                patches = IH.sample_structure_anchor_patches(G, n_p, hp['sample_walk_len'], hp['rw_beta'], seed)
                iw = IH.perform_random_walks(G, patches, hp['n_triangular_walks'], hp['random_walk_len'], hp['rw_beta'], True, seed)
                bw = IH.perform_random_walks(G, patches, hp['n_triangular_walks'], hp['random_walk_len'], hp['rw_beta'], False, seed)
                pp, pf = cbind.ragged([[int(v) for v in row if v] for row in patches])
                pi, pe = cbind.degree_sequence(rowptr, col, full, pp, pf, True)
                pext = {l: IH.position_anchors_border(G, hp['n_anchor_patches_pos_out'], seed, l) for l in range(L)}
                spick = {l: IH.structure_anchor_indices(n_p, hp['n_anchor_patches_structure'], seed, l) for l in range(L)}
                shared = (patches, iw, bw, pext, spick)
            p_int = {l: IH.position_anchors_internal(subs[split], hp['n_anchor_patches_pos_in'], seed, split, l) for l in range(L)}
            cp, cf = cbind.ragged([[int(v) for v in r if v] for r in cc_ids.reshape(S * C, Lc)])
            ci, ce = cbind.degree_sequence(rowptr, col, full, cp, cf, True)
            real = (cc_ids[:, :, 0] != 0)[..., None]
            extra = (p_int, cbind.fastdtw_sim(cp, ci, pp, pi).reshape(S, C, -1) * real,
                     cbind.fastdtw_sim(cp, ce, pp, pe).reshape(S, C, -1) * real)
        prep[split] = (cc_ids, n_int, n_bor, sims, extra)
    t_prep = time.perf_counter() - t0
    cc_ids, n_int, n_bor, sims, extra = prep['train']
    g = torch.Generator().manual_seed(0)
    params = {'node_embeddings.weight': torch.cat([torch.zeros(1, D), emb.float()], 0).requires_grad_(True)}

    def lin(name, o, i):
        params[name + '.weight'] = (torch.randn(o, i, generator=g) / i ** 0.5).requires_grad_(True)
        params[name + '.bias'] = torch.zeros(o, requires_grad=True)
    for l in range(L):
        for side in ('internal', 'border'):
            lin('neighborhood_mpns.%d.%s.linear' % (l, side), D, 2 * D)
            lin('neighborhood_mpns.%d.%s.linear_position' % (l, side), 1, D)
    hid = D + 2 * D * L
    if hp['use_position']:
        for l in range(L):
            for side in ('internal', 'border'):
                lin('position_mpns.%d.%s.linear' % (l, side), D, 2 * D)
                lin('position_mpns.%d.%s.linear_position' % (l, side), 1, D)
        hid += L * (hp['n_anchor_patches_pos_in'] + hp['n_anchor_patches_pos_out'])
    if hp['use_structure']:
        for l in range(L):
            for side in ('internal', 'border'):
                lin('structure_mpns.%d.%s.linear' % (l, side), D, 2 * D)
                lin('structure_mpns.%d.%s.linear_position' % (l, side), 1, D)
        for k in range(hp['lstm_n_layers']):
            for sfx in ('', '_reverse'):
                i_ = D if k == 0 else 2 * D
                params['lstm.lstm.weight_ih_l%d%s' % (k, sfx)] = (torch.randn(4 * D, i_, generator=g) / i_ ** 0.5).requires_grad_(True)
                params['lstm.lstm.weight_hh_l%d%s' % (k, sfx)] = (torch.randn(4 * D, D, generator=g) / D ** 0.5).requires_grad_(True)
                params['lstm.lstm.bias_ih_l%d%s' % (k, sfx)] = torch.zeros(4 * D, requires_grad=True)
                params['lstm.lstm.bias_hh_l%d%s' % (k, sfx)] = torch.zeros(4 * D, requires_grad=True)
        lin('lstm.linear', D, 2 * D)
        hid += L * 2 * hp['n_anchor_patches_structure']
    lin('lin', hp['linear_hidden_dim_1'], hid)
    lin('lin2', hp['linear_hidden_dim_2'], hp['linear_hidden_dim_1'])
    lin('lin3', max(len(names), 2), hp['linear_hidden_dim_2'])
    opt = torch.optim.Adam(list(params.values()), lr=hp['learning_rate'])
    Tt = torch.from_numpy
    anchors = {'N_int': {'train': {l: Tt(n_int[l]) for l in range(L)}}, 'N_bor': {'train': {l: Tt(n_bor[l]) for l in range(L)}},
               'P_int': {}, 'P_ext': {}, 'S': {}}
    if extra is not None:
        patches, iw, bw, pext, spick = shared
        anchors['P_int'] = {'train': {l: Tt(extra[0][l]) for l in range(L)}}
        anchors['P_ext'] = {l: Tt(pext[l]) for l in range(L)}
        anchors['S'] = {l: (Tt(patches[spick[l]]), spick[l], Tt(iw[spick[l]]), Tt(bw[spick[l]])) for l in range(L)}
    lab = torch.tensor(labels['train'])
    B, S = hp['batch_size'], cc_ids.shape[0]
    order = np.random.default_rng(0).permutation(S)

    def step(i):
        idx = np.sort(order[(i * B) % (S - B + 1):(i * B) % (S - B + 1) + B])
        batch = {'cc_ids': Tt(IH.trim_zero_columns(cc_ids[idx])), 'subgraph_idx': Tt(idx).view(-1, 1), 'NP_sim': Tt(sims[idx]),
                 'I_S_sim': Tt(extra[1][idx]).float() if extra is not None else None,
                 'B_S_sim': Tt(extra[2][idx]).float() if extra is not None else None}
        loss = torch.nn.functional.cross_entropy(FH.forward(params, hp, 'train', batch, anchors, None), lab[idx])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), hp['grad_clip'])
        opt.step()
    step(0)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i + 1)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return dict(kind='port (oracle restatement, CPU)', cores=os.cpu_count(), torch_threads=torch.get_num_threads(),
                prepare_data_s=round(t_prep, 2), ms_per_step=ms, subgraphs_per_s=B * 1e3 / ms, batch=B, steps=steps)


def run_gpu(root, steps, H1=H1):
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
    ap.add_argument('--mode', choices=['reference', 'port', 'gpu'], required=True)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--hparams', choices=sorted(HP), default='n_density',
                    help='n_density: BASELINE configs[0] as worded (neighbourhood channel only); all_density: the same dataset with '
                         'all three channels on (the hyper-parameters of bench.py\'s workload, D = 32)')
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix='config1_')
    t0 = time.perf_counter()
    out, info = write_config1(root)
    t_write = time.perf_counter() - t0
    res = {'reference': run_reference, 'port': run_port, 'gpu': run_gpu}[args.mode](root, args.steps, HP[args.hparams])
    res['hparams'] = args.hparams
    what = ('neighbourhood channel only, N_density_hyperparams (5 layers, N 20/37, D=32, batch 64)' if args.hparams == 'n_density' else
            'ALL THREE channels, all_density_hyperparams (1 layer, N 10/43, P 57/183, S 42, 210 structure patches, D=32, batch 64)')
    res.update(workload='BASELINE configs[0] dataset: DENSITY recipe, BA n=1000 m=5 (%d nodes, %d edges after editing), %d BFS '
                        'subgraphs of 20 nodes; %s' % (info['n_nodes'], info['n_edges'], info['n_subgraphs'], what),
               dataset_write_s=round(t_write, 2))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
