"""Stand-in datasets for the BASELINE.json configurations whose real data is not available offline
(PPI-BP, HPO-METAB, EM-USER: a Dropbox link, reference README.md:24) plus the reference's own
DENSITY recipe at its published scale.  Graph statistics as quoted in SURVEY.md section 8; every
result produced from them is labelled "stand-in".  Everything is written in the reference's on-disk
formats (edge_list.txt, subgraphs.pth, *_embeddings.pth; graph metrics by
precompute_graph_metrics.calculate_stats) so that the drop-in ``SubGNN(hparams, **dataset_paths())``
constructor reads them like real data.

  density_n  configs[0]  DENSITY recipe (prepare_dataset.py) BA n=1000 m=5, 250 BFS subgraphs x 20
                         nodes; best_model_hyperparameters/density/N_density_hyperparams.json
                         (neighbourhood channel only, 5 layers)
  ppi_bp     configs[1]  BA n=17 080 m=19 (~322 k edges), 1 591 subgraphs of ~10 nodes in ~7 pieces;
                         train.py:109-148 hyper-parameters (all three channels, B=64, D=128)
  hpo_metab  configs[2]  BA n=14 587 m=222 (~3.0 M edges), 2 400 subgraphs of ~14 nodes in 1-2
                         pieces; best_model_hyperparameters/hpo_metab/hyperparams.json with all
                         three channels on (4 layers, 360 structure patches: DTW stressed)
  em_user    configs[4]  BA n=57 333 m=80 (~4.5 M edges), 324 subgraphs of ~155 nodes in ~52 pieces;
                         best_model_hyperparameters/em_user/hyperparams.json (k=2 border, B=32,
                         trainable_cc) with all three channels on, fp16-stored table, ff_attn read-out; the dense
                         (N, N) float64 hop matrix would be 26 GB -> hotpath.prepare_sparse
"""
import os

import numpy as np
import torch

_COMMON = {"max_epochs": 200, "structure_patch_type": "triangular_random_walk", "lstm_aggregator": "last",
           "n_processes": 4, "resample_anchor_patches": False, "freeze_node_embeds": False, "use_mpn_projection": True,
           "print_train_times": False, "compute_similarities": True, "set2set": False, "ff_attn": False,
           "linear_hidden_dim_1": 64, "linear_hidden_dim_2": 32, "max_sim_epochs": 5, "embedding_type": "gin",
           "use_neighborhood": True, "use_structure": True, "use_position": True, "node_embed_size": 128}

H1 = {  # reference best_model_hyperparameters/density/N_density_hyperparams.json (+ the two config-file keys)
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

H2 = {  # reference SubGNN/train.py:109-148 get_hyperparams (+ the two keys the config files add)
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


def _ppi_pieces(rng):
    return [int(s) for s in rng.permutation([1, 1, 1, 1, 1, 2, 3])[:int(rng.integers(5, 8))]]


PRESETS = {
    'density_n': dict(recipe='density', n=1000, n_sub=250, D=32, sparse=False, hp=H1),
    'ppi_bp': dict(n=17080, m=19, n_sub=1591, n_classes=6, D=128, sparse=False, pieces=_ppi_pieces, hp=H2),
    'hpo_metab': dict(
        n=14587, m=222, n_sub=2400, n_classes=6, D=128, sparse=False,
        pieces=lambda rng: [10, 4] if rng.random() < 0.6 else [14],
        hp=dict(_COMMON, **{
            "seed": 0, "sample_walk_len": 50, "n_triangular_walks": 5, "random_walk_len": 10, "rw_beta": 0.65,
            "batch_size": 64, "learning_rate": 0.0003658242069498871, "grad_clip": 0.26758489792349655, "n_layers": 4,
            "neigh_sample_border_size": 2, "n_anchor_patches_pos_out": 90, "n_anchor_patches_pos_in": 56,
            "n_anchor_patches_N_in": 13, "n_anchor_patches_N_out": 34, "n_anchor_patches_structure": 18,
            "lstm_dropout": 0.09909551715384933, "lstm_n_layers": 2, "lin_dropout": 0.21096188558408646,
            "cc_aggregator": "sum", "trainable_cc": False})),
    'em_user': dict(
        n=57333, m=80, n_sub=324, n_classes=2, D=128, sparse=True,
        pieces=lambda rng: [int(rng.integers(40, 60)), int(rng.integers(30, 50))] + [1] * int(rng.integers(40, 56))
        + [int(rng.integers(2, 5)) for _ in range(4)],
        hp=dict(_COMMON, **{
            "seed": 160761, "batch_size": 32, "learning_rate": 0.0007225432908901084, "grad_clip": 0.13742538368745078,
            "n_layers": 1, "neigh_sample_border_size": 2, "n_anchor_patches_pos_out": 77, "n_anchor_patches_pos_in": 48,
            "n_anchor_patches_N_in": 16, "n_anchor_patches_N_out": 32, "n_anchor_patches_structure": 35,
            "n_triangular_walks": 10, "random_walk_len": 23, "sample_walk_len": 22, "rw_beta": 0.1816027331132596,
            "lstm_dropout": 0.01599628663889252, "lin_dropout": 0.003486968525571843, "lstm_n_layers": 1,
            "cc_aggregator": "sum", "trainable_cc": True, "structure_similarity_fn": "dtw",
            "embedding_dtype": "fp16", "ff_attn": True})),      # configs[4]: "fp16 embeddings with MFMA attention scores"
            # (the attention read-out over a batch's 32 x 20 component rows: 640 rows take the library GEMM on half-rounded
            # operands + the fused epilogue; the hand-written v_mfma_f32_32x32x16_f16 kernel serves calls of >= 2048 rows --
            # whole-split evaluation, tests/test_gpu_configs.py::test_em_user_with_ff_attn_half_mfma_scores)
}


def write_standin(root, name, seed=7):
    """Write the stand-in dataset ``name`` under ``root``/<name>_standin -> (directory, #edges)."""
    from . import synthetic
    P = PRESETS[name]
    d = os.path.join(str(root), name + '_standin')
    if P.get('recipe'):
        from . import prepare_dataset as pd
        out, info = pd.write_dataset(d, P['recipe'], seed=42, embed_dim=P['D'], n=P['n'], n_subgraphs=P['n_sub'])
        with open(os.path.join(str(out), 'edge_list.txt')) as f:
            n_edges = sum(1 for line in f if line.strip())
        return str(out), n_edges
    os.makedirs(os.path.join(d, 'similarities'), exist_ok=True)
    n, m, n_sub = P['n'], P['m'], P['n_sub']
    edges = synthetic.barabasi_albert_edges(n, m, seed)
    rowptr, col = synthetic.sorted_csr(edges, n)
    und = np.unique(np.sort(edges, axis=1), axis=0)
    np.savetxt(os.path.join(d, 'edge_list.txt'), und, fmt='%d')
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(n_sub):
        nodes = []
        for size in P['pieces'](rng):
            nodes.extend(synthetic.bfs_subgraphs(rowptr, col, 1, int(size), int(rng.integers(1 << 30)))[0])
        nodes = list(dict.fromkeys(nodes))
        sp = 'train' if i < int(0.8 * n_sub) else ('val' if i < int(0.9 * n_sub) else 'test')
        lines.append('-'.join(str(v - 1) for v in nodes) + '\t' + str(i % P['n_classes']) + '\t' + sp + '\t\n')
    with open(os.path.join(d, 'subgraphs.pth'), 'w') as f:
        f.write(''.join(lines))
    torch.save(torch.randn(n, P['D'], generator=torch.Generator().manual_seed(seed)),
               os.path.join(d, 'gin_embeddings.pth'))
    return d, len(und)


def build_model(root, name, hp_over=None, device=None, prepare=True):
    """Write the stand-in, its graph metrics (GPU precompute) and return the prepared drop-in model:
    dense reference-shaped ``prepare_data`` or, for the presets whose (N, N) matrices cannot exist,
    ``hotpath.prepare_sparse`` on the train and val splits."""
    from . import config, hotpath, precompute_graph_metrics as pgm
    from .SubGNN import SubGNN, dataset_paths
    P = PRESETS[name]
    d, n_edges = write_standin(root, name)
    pgm.calculate_stats(d, device, shortest_paths=not P['sparse'], ego=not P['sparse'])
    config.PROJECT_ROOT = str(root)
    hp = dict(P['hp'])
    if hp_over:
        hp.update(hp_over)
    torch.manual_seed(3)
    model = SubGNN(hp, **dataset_paths(os.path.basename(d)))
    if prepare:
        if P['sparse']:
            for sp in ('val', 'train'):
                hotpath.prepare_sparse(model, sp)
        else:
            model.prepare_data()
    return model, d, n_edges


def count_kernels(fn):
    """Device kernels ``fn()`` launches (memory copies and fills by the runtime included), counted by torch's profiler
    (roctracer / rocprofiler-sdk underneath); None where the profiler cannot run (e.g. under rocprofv3, which owns the
    tracing interface).  Measurement aid of the launch-bound configurations: not on any product path."""
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        n = 0
        for ev in prof.events():
            dt = str(getattr(ev, 'device_type', ''))
            if 'CUDA' in dt or 'PrivateUse' in dt:
                n += 1
        return n if n > 0 else None
    except Exception:
        return None


def time_steps(model, opt, hp, steps, warmup, graph, count=False):
    """ms per batch-sized training step (fwd + bwd + clip + Adam: Trainer.fit's body) -> (ms, last loss, kernels per step).
    ``graph``: replayed from a hipGraph (graph_step.CapturedTrainStep) instead of queued eagerly."""
    import time
    B = hp['batch_size']

    def index_batches():
        while True:
            for idx in model.train_dataloader().index_batches():
                if idx.numel() == B:
                    yield idx
    it = index_batches()
    from .optim import ClipAdam
    own_clip = isinstance(opt, ClipAdam)                     # (clips inside its step: train_config.Trainer does the same)
    if graph:
        from .graph_step import CapturedTrainStep
        cap = CapturedTrainStep(model, opt, B, 0.0 if own_clip else hp['grad_clip'])

        def step():
            return cap.replay(next(it))[0]
    else:
        def step():
            out = model.training_step(model.make_batch('train', next(it)), 0)
            opt.zero_grad(set_to_none=True)
            model.backward(None, out['loss'], opt, 0)
            if not own_clip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), hp['grad_clip'])
            opt.step()
            return out['loss']
    for _ in range(warmup):
        step()
    # ``steps`` steps, timed in up to five equal blocks, the MEDIAN block reported: the boxes of this pool hold the device for
    # 30-60 ms two or three times in four seconds (tools/device_stall_probe.py: a stream of trivial kernels shows it), which
    # is the whole of a 20-step measurement of a 0.4 ms step -- one line of the driver's run had DENSITY at 0.91 ms instead of 0.42
    blocks = 5 if steps >= 10 else 1
    per = max(1, steps // blocks)
    times = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            loss = step()
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0) / per)
    ms = sorted(times)[(len(times) - 1) // 2]
    n_k = count_kernels(step) if count else None
    return ms, float(loss.detach()), n_k


def time_epochs(model, hp, epochs=4):
    """Whole EPOCHS of the reference's loop (train_config.py:156-186 / SubGNN.py:350-464) on a prepared model, as
    train_config.Trainer runs them: replayed training steps, the validation steps, validation_epoch_end (metrics,
    init_all_embeddings, anchor resample when hparams ask for it) and any re-recording -> dict: wall ms per epoch (mean over
    the epochs after the first, which also pays the eager warm-up steps and the recordings) and where it goes."""
    import time
    from .train_config import Trainer
    import gc
    tr = Trainer(epochs, hp.get('grad_clip', 0.0), log=lambda *a, **k: None, hip_graph_step=bool(hp.get('hip_graph_step', True)))
    tr.phase_times = []
    # (Trainer.fit freezes the heap after prepare_data: a full collection over the few million container objects the loaded
    # dataset left behind costs tens of ms when it happens.  The 50-80 ms stalls of single epochs -- 22 / 78 / 22 / 78 ms on the
    # PPI-BP stand-in -- are NOT the collector's, though: tools/epoch_stall_probe.py logs every collector pass and finds none;
    # the device itself stalls, also under a stream of trivial kernels: tools/device_stall_probe.py)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.fit(model, prepared=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    later = tr.phase_times[2:] or tr.phase_times[1:] or tr.phase_times      # (epoch 0: warm-up steps + the training recording; epoch 1 may
    keys = ('train_steps_s', 'validation_steps_s', 'validation_epoch_end_s')  #  still record the validation forward of a one-batch split)
    # the MEDIAN epoch (every epoch is listed beside it): in a long-lived process single epochs stall by tens of ms for reasons
    # outside the loop (allocator growth after an empty_cache, a collector pass) -- the mean of four epochs was 17 / 17 / 90 / 95 ms
    order = sorted(later, key=lambda r: sum(r.get(k, 0.0) for k in keys))
    med = order[(len(order) - 1) // 2]
    mean = {k: 1e3 * med.get(k, 0.0) for k in keys}
    first = {k[:-2]: round(1e3 * tr.phase_times[0].get(k, 0.0), 2) for k in keys}
    return {'epochs': epochs, 'wall_s': round(wall, 3), 'epoch_ms': round(sum(mean.values()), 3),
            'epoch_ms_is': 'the median epoch of epochs 2.. (all listed in every_epoch_ms)',
            'breakdown_ms': {k[:-2] + '_ms': round(v, 3) for k, v in mean.items()},
            # (single epochs stall by 50-80 ms on some boxes: DEVICE time -- one recorded step of 1 ms measured at 49 ms between
            # its HIP events, no collector pass, no host work: tools/epoch_stall_probe.py -- so the fastest epoch stands beside
            # the median, which is a stalled one when more than half of nine epochs are hit)
            'epoch_ms_fastest': round(1e3 * sum(order[0].get(k, 0.0) for k in keys), 3),
            'first_epoch_ms(warm-up steps + recordings)': first,
            'per_epoch': {k: later[-1].get(k) for k in ('replayed_steps', 'eager_steps', 'validation_batches')},
            'recordings_after_the_first_epoch': sum(r.get('recordings', 0) for r in later),
            'resample_anchor_patches': bool(hp.get('resample_anchor_patches', False)),
            'every_epoch_ms': [{k[:-2]: round(1e3 * r.get(k, 0.0), 2) for k in keys} for r in tr.phase_times],
            'monitor_last': tr.history[-1] if tr.history else None}


def bench_config(name, steps=30, warmup=5, deterministic=True, root=None, count=True, also_atomics=False, epochs=0):
    """One BASELINE configuration's stand-in end to end -> dict: dataset write, graph metrics, prepare_data, then the
    batch-sized training step eager and replayed (ms per step, subgraphs/s, kernels per step).  What bench.py's
    ``configs`` object and tools/bench_standin.py report."""
    import tempfile
    import time
    from . import config, hotpath, precompute_graph_metrics as pgm
    from .SubGNN import SubGNN, dataset_paths
    P = PRESETS[name]
    hp = dict(P['hp'])
    hp['deterministic'] = bool(deterministic)
    own_root = root is None
    root = root or tempfile.mkdtemp(prefix=name + '_')
    t0 = time.time()
    d, n_edges = write_standin(root, name)
    t_write = time.time() - t0
    t0 = time.time()
    pgm.calculate_stats(d, shortest_paths=not P['sparse'], ego=not P['sparse'])
    t_metrics = time.time() - t0
    old_root = config.PROJECT_ROOT
    config.PROJECT_ROOT = root
    try:
        torch.manual_seed(3)
        model = SubGNN(dict(hp), **dataset_paths(name + '_standin'))
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
        from .optim import accelerate
        opt = accelerate(model.configure_optimizers(), hp['grad_clip'], capturable=True)     # what train_config.Trainer steps with
        model.train()
        ms_eager, loss, k_eager = time_steps(model, opt, model.hparams, steps, warmup, graph=False, count=count)
        ms_graph, loss_g, _ = time_steps(model, opt, model.hparams, steps, warmup, graph=True)
        epoch = None
        if epochs:
            # whole epochs on a FRESH model of the same dataset (the timed steps above moved this one's parameters and step counts)
            torch.manual_seed(3)
            m_e = SubGNN(dict(hp), **dataset_paths(name + '_standin'))
            if P['sparse']:
                for sp in ('val', 'train'):
                    hotpath.prepare_sparse(m_e, sp)
            else:
                m_e.prepare_data()
            epoch = time_epochs(m_e, m_e.hparams, epochs)
            n_full = epoch['per_epoch']['replayed_steps'] or 0
            epoch['replayed_step_ms'] = round(ms_graph, 3)
            epoch['epoch_over_steps'] = round(epoch['epoch_ms'] / max(n_full * ms_graph, 1e-9), 3) if n_full else None
            epoch['fastest_epoch_over_steps'] = round(epoch['epoch_ms_fastest'] / max(n_full * ms_graph, 1e-9), 3) if n_full else None
            del m_e
        ms_atomics = k_atomics = None
        if also_atomics and deterministic:
            # the same dataset with hparams['deterministic'] = False (float atomics in the backward pass: no sorts, no segmented
            # sums -- fewer kernels, sums in arbitrary order): replayed only
            hp2 = dict(hp)
            hp2['deterministic'] = False
            torch.manual_seed(3)
            m2 = SubGNN(hp2, **dataset_paths(name + '_standin'))
            if P['sparse']:
                for sp in ('val', 'train'):
                    hotpath.prepare_sparse(m2, sp)
            else:
                m2.prepare_data()
            o2 = accelerate(m2.configure_optimizers(), hp['grad_clip'], capturable=True)
            m2.train()
            ms_atomics, _, _ = time_steps(m2, o2, m2.hparams, steps, warmup, graph=True)
            if count:
                _, _, k_atomics = time_steps(m2, o2, m2.hparams, 1, 2, graph=False, count=True)
            del m2, o2
    finally:
        config.PROJECT_ROOT = old_root
        if own_root:                       # (the dense stand-ins write an (N, N) float64 hop matrix: 1.7-2.3 GB each)
            import shutil
            shutil.rmtree(root, ignore_errors=True)
    B = hp['batch_size']
    return {
        'metric': 'subgraphs/sec fwd+bwd (all 3 channels on)' if (hp['use_position'] and hp['use_structure']) else
                  'subgraphs/sec fwd+bwd (neighborhood channel only: configs[0] as BASELINE words it)',
        'unit': 'subgraphs/s', 'n_gpus': 1,
        'value': B * 1e3 / ms_graph, 'ms_per_step': ms_graph, 'hip_graph_step': True,
        'eager': {'value': B * 1e3 / ms_eager, 'ms_per_step': ms_eager},
        'kernels_per_step': k_eager, 'epoch': epoch,
        'atomics': None if ms_atomics is None else {'ms_per_step': ms_atomics, 'value': B * 1e3 / ms_atomics, 'kernels_per_step': k_atomics,
                                                    'what': "hparams['deterministic'] = False: float atomics in the backward pass (not bit-reproducible)"},
        'steps': steps, 'warmup': warmup, 'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic (stand-in)',
        'config': {'workload': '%s stand-in (BA n=%d m=%d, %d edges, %d subgraphs), %s prepare, batch of %d, training '
                               'step = fwd + bwd + clip + Adam' % (name, P['n'], P.get('m', 5), n_edges, P['n_sub'],
                                                                   'sparse' if P['sparse'] else 'dense reference-shaped', B),
                   'cc_ids_shape': list(model.train_cc_ids.shape), 'n_layers': hp['n_layers'],
                   'structure_patches': int(model.structure_anchors.shape[0]) if getattr(model, 'structure_anchors', None) is not None else 0},
        'deterministic_backward': bool(deterministic), 'prepare_data_s': round(t_prep, 2), 'prepare_stages_ms_train_split': stages,
        'dataset_write_s': round(t_write, 2), 'graph_metrics_s': round(t_metrics, 2),
        'loss': loss, 'loss_graph': loss_g}
