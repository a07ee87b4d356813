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

from subgnn_amd.standins import PRESETS                          # noqa: E402  (presets and the timing live with the package: bench.py's ``configs`` object uses them too)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=sorted(PRESETS), required=True)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--atomics', action='store_true', help="hparams['deterministic'] = False: float atomics in the backward pass")
    ap.add_argument('--no-count', action='store_true', help='skip the kernels-per-step count (torch profiler)')
    ap.add_argument('--epochs', type=int, default=0, help='also time this many whole epochs (train_config.Trainer)')
    args = ap.parse_args()
    from subgnn_amd.standins import bench_config
    print(json.dumps(bench_config(args.config, args.steps, args.warmup, deterministic=not args.atomics, count=not args.no_count,
                                  also_atomics=not args.atomics and not args.epochs, epochs=args.epochs)))


if __name__ == '__main__':
    main()
