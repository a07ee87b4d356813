"""Build-owned counterpart of the reference's 10-seed experiment driver (SubGNN/test.py:27-102).

For each seed: train a fresh model (train_config.train_model with ``hyperparams_fix.seed`` set to
the seed, SubGNN/test.py:63-70 -> train.py), run the test split, collect ``test_micro_f1``,
``test_acc``, ``test_auroc`` (SubGNN/test.py:84-86); then write means / standard deviations and the
per-seed lists to ``experiment_results.json`` with the reference's keys (SubGNN/test.py:88-101).
Seeds are 0..n-1, or random in [0, 10^6] with ``-random_seeds`` (SubGNN/test.py:65).
"""
import argparse
import copy
import json
import random
from pathlib import Path

import numpy as np

from . import config
from .train_config import read_json, train_model


def run_seeds(run_config, n_seeds=10, random_seeds=False, results_dir=None, log=print):
    """Returns the experiment_results dict of SubGNN/test.py:52-57."""
    exp = {"test_acc_mean": 0, "test_acc_sd": 0, "test_micro_f1_mean": 0, "test_micro_f1_sd": 0,
           "test_auroc_mean": 0, "test_auroc_sd": 0, "test_acc": [], "test_micro_f1": [], "test_auroc": [],
           "call": {"task": run_config['data']['task'], "n_seeds": n_seeds, "random_seeds": bool(random_seeds)}}
    for rnd in range(n_seeds):
        seed = random.randint(0, 1000000) if random_seeds else rnd
        log('Running Round %d\nSeed used:  %d' % (rnd + 1, seed))
        cfg = copy.deepcopy(run_config)
        cfg['hyperparams_fix']['seed'] = seed
        out = Path(results_dir) / ('version_%d' % rnd) if results_dir is not None else None
        _, model, trainer = train_model(cfg, results_dir=out, log=lambda *a: None)
        trainer.test(model)
        res = model.test_results
        for k in ('test_micro_f1', 'test_acc', 'test_auroc'):
            exp[k].append(float(res[k]))
    for k in ('test_acc', 'test_micro_f1', 'test_auroc'):
        exp[k + '_mean'] = float(np.mean(exp[k]))
        exp[k + '_sd'] = float(np.std(exp[k]))
    log('OVERALL RESULTS:')
    log(exp)
    if results_dir is not None:
        Path(results_dir).mkdir(parents=True, exist_ok=True)
        with open(Path(results_dir) / 'experiment_results.json', 'w') as f:
            json.dump(exp, f, indent=4)
    return exp


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train and test SubGNN on MI355X for several seeds')
    ap.add_argument('-config_path', type=str, required=True, help='reference-format config.json')
    ap.add_argument('-project_root', type=str, default=None)
    ap.add_argument('-results_dir', type=str, default='tensorboard_test/sg')
    ap.add_argument('-n_seeds', type=int, default=10)
    ap.add_argument('-random_seeds', action='store_true')
    args = ap.parse_args(argv)
    if args.project_root:
        config.PROJECT_ROOT = Path(args.project_root)
    return run_seeds(read_json(args.config_path), args.n_seeds, args.random_seeds,
                     Path(config.PROJECT_ROOT) / args.results_dir)


if __name__ == '__main__':
    main()
