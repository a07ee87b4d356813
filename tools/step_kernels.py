#!/usr/bin/env python3
"""The kernels ONE eager batch-sized training step of a stand-in configuration launches, in launch order, with their device
time (torch profiler) -- what the launch diet of the B = 64 configurations is read from.

    python tools/step_kernels.py --config ppi_bp [--out profiles/r05_step_kernels_ppi_bp.txt]
"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    from torch.profiler import ProfilerActivity, profile
    from subgnn_amd import config, hotpath, precompute_graph_metrics as pgm, standins
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    P = standins.PRESETS[args.config]
    root = tempfile.mkdtemp(prefix=args.config + '_')
    d, _ = standins.write_standin(root, args.config)
    pgm.calculate_stats(d, shortest_paths=not P['sparse'], ego=not P['sparse'])
    config.PROJECT_ROOT = root
    torch.manual_seed(3)
    model = SubGNN(dict(P['hp']), **dataset_paths(args.config + '_standin'))
    if P['sparse']:
        for sp in ('val', 'train'):
            hotpath.prepare_sparse(model, sp)
    else:
        model.prepare_data()
    from subgnn_amd.optim import ClipAdam, accelerate
    opt = accelerate(model.configure_optimizers(), model.hparams['grad_clip'], capturable=True)      # what train_config.Trainer steps with
    model.train()
    hp = model.hparams
    it = iter([i for i in model.train_dataloader().index_batches() if i.numel() == hp['batch_size']] * 50)

    def step():
        out = model.training_step(model.make_batch('train', next(it)), 0)
        opt.zero_grad(set_to_none=True)
        model.backward(None, out['loss'], opt, 0)
        if not isinstance(opt, ClipAdam):
            torch.nn.utils.clip_grad_norm_(model.parameters(), hp['grad_clip'])
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if 'CUDA' in str(getattr(e, 'device_type', ''))]
    evs.sort(key=lambda e: e.time_range.start)
    lines = []
    tot = 0.0
    for i, e in enumerate(evs):
        dur = e.time_range.end - e.time_range.start
        tot += dur
        lines.append('%4d %8.1f  %s' % (i, dur, e.name[:150]))
    lines.append('kernels %d, device time %.1f us' % (len(evs), tot))
    txt = '\n'.join(lines)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
