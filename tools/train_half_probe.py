"""The training half (component embeddings .. clip + Adam) of the benchmark pass on its own: eager vs replayed from a hipGraph,
with NOTHING else on the device (no preparation stream beside it), at a shard size.
usage: python tools/train_half_probe.py [subgraphs=6250] [reps=30]  ->  one JSON line"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from subgnn_amd import ops, hotpath, optim                      # noqa: E402
from subgnn_amd.SubGNN import SubGNN                            # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    sys.argv = sys.argv[:1]
    args = bench.parse()
    rowptr, col, subs, _, _ = bench.build_inputs(args, 0, 1)
    subs = subs[:S]
    dev = torch.device('cuda', 0)
    g = ops.DeviceGraph(rowptr, col, np.arange(1, args.nodes + 1, dtype=np.int32), dev)
    torch.manual_seed(0)
    emb = torch.randn(args.nodes, args.embed, device=dev)
    hp = dict(bench.ALL_DENSITY_HP)
    if os.environ.get('SGNN_BENCH_HP'):
        hp.update(json.loads(os.environ['SGNN_BENCH_HP']))
    labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
    labels[:3] = torch.tensor([0, 1, 2])
    model = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]},
                               emb, num_classes=3)
    model.train()
    opt = optim.ClipAdam(model.parameters(), hp['learning_rate'], max_norm=hp['grad_clip'], capturable=True)
    trainer = hotpath.CapturedTraining(model, opt, 'train', warmup=2)
    st = hotpath.prepare_pass(model, 'train')
    trainer.install(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, 1e3 * (time.perf_counter() - t0) / reps
    eager_ev, eager_wall = timed(trainer._body)                  # (warm-up steps are eager bodies)
    trainer._warm_left = 0
    trainer.step()                                               # records
    graph_ev, graph_wall = timed(trainer.step)
    kernels = None
    try:
        from torch.profiler import profile, ProfilerActivity
        trainer._body()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            trainer._body()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        kernels = len(evs)
        ktime = sum(e.device_time_total for e in evs) / 1e3
        names = {}
        for e in evs:
            c, t = names.get(e.name[:70], (0, 0.0))
            names[e.name[:70]] = (c + 1, round(t + e.device_time_total, 1))
    except Exception as ex:                                      # noqa: BLE001
        ktime, names = None, {'error': repr(ex)}
    print(json.dumps({'subgraphs': S, 'eager_ms': round(eager_ev, 3), 'eager_wall_ms': round(eager_wall, 3),
                      'graph_replay_ms': round(graph_ev, 3), 'graph_wall_ms': round(graph_wall, 3), 'kernels': kernels,
                      'kernel_time_ms': ktime, 'by_name(count, us)': dict(sorted(names.items(), key=lambda kv: -kv[1][1] if isinstance(kv[1], tuple) else 0)[:80])}))


if __name__ == '__main__':
    main()
