"""Does a training run of the benchmark repeat bit for bit ACROSS processes?  (Within one process it does:
tools/train_determinism_probe.py.)

    python tools/cross_process_probe.py --procs 10 [--passes 4] [--variant name ...]       # the driver
    python tools/cross_process_probe.py --child out.json [--passes 4] [--variant name ...] # one fresh process

The driver never touches the GPU: it starts ``--procs`` fresh child processes one after the other; every child runs
``--passes`` sequential passes of the benchmark workload (prepare -> forward -> backward -> clip + Adam, dropout 0) and
writes a checksum of EVERY tensor on the way: the prepared state, logits, loss, each gradient and each parameter after
the step, per pass.  The driver groups the children by outcome and, for every child that deviates from the first one,
names the first checksum (in execution order) that differs -- the operation to look at.

Variants (bisecting the cause; combinable):
  one_stream        hparams['overlap_streams'] = False (preparation on the caller's stream)
  poison            torch.empty / empty_like / new_empty return NaN- (floats) or 0x7f7f..-filled (integers) memory: a
                    kernel that reads a buffer it has not written shows up as NaN / a changed result in EVERY process
  hipblas/hipblaslt torch.backends.cuda.preferred_blas_library(...)
  no_tall           ops.linear always takes the library path (no split-contraction backward)
  no_shared_gemm    SHARED layers by the hand-written kernel at every size (ops.SHARED_GEMM_MIN_ROWS = infinity)
  atomics_off       ROCBLAS_DEFAULT_ATOMICS_MODE=0 in the child's environment
  dropout           lin_dropout as in the benchmark's hyper-parameters (default here: 0)
  nosync            no host synchronisation inside the run: the per-pass losses are kept on the device and checksummed at the
                    end (use with --detail-passes 0), so the host runs ahead as it does in bench.py
  pipelined         the passes go through hotpath.PassPipeline (two in flight) as bench.py's default schedule does
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(args):
    variants = set(args.variant)
    import numpy as np
    import torch
    if 'poison' in variants:
        real_empty, real_like = torch.empty, torch.empty_like

        def fill(t):
            if t.is_cuda and t.numel():
                if t.dtype.is_floating_point:
                    t.fill_(float('nan'))
                elif t.dtype == torch.bool:
                    t.fill_(True)
                else:
                    t.fill_(0x7f7f7f7f if t.dtype in (torch.int32, torch.int64) else 0x7f)
            return t

        def empty(*a, **k):
            return fill(real_empty(*a, **k))

        def empty_like(*a, **k):
            return fill(real_like(*a, **k))
        torch.empty, torch.empty_like = empty, empty_like
        real_new_empty = torch.Tensor.new_empty
        torch.Tensor.new_empty = lambda self, *a, **k: fill(real_new_empty(self, *a, **k))
    for lib in ('hipblas', 'hipblaslt'):
        if lib in variants:
            torch.backends.cuda.preferred_blas_library(lib)
    from subgnn_amd import hotpath, ops, optim, synthetic
    from subgnn_amd.SubGNN import SubGNN
    import bench
    if 'no_tall' in variants:
        ops.linear = lambda x, w, b: torch.nn.functional.linear(x, w, b)
    if 'no_shared_gemm' in variants:
        ops.SHARED_GEMM_MIN_ROWS = 1 << 60
    n, m, S = args.nodes, 10, args.subgraphs
    dev = torch.device('cuda:0')
    cache = '/tmp/sgnn_probe_graph_%d_%d.npz' % (n, S)
    if os.path.exists(cache):
        z = np.load(cache, allow_pickle=True)
        rowptr, col, subs = z['rowptr'], z['col'], [list(map(int, r)) for r in z['subs']]
    else:
        edges = synthetic.barabasi_albert_edges(n, m, seed=42)
        rowptr, col = synthetic.sorted_csr(edges, n)
        subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
        np.savez(cache + '.tmp.npz', rowptr=rowptr, col=col, subs=np.asarray(subs, dtype=np.int64))
        os.replace(cache + '.tmp.npz', cache)
    g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
    hp = dict(bench.ALL_DENSITY_HP, lin_dropout=0.0)
    if 'dropout' in variants:
        hp['lin_dropout'] = bench.ALL_DENSITY_HP['lin_dropout']
        torch.cuda.manual_seed(0)
    if 'one_stream' in variants:
        hp['overlap_streams'] = False
    emb = torch.randn(n, hp['node_embed_size'], generator=torch.Generator().manual_seed(0))
    labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
    torch.manual_seed(0)
    model = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []},
                               {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3)
    model.train()
    opt = optim.ClipAdam(model.parameters(), hp['learning_rate'], max_norm=hp['grad_clip'])
    sums = []                                     # (name, checksum) in execution order

    def ck(name, t):
        if t is None:
            return
        if hasattr(t, 'dense') and not isinstance(t, torch.Tensor):
            return                                # ZeroSims
        t = t.detach().reshape(-1).contiguous()
        if t.numel() == 0:
            sums.append((name, [0, 0]))
            return
        b = t.view(torch.uint8).reshape(-1)
        pad = (-b.numel()) % 4
        if pad:
            b = torch.cat([b, b.new_zeros(pad)])
        v = b.view(torch.int32).long()
        w = (torch.arange(v.numel(), device=v.device) % 65521) + 1
        sums.append((name, [int(v.sum()), int((v * w).sum())]))

    def walk(prefix, o):
        if isinstance(o, torch.Tensor):
            ck(prefix, o)
        elif isinstance(o, dict):
            for k, v in o.items():
                walk('%s[%r]' % (prefix, k), v)
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                walk('%s[%d]' % (prefix, i), v)

    kept = []
    pipe = None
    if 'pipelined' in variants:
        pipe = hotpath.PassPipeline(model, 'train')
        pipe.start()
        pipe.start()
    for p_ in range(args.passes):
        if pipe is not None:
            pipe.install()
            pipe.start()
        else:
            hotpath.prepare_sparse(model, 'train')
        tag = 'pass%d/' % p_
        if p_ < args.detail_passes:
            for nm in ('train_cc_ids', 'train_neigh_pos_similarities', 'train_int_struc_similarities',
                       'train_bor_struc_similarities', 'anchors_neigh_int', 'anchors_neigh_border', 'anchors_pos_int',
                       'anchors_pos_ext', 'anchors_structure', 'structure_anchors'):
                walk(tag + 'prepared/' + nm, getattr(model, nm, None))
        batch = hotpath.full_split_batch(model, 'train')
        out = model.training_step(batch, 0)
        if 'nosync' in variants:
            kept.append((tag + 'loss', out['loss'].detach().clone()))
        else:
            ck(tag + 'loss', out['loss'])
        model.backward(None, out['loss'], None, 0)
        if p_ < args.detail_passes:
            for nm, p in model.named_parameters():
                if p.grad is not None:
                    ck(tag + 'grad/' + nm, p.grad)
        opt.step()
        opt.zero_grad(set_to_none=True)
        if p_ < args.detail_passes:
            for nm, p in model.named_parameters():
                ck(tag + 'param_after/' + nm, p)
    torch.cuda.synchronize()
    for nm, t in kept:
        ck(nm, t)
    for nm, p in model.named_parameters():
        ck('final/' + nm, p)
    with open(args.child, 'w') as f:
        json.dump({'loss': float(out['loss']), 'sums': sums}, f)


def driver(args):
    outdir = os.path.join(REPO, 'gpurun_out', 'xproc')
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ)
    if 'atomics_off' in args.variant:
        env['ROCBLAS_DEFAULT_ATOMICS_MODE'] = '0'
    results = []
    for i in range(args.procs):
        out = os.path.join(outdir, 'child_%s_%d.json' % ('-'.join(args.variant) or 'default', i))
        cmd = [sys.executable, os.path.abspath(__file__), '--child', out, '--passes', str(args.passes),
               '--detail-passes', str(args.detail_passes), '--nodes', str(args.nodes), '--subgraphs', str(args.subgraphs)]
        for v in args.variant:
            cmd += ['--variant', v]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print('child %d failed:\n%s' % (i, r.stdout[-3000:]))
            continue
        results.append(json.load(open(out)))
    if not results:
        return 1
    ref = results[0]['sums']
    groups = {}
    report = {'variant': args.variant, 'procs': len(results), 'passes': args.passes, 'losses': [r['loss'] for r in results],
              'first_difference': []}
    for i, r in enumerate(results):
        key = json.dumps(r['sums'])
        groups.setdefault(key, []).append(i)
        if i and r['sums'] != ref:
            first = next(((a[0], a[1], b[1]) for a, b in zip(ref, r['sums']) if a != b), ('(length)', len(ref), len(r['sums'])))
            ndiff = sum(1 for a, b in zip(ref, r['sums']) if a != b)
            report['first_difference'].append({'child': i, 'first': first[0], 'differing_checksums': ndiff, 'of': len(ref)})
    report['distinct_outcomes'] = len(groups)
    report['group_sizes'] = sorted((len(v) for v in groups.values()), reverse=True)
    report['identical'] = len(groups) == 1
    print(json.dumps(report, indent=1))
    with open(os.path.join(outdir, 'report_%s.json' % ('-'.join(args.variant) or 'default')), 'w') as f:
        json.dump(report, f, indent=1)
    return 0


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--child')
    ap.add_argument('--procs', type=int, default=10)
    ap.add_argument('--passes', type=int, default=4)
    ap.add_argument('--detail-passes', type=int, default=2)
    ap.add_argument('--nodes', type=int, default=1_000_000)
    ap.add_argument('--subgraphs', type=int, default=50_000)
    ap.add_argument('--variant', action='append', default=[])
    a = ap.parse_args()
    if a.child:
        child(a)
    else:
        sys.exit(driver(a))
