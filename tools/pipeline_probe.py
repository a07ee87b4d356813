"""Feasibility: how much does a pass gain when the NEXT pass's sampling + similarity stages (hotpath.prepare_sparse without
its table-dependent tail) run on a side stream while THIS pass's forward / backward / Adam run on the main stream?
Two models with separate per-pass state stand in for the two passes in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from subgnn_amd import ops, hotpath, synthetic
from subgnn_amd.SubGNN import SubGNN

dev = torch.device('cuda:0')
n, S = 1_000_000, 50_000
rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, 10, seed=42), n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
torch.manual_seed(0)
emb = torch.randn(n, 64, device=dev)
hp = dict(bench.ALL_DENSITY_HP)
hp['node_embed_size'] = 64
labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
models = []
for _ in range(2):
    m = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]},
                           emb, num_classes=3)
    m.train()
    models.append((m, m.configure_optimizers(), [p for p in m.parameters() if p.requires_grad]))

def train(k):
    m, opt, params = models[k]
    out = m.training_step(hotpath.full_split_batch(m, 'train'), 0)
    out['loss'].backward()
    torch.nn.utils.clip_grad_norm_(params, hp['grad_clip'])
    opt.step(); opt.zero_grad(set_to_none=True)

for k in (0, 1):
    for _ in range(3):
        hotpath.prepare_sparse(models[k][0], 'train'); train(k)
torch.cuda.synchronize()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

def seq():
    hotpath.prepare_sparse(models[0][0], 'train')
    train(1)

def par():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        hotpath.prepare_sparse(models[0][0], 'train')
    train(1)
    main.wait_stream(side)

def wall(f, reps=6):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3

print('prepare(A) then train(B), one stream : %.2f ms' % wall(seq))
print('prepare(A) || train(B), two streams  : %.2f ms' % wall(par))
print('prepare(A) then train(B), one stream : %.2f ms' % wall(seq))
print('prepare(A) || train(B), two streams  : %.2f ms' % wall(par))
