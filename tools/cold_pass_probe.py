import os, sys, time
sys.path.insert(0, '/root/repo')
t00 = time.perf_counter()
import numpy as np, torch
import bench
from subgnn_amd import ops, hotpath, synthetic, optim
from subgnn_amd.SubGNN import SubGNN
print('imports %.0f ms' % ((time.perf_counter() - t00) * 1e3))
n, S = 1_000_000, 50_000
rowptr, col = synthetic.sorted_csr(synthetic.barabasi_albert_edges(n, 10, seed=42), n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
dev = torch.device('cuda:0')
def T(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print('%-40s %8.1f ms' % (name, (time.perf_counter() - t) * 1e3)); return r
T('cuda init (first sync)', lambda: torch.zeros(1, device=dev))
g = T('DeviceGraph upload', lambda: ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev))
emb = T('randn table', lambda: torch.randn(n, 64, device=dev))
labels = torch.randint(0, 3, (S,))
hp = dict(bench.ALL_DENSITY_HP)
model = T('model build', lambda: SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3))
model.train()
opt = optim.ClipAdam(model.parameters(), hp['learning_rate'], max_norm=hp['grad_clip'])
for k in range(3):
    tm = hotpath.StageTimer(True)
    st = T('pass %d: prepare_pass' % k, lambda: hotpath.prepare_pass(model, 'train', tm))
    if k == 0:
        print('   host ms per stage:', {a: round(b, 1) for a, b in tm.host_summary().items()})
        print('   device ms per stage:', {a: round(b, 1) for a, b in tm.summary().items()})
    T('pass %d: install' % k, lambda: hotpath.install_pass(model, st))
    out = T('pass %d: forward' % k, lambda: model.training_step(hotpath.full_split_batch(model, 'train'), 0))
    T('pass %d: backward' % k, lambda: model.backward(None, out['loss'], None, 0))
    T('pass %d: optimizer' % k, lambda: (opt.step(), opt.zero_grad(set_to_none=True)))
