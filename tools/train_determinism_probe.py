"""Is the training half bit-reproducible at the benchmark's size?  One prepared pass, the same parameters: forward + backward
repeated, logits / loss / every gradient compared bit for bit between repetitions (names the tensors that differ).
usage: python tools/train_determinism_probe.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import hotpath, ops, synthetic
from subgnn_amd.SubGNN import SubGNN
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n, m, S = 1_000_000, 10, 50_000
dev = torch.device('cuda:0')
edges = synthetic.barabasi_albert_edges(n, m, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
hp = dict(bench.ALL_DENSITY_HP)
hp['lin_dropout'] = 0.0
emb = torch.randn(n, hp['node_embed_size'], generator=torch.Generator().manual_seed(0))
labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
torch.manual_seed(0)
model = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]},
                           emb, num_classes=3)
model.train()
hotpath.prepare_sparse(model, 'train')
batch = hotpath.full_split_batch(model, 'train')
ref = None
for r in range(reps):
    for p in model.parameters():
        p.grad = None
    out = model.training_step(batch, 0)
    out['loss'].backward()
    torch.cuda.synchronize()
    cur = {'loss': out['loss'].detach().clone()}
    for nm, p in model.named_parameters():
        if p.grad is not None:
            cur[nm] = p.grad.detach().clone()
    if ref is None:
        ref = cur
        print('rep 0 loss', float(cur['loss']))
        continue
    diff = [(k, float((cur[k].double() - ref[k].double()).abs().max())) for k in ref if not torch.equal(cur[k], ref[k])]
    print('rep', r, 'loss', float(cur['loss']), 'differing tensors:', diff[:12])

# ---- the preparation: the same pass prepared again must give the same tensors (the draws depend on the seed only) ----------
def snapshot():
    out = {}
    def walk(prefix, o):
        if isinstance(o, torch.Tensor):
            out[prefix] = o.detach().clone()
        elif isinstance(o, dict):
            for k, v in o.items():
                walk('%s[%r]' % (prefix, k), v)
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                walk('%s[%d]' % (prefix, i), v)
    for nm in ('train_cc_ids', 'train_neigh_pos_similarities', 'train_int_struc_similarities', 'train_bor_struc_similarities',
               'anchors_neigh_int', 'anchors_neigh_border', 'anchors_pos_int', 'anchors_pos_ext', 'anchors_structure',
               'structure_anchors', '_mpn_edge_plans'):
        walk(nm, getattr(model, nm, None))
    return out


first = snapshot()
for r in range(1, reps):
    hotpath.prepare_sparse(model, 'train')
    torch.cuda.synchronize()
    cur = snapshot()
    diff = [k for k in first if k not in cur or cur[k].shape != first[k].shape or not torch.equal(cur[k], first[k])]
    print('prepare', r, 'differing tensors:', diff[:10])

# ---- clip + Adam from the same gradients -----------------------------------------------------------------------------------
from subgnn_amd import optim
state = {nm: p.detach().clone() for nm, p in model.named_parameters()}
for p in model.parameters():
    p.grad = None
results = []
for r in range(4):
    with torch.no_grad():
        for nm, p in model.named_parameters():
            p.copy_(state[nm])
    opt = optim.ClipAdam(model.parameters(), hp['learning_rate'], max_norm=hp['grad_clip'])
    for k in range(3):
        out = model.training_step(hotpath.full_split_batch(model, 'train'), 0)
        out['loss'].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    results.append({nm: p.detach().clone() for nm, p in model.named_parameters()})
    if r:
        diff = [(nm, float((results[r][nm].double() - results[0][nm].double()).abs().max())) for nm in state
                if not torch.equal(results[r][nm], results[0][nm])]
        print('3 steps of clip + Adam, repetition', r, 'loss', float(out['loss']), 'differing parameters:', diff[:10])
