"""Which call of hotpath.prepare_pass invalidates a stream capture?  Wraps every function of subgnn_amd.ops /
anchor_patch_samplers / subgraph_utils / gamma and asks HIP for the capture status of the current stream after each call.
usage: python tools/capture_probe.py [n_nodes] [n_subgraphs]"""
import ctypes
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from subgnn_amd import hotpath, ops, optim, synthetic, subgraph_utils, gamma
from subgnn_amd import anchor_patch_samplers as aps
from subgnn_amd.SubGNN import SubGNN
import bench

hip = ctypes.CDLL('libamdhip64.so')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device('cuda:0')
edges = synthetic.barabasi_albert_edges(n, 10, seed=42)
rowptr, col = synthetic.sorted_csr(edges, n)
subs = synthetic.bfs_subgraphs(rowptr, col, S, 20, seed=1000)
g = ops.DeviceGraph(rowptr, col, np.arange(1, n + 1, dtype=np.int32), dev)
hp = dict(bench.ALL_DENSITY_HP, lin_dropout=0.0)
emb = torch.randn(n, hp['node_embed_size'], generator=torch.Generator().manual_seed(0))
labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
model = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]},
                           emb, num_classes=3)
model.train()
for _ in range(2):
    hotpath.prepare_sparse(model, 'train')
torch.cuda.synchronize()


def status():
    st = ctypes.c_int(0)
    rc = hip.hipStreamIsCapturing(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(st))
    return rc, st.value


seen = {'bad': None}


def wrap(mod, name, fn):
    def w(*a, **k):
        out = fn(*a, **k)
        if seen['bad'] is None:
            rc, st = status()
            if rc != 0 or st == 2:
                seen['bad'] = '%s.%s (hipStreamIsCapturing rc %d status %d)' % (mod.__name__, name, rc, st)
                print('capture invalidated after', seen['bad'], flush=True)
        return out
    return w


for mod in (ops, aps, subgraph_utils, gamma):
    for name, fn in list(vars(mod).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith('__'):
            setattr(mod, name, wrap(mod, name, fn))
model.__dict__.setdefault('_bfs_status_pool', []).extend(torch.empty(4, dtype=torch.int32).pin_memory() for _ in range(8))
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr):
        st = hotpath.prepare_pass(model, 'train')
    print('capture ok')
    gr.replay()
    torch.cuda.synchronize()
    print('replay ok')
except Exception as ex:
    print('capture failed:', str(ex).splitlines()[0])
