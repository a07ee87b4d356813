"""Where the HOST spends a pass of the emulated strong rank (bench.time_strong_rank): cProfile over the whole call (60 timed steps
dominate).  usage: python tools/strong_host_profile.py"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from subgnn_amd import ops, hotpath                              # noqa: E402
from subgnn_amd.SubGNN import SubGNN                            # noqa: E402

sys.argv = sys.argv[:1]
args = bench.parse()
rowptr, col, subs, _, _ = bench.build_inputs(args, 0, 1)
dev = torch.device('cuda', 0)
g = ops.DeviceGraph(rowptr, col, np.arange(1, args.nodes + 1, dtype=np.int32), dev)
torch.manual_seed(0)
emb = torch.randn(args.nodes, args.embed, device=dev)
hp = dict(bench.ALL_DENSITY_HP)
S = len(subs)
labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
full = SubGNN.from_memory(dict(hp), g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3)
hotpath.prepare_sparse(full, 'train')
bench.time_strong_rank(g, subs, labels, emb, hp, full, steps=5, warmup=1)      # (imports, code objects: not the profile's business)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
line = bench.time_strong_rank(g, subs, labels, emb, hp, full, steps=60, warmup=3)
pr.disable()
print('ms per step', line['ms_per_step_device'], line['stages_ms'])
print('host', line['host_ms_per_stage(time the host needs to queue it)'])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:12000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
print(s.getvalue()[:6000])
