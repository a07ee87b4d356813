"""Where the HOST spends a 6 250-subgraph pass (training half replayed, passes pipelined): cProfile over the steady-state steps
of bench.time_shard's loop.  usage: python tools/shard_host_profile.py [subgraphs=6250] [steps=30]"""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from subgnn_amd import ops, hotpath, optim                      # noqa: E402
from subgnn_amd.SubGNN import SubGNN                            # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sys.argv = sys.argv[:1]
args = bench.parse()
rowptr, col, subs, _, _ = bench.build_inputs(args, 0, 1)
subs = subs[:S]
dev = torch.device('cuda', 0)
g = ops.DeviceGraph(rowptr, col, np.arange(1, args.nodes + 1, dtype=np.int32), dev)
torch.manual_seed(0)
emb = torch.randn(args.nodes, args.embed, device=dev)
hp = dict(bench.ALL_DENSITY_HP)
labels = torch.randint(0, 3, (S,), generator=torch.Generator().manual_seed(0))
model = SubGNN.from_memory(hp, g, {'train': subs, 'val': [], 'test': []}, {'train': labels, 'val': labels[:0], 'test': labels[:0]}, emb, num_classes=3)
model.train()
opt = optim.ClipAdam(model.parameters(), hp['learning_rate'], max_norm=hp['grad_clip'], capturable=True)
trainer = hotpath.CapturedTraining(model, opt, 'train', warmup=1)
pipe = hotpath.PassPipeline(model, 'train', None)


def step():
    pipe.install(None, installer=trainer.install)
    installed = torch.cuda.Event()
    installed.record()
    trainer.step()
    pipe.start(False, after=installed)


for _ in range(2):
    pipe.start()
for _ in range(6):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
print('ms per step (profiled)', 1e3 * (time.perf_counter() - t0) / steps)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
print(s.getvalue()[:5000])
