#!/usr/bin/env python3
"""Where single epochs of a stand-in stall: every collector pass (gc.callbacks: generation, ms) and every call of the epoch loop's
pieces (recorded training step, recorded validation forward, validation_epoch_end) that takes more than 3 ms, with the epoch it
fell into.      python tools/epoch_stall_probe.py ppi_bp [epochs]"""
import gc
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from subgnn_amd import config, graph_step, hotpath, standins, train_config
from subgnn_amd import precompute_graph_metrics as pgm
from subgnn_amd.SubGNN import SubGNN, dataset_paths

name = sys.argv[1] if len(sys.argv) > 1 else 'ppi_bp'
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
P = standins.PRESETS[name]
hp = dict(P['hp'])
root = tempfile.mkdtemp(prefix=name + '_')
d, _ = standins.write_standin(root, name)
pgm.calculate_stats(d, shortest_paths=not P['sparse'], ego=not P['sparse'])
config.PROJECT_ROOT = root
torch.manual_seed(3)
m = SubGNN(dict(hp), **dataset_paths(name + '_standin'))
if P['sparse']:
    for sp in ('val', 'train'):
        hotpath.prepare_sparse(m, sp)
else:
    m.prepare_data()
events = []
t_gc = [0.0]


def on_gc(phase, info):
    if phase == 'start':
        t_gc[0] = time.perf_counter()
    else:
        events.append(('gc gen %d (collected %d)' % (info['generation'], info['collected']), time.perf_counter(), (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(on_gc)


def timed(cls, meth, label):
    f = getattr(cls, meth)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        dt = (time.perf_counter() - t) * 1e3
        if dt > 3.0:
            events.append((label, time.perf_counter(), dt))
        return r
    setattr(cls, meth, g)


dev_events = []                                          # (epoch-relative) HIP events around every recorded training step: DEVICE time
_orig_replay = graph_step.CapturedTrainStep.replay


def replay_with_events(self, idx):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = _orig_replay(self, idx)
    e1.record()
    dev_events.append((time.perf_counter(), e0, e1))
    return r


graph_step.CapturedTrainStep.replay = replay_with_events
timed(graph_step.CapturedTrainStep, 'replay', 'CapturedTrainStep.replay')
timed(train_config.Trainer, '_validation_outputs', 'Trainer._validation_outputs')
timed(SubGNN, 'validation_epoch_end', 'validation_epoch_end')
timed(torch.cuda, 'synchronize', 'torch.cuda.synchronize')
tr = train_config.Trainer(epochs, hp.get('grad_clip', 0.0), log=lambda *a, **k: None, hip_graph_step=True)
tr.phase_times = []
marks = []
orig_phase = tr._phase


def phase(rec, key, t):
    r = orig_phase(rec, key, t)
    marks.append((key, time.perf_counter()))
    return r


tr._phase = phase
t0 = time.perf_counter()
tr.fit(m, prepared=True)
print('epochs (ms):', [round(1e3 * sum(v for k, v in r.items() if k.endswith('_s')), 1) for r in tr.phase_times])
ends = [t for k, t in marks if k == 'validation_epoch_end_s']
for label, t, dt in events:
    ep = sum(1 for e in ends if e < t)
    if ep >= 1:
        print('epoch %2d  %-32s %7.1f ms' % (ep, label, dt))

torch.cuda.synchronize()
print('device time of single recorded training steps beyond 3 ms (HIP events), and the gaps between consecutive steps beyond 3 ms:')
prev = None
for t, e0, e1 in dev_events:
    ep = sum(1 for e in ends if e < t)
    d = e0.elapsed_time(e1)
    if d > 3.0 and ep >= 1:
        print('epoch %2d  step on the device %7.1f ms' % (ep, d))
    if prev is not None:
        gap = prev.elapsed_time(e0)
        if gap > 3.0 and ep >= 1:
            print('epoch %2d  device gap before a step %7.1f ms' % (ep, gap))
    prev = e1
