#!/usr/bin/env python3
"""Does the DEVICE stall on its own?  A stream of identical small torch kernels (no hipGraph, no library of this repository, no
allocation inside the loop), a HIP event every 50 launches: the distribution of event-to-event times.  On a quiet device every
interval is the same; intervals of tens of milliseconds are the box's, not the program's (tools/epoch_stall_probe.py found single
recorded 1 ms training steps taking 49 ms between their events).      python tools/device_stall_probe.py [seconds]"""
import sys
import time
import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
x = torch.zeros(1 << 16, device='cuda')
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    for _ in range(50):
        x.add_(1.0)
    if len(evs) % 64 == 0:
        torch.cuda.synchronize()                       # (keeps the queue short: the host never runs far ahead)
torch.cuda.synchronize()
d = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1))
n = len(d)
print('%d intervals of 50 launches over %.1f s: median %.3f ms, p99 %.3f, max %.3f; intervals beyond 5 ms: %s'
      % (n, secs, d[n // 2], d[int(n * 0.99)], d[-1], [round(v, 1) for v in d if v > 5.0]))
