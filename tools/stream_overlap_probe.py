"""Do chains of tiny kernels queued on different HIP streams overlap on this runtime?  The host is taken out of the
measurement by queueing everything behind a long kernel: device time from the long kernel's end to the last chain's end."""
import sys, time
import torch
dev = torch.device('cuda:0')
big = torch.randn(8192, 8192, device=dev)
x = [torch.randn(50000, 64, device=dev) for _ in range(8)]
w = torch.randn(64, 64, device=dev)
streams = [torch.cuda.Stream() for _ in range(8)]
main = torch.cuda.current_stream()

def chain(t, n, kind):
    for _ in range(n):
        if kind == 'tiny':
            t = t[:64] + 1.0
        elif kind == 'ew':
            t = torch.relu(t) + 1.0             # 2 kernels over 12.8 MB
        else:
            t = torch.relu(t @ w)               # GEMM 50000x64x64 + relu
    return t

def run(n_streams, per, kind):
    torch.cuda.synchronize()
    for _ in range(3):
        big @ big
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    outs = []
    if n_streams == 1:
        for i in range(8):
            outs.append(chain(x[i], per, kind))
    else:
        for i in range(8):
            s = streams[i % n_streams]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(chain(x[i], per, kind))
        for s in streams[:n_streams]:
            main.wait_stream(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)

for kind in ('tiny', 'ew', 'gemm'):
    for ns in (1, 2, 4, 8):
        run(ns, 20, kind)
        t = min(run(ns, 20, kind) for _ in range(3))
        print('%-5s 8 chains x 20 steps on %d stream(s): %.3f ms after the long kernel' % (kind, ns, t))
