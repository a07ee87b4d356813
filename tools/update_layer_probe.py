import torch, time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from subgnn_amd import ops
dev = 'cuda:0'
R, D = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50000, 64)
cc = torch.randn(R, D, device=dev, requires_grad=True)
agg = torch.randn(R, D, device=dev, requires_grad=True)
lin = torch.nn.Linear(2 * D, D).to(dev)
gout = torch.randn(R, D, device=dev)
FUSED = len(sys.argv) > 1 and sys.argv[1] == 'fused'
def fwd():
    if FUSED:
        return ops.update_layer(cc, agg, lin.weight, lin.bias)
    return F.relu(ops.linear(torch.cat([cc, agg], dim=1), lin.weight, lin.bias))
def fb():
    out = fwd()
    out.backward(gout)
    cc.grad = None; agg.grad = None; lin.weight.grad = None; lin.bias.grad = None
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.randn(8192, 8192, device=dev); big @ big        # host runs ahead behind this
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print('fwd only (no grad) us', t(fwd))
print('fwd+bwd us', t(fb))

# ---- the three kernels of the fused form, one by one ----
from subgnn_amd import _lib
lib = _lib.load()
P = ops._ptr
out = ops.update_layer(cc.detach(), agg.detach(), lin.weight.detach(), lin.bias.detach())
gx, ga = torch.empty_like(out), torch.empty_like(out)
gW, gb = torch.empty_like(lin.weight), torch.empty_like(lin.bias)
wsb = lib.sgnn_update_bwd_workspace_bytes(R, D)
ws = torch.empty(wsb // 4 + 1, device=dev)
W_, b_ = lin.weight.detach(), lin.bias.detach()
x_, a_ = cc.detach(), agg.detach()
with torch.no_grad():
    print('sgnn_update_fwd us', t(lambda: lib.sgnn_update_fwd(P(x_), P(a_), P(W_), P(b_), R, D, P(out), ops._stream())))
    print('sgnn_update_bwd dx only us', t(lambda: lib.sgnn_update_bwd(P(gout), P(out), P(x_), P(a_), P(W_), R, D, P(gx), P(ga), None, None, P(ws), wsb, ops._stream())))
    print('sgnn_update_bwd dW, db only us', t(lambda: lib.sgnn_update_bwd(P(gout), P(out), P(x_), P(a_), P(W_), R, D, None, None, P(gW), P(gb), P(ws), wsb, ops._stream())))
