"""Time the additive-attention score kernels (f32 MFMA vs half-operand MFMA) at shard and batch sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from subgnn_amd import ops
dev = 'cuda:0'
for R, H in ((50000, 420), (50000, 615), (448, 615), (64 * 20, 615)):
    X = torch.randn(R, H, device=dev); U = torch.randn(H, H, device=dev) / H ** 0.5
    q = torch.randn(R, H, device=dev); v = torch.randn(H, device=dev)
    res = {}
    ops.ATTN_F16_MIN_ROWS = 0
    for half in (False, True):
        with torch.no_grad():
            ops.attn_scores(X, U, q, v, 1, half_operands=half); torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): ops.attn_scores(X, U, q, v, 1, half_operands=half)
            torch.cuda.synchronize(); res[half] = (time.perf_counter() - t) / 10 * 1e3
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(10): (torch.tanh(q + X @ U) * v).sum(1)
        torch.cuda.synchronize(); lib = (time.perf_counter() - t0) / 10 * 1e3
    print('R=%d H=%d  exact (GEMM + fused epilogue) %.3f ms  f16 mfma kernel %.3f ms  (torch: GEMM + tanh + reduce %.3f ms)  %.1f TFLOP/s f16' %
          (R, H, res[False], res[True], lib, 2.0 * R * H * H / res[True] / 1e9))
