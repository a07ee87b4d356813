import time, torch, sys
sys.path.insert(0, '/root/repo')
t0=time.perf_counter(); torch.zeros(1, device='cuda'); torch.cuda.synchronize(); print('cuda init %.0f ms' % (1e3*(time.perf_counter()-t0)))
from subgnn_amd import _lib, ops
t0=time.perf_counter(); lib=_lib.load(); print('lib load %.0f ms' % (1e3*(time.perf_counter()-t0)))
import ctypes
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names=['degree_sequence','graph_sets','samplers','similarity','dtw','embed','mpn','attention','lstm','probe','scatter','update','optim','readout','loss','head']
libc=ctypes.CDLL(_lib.LIB_PATH)
for n in names:
    f=getattr(libc,'sgnn_warm_'+n); f.argtypes=[ctypes.c_void_p]
    t0=time.perf_counter(); f(st); torch.cuda.synchronize(); print('  warm %-16s %.1f ms' % (n, 1e3*(time.perf_counter()-t0)))
t0=time.perf_counter(); ops.warm_up(torch.device('cuda:0')); print('ops.warm_up rest (torch kernels + BLAS) %.0f ms' % (1e3*(time.perf_counter()-t0)))
