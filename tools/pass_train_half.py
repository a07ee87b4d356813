"""The training half (component embeddings .. Adam) of tools/pass_kernels.py's listing, names shortened; total kernel time.
usage: python tools/pass_train_half.py [gpurun_out/pk_pass_kernels.txt]"""
import re
import sys

L = [l.rstrip() for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pk_pass_kernels.txt')]
dtw = [i for i, l in enumerate(L) if 'dtw_similarity' in l][-1]
tot = gaps = 0.0
for i, l in enumerate(L[dtw + 1:]):
    m = re.match(r't=\s*([\d.]+) gap\s*(-?[\d.]+) dur\s*([\d.]+)\s+grid (\S+)\s+(.*)', l)
    t, g, d, grid, n = m.groups()
    n = re.sub(r'elementwise_kernel_manual_unroll<\d+, \d+, gpu_kernel_impl(_nocast)?<', 'EW<', n)
    n = re.sub(r'vectorized_elementwise_kernel<\d+, ', 'VEW<', n)
    n = re.sub(r'Cijk_(\w{4})_(\w{4})_S_B_Bias_HA_S_SAV_UserArgs_(MT\w+?)_.*', r'GEMM \1 \2 \3', n)
    n = re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::', 'rocprim::', n)
    tot += float(d)
    gaps += max(float(g), 0.0)
    print('%3d %7.1f g%5.1f %8s %s' % (i, float(d), float(g), grid, n[:100]))
print('kernels %d, kernel time %.1f us, gaps %.1f us' % (len(L) - dtw - 1, tot, gaps))
