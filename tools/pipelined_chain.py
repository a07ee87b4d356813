"""The training chain of one pipelined bench step from a rocprofv3 kernel trace: for every kernel on the queue that runs
adam_step_kernel -- gap to the previous kernel of that queue, duration, and the kernel of the OTHER queue(s) that overlapped
it longest.  usage: python tools/pipelined_chain.py kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [r for r in rows if 'adam_step_kernel' in r['Kernel_Name']]
q = adam[-1]['Queue_Id']
t0, t1 = adam[-3]['e'], adam[-2]['e']                 # the step before the last one
chain = [r for r in rows if r['Queue_Id'] == q and t0 <= r['s'] < t1]
other = [r for r in rows if r['Queue_Id'] != q and r['e'] > t0 and r['s'] < t1]


def short(n):
    n = re.sub(r'void |at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_\d+_NS::detail::', '', n)
    n = re.sub(r'elementwise_kernel_manual_unroll<\d+, \d+, gpu_kernel_impl(_nocast)?<', 'EW<', n)
    n = re.sub(r'vectorized_elementwise_kernel<\d+, ', 'VEW<', n)
    n = re.sub(r'Cijk_(\w{4})_(\w{4})_S_B_Bias_HA_S_SAV_UserArgs_(MT\w+?)_.*', r'GEMM \1 \2 \3', n)
    return n[:60]


prev = t0
tot_d = tot_g = 0
by_other = defaultdict(lambda: [0.0, 0.0, 0])
for r in chain:
    best, bo = None, 0
    for o in other:
        ov = min(o['e'], r['e']) - max(o['s'], prev)
        if ov > bo:
            best, bo = o, ov
    on = short(best['Kernel_Name'])[:28] if best else '-'
    g, d = (r['s'] - prev) / 1e3, (r['e'] - r['s']) / 1e3
    tot_d += d
    tot_g += max(g, 0)
    b = by_other[on]
    b[0] += d
    b[1] += max(g, 0)
    b[2] += 1
    print('t=%8.1f gap %7.1f dur %7.1f  %-60s | %s' % ((r['s'] - t0) / 1e3, g, d, short(r['Kernel_Name']), on))
    prev = max(prev, r['e'])
print('step %.1f us: %d kernels, kernel time %.1f, gaps %.1f' % ((t1 - t0) / 1e3, len(chain), tot_d, tot_g))
for k, (d, g, n) in sorted(by_other.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print('  beside %-30s %4d kernels: time %8.1f gaps %8.1f' % (k, n, d, g))
