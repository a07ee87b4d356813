"""Every kernel of one bench pass, in launch order, from a rocprofv3 kernel-trace CSV: start offset, gap to the
previous kernel's end, duration, name (the pass before the last).  usage: pass_kernels.py trace.csv [from_kernel_substring]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'cc_labels_kernel' in r['Kernel_Name']]
p = rows[idx[-2]:idx[-1]]
start = sys.argv[2] if len(sys.argv) > 2 else None
t0 = int(p[0]['Start_Timestamp'])
prev_end = t0
on = start is None


def short(n):
    n = re.sub(r'void |at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_400001_NS::detail::', '', n)
    n = re.sub(r'std::array<char\*, (\d)ul>', r'arr\1', n)
    return n[:150]


for r in p:
    n = r['Kernel_Name']
    if not on and start in n:
        on = True
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if on:
        print('t=%8.1f gap %6.1f dur %7.1f  grid %-9s %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                                                          r.get('Grid_Size_X', r.get('Grid_Size', '')), short(n)))
    prev_end = max(prev_end, e)
