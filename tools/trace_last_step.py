"""Print the kernel sequence of the last N microseconds window of a rocprofv3 kernel trace CSV
(name, duration, gap to previous) -- used to see what one replayed training step is made of."""
import csv
import sys

path, n_last = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n_last:]
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print('%8.1f us  gap %7.1f  grid %8s wg %5s  %s' % ((e - s) / 1e3, gap, r.get('Grid_Size_X', r.get('Grid_Size', '')),
                                                       r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r['Kernel_Name'][:100]))
    prev_end = e
