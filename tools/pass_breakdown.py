"""Per-stage kernel time of one bench pass from a rocprofv3 kernel-trace CSV: the pass before the
last (the last one is followed by the CPU-baseline leg), HIP kernels of this library vs everything
else (torch glue, library GEMMs), and the largest 'other' kernels per stage.
usage: pass_breakdown.py trace.csv [min_us_to_list]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'cc_labels_kernel' in r['Kernel_Name']]
p = rows[idx[-2]:idx[-1]]
ours = ('cc_labels', 'cc_compact', 'cc_embed', 'choice_ragged', 'msbfs', 'triangular', 'degseq', 'khop', 'dtw_', 'mpn_',
        'masked_sum', 'sample_anchors', 'patch_in_border', 'attn_scores', 'sp_sim', 'min_hops', 'sort_sets', 'scatter_runs',
        'scatter_chains', 'lstm_', 'update_', 'head_')
marks = [('triangular_walks', 'patches'), ('msbfs_init', 'position'), ('triangular_walks', 'walks'), ('sort_sets', 'border'),
         ('khop_border', 'border'), ('degseq_wave_kernel<true, false', 'degseq+dtw_prep'), ('dtw_pyramid', 'dtw'),
         ('cc_embed_fwd_kernel', 'fwd+bwd+opt')]
order = ['components', 'patches', 'position', 'walks', 'border', 'degseq+dtw_prep', 'dtw', 'fwd+bwd+opt']
stage = 'components'
tot = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
names = collections.defaultdict(collections.Counter)
t0 = int(p[0]['Start_Timestamp'])
big = []
for r in p:
    n = r['Kernel_Name']
    for m, s in marks:
        if m in n and order.index(s) == order.index(stage) + 1:          # stages follow each other in this order
            stage = s
            break
        if m in n and s in ('border',) and order.index(s) > order.index(stage):
            stage = s
            break
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    o = any(k in n for k in ours)
    tot[stage][0 if o else 1] += d
    tot[stage][2 if o else 3] += 1
    if not o:
        names[stage][n[:90]] += d
    if d >= min_us:
        big.append(((int(r['Start_Timestamp']) - t0) / 1e3, d, stage, n[:120]))
print('pass span %.1f us, kernel time %.1f us' % ((int(p[-1]['End_Timestamp']) - t0) / 1e3, sum(v[0] + v[1] for v in tot.values())))
for s in order:
    v = tot[s]
    print('%-16s ours %7.0f us (%3d)   other %7.0f us (%3d)' % (s, v[0], v[2], v[1], v[3]))
    for n, d in names[s].most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
        print('        %6.0f  %s' % (d, n))
print()
for t, d, s, n in big:
    print('t=%8.1f %8.1f us  %-16s %s' % (t, d, s, n))
