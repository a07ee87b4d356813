#!/usr/bin/env python3
"""profiles/rNN_degseq_traffic.json from the outputs of tools/run_hbm_probe.sh: the timing line of every
probe run (gpurun_out/rNN_hbm_<case>.json) joined with the per-kernel counter averages
(gpurun_out/rNN_pmc_<case>.json).  FETCH_SIZE is scaled by the factor its own calibration copy shows
(known bytes / reported bytes, 4 B/lane -- the gather's access width); WRITE_SIZE is used as reported
(the calibration copy confirms it).

    python tools/make_traffic_profile.py gpurun_out r02 > profiles/r02_degseq_traffic.json
"""
import json
import sys

PEAK = 8000.0


def case(d, tag, name):
    t = json.load(open('%s/%s_hbm_%s.json' % (d, tag, name)))
    p = json.load(open('%s/%s_pmc_%s.json' % (d, tag, name)))
    cal = p['probe_copy_kernel<unsigned int>']
    f_fetch = t['calibration']['bytes_read'] / (cal['FETCH_SIZE']['mean'] * 1024.0)
    f_write = t['calibration']['bytes_written'] / (cal['WRITE_SIZE']['mean'] * 1024.0)
    out = {k: t[k] for k in ('graph', 'nnz', 'csr_bytes', 'family', 'sets', 'set_nodes', 'mean_member_degree',
                             'max_member_degree', 'distinct_members', 'distinct_list_bytes', 'forms_agree')}
    out['calibration'] = {'copy_bytes_read': t['calibration']['bytes_read'],
                          'FETCH_SIZE_KB_reported_4B_per_lane': cal['FETCH_SIZE']['mean'],
                          'FETCH_SIZE_KB_reported_16B_per_lane': p['probe_copy_kernel<HIP_vector_type<unsigned int, 4u> >']['FETCH_SIZE']['mean'],
                          'WRITE_SIZE_KB_reported': cal['WRITE_SIZE']['mean'],
                          'fetch_factor': f_fetch, 'write_factor': f_write,
                          'copy_GBs_read_plus_write': t['calibration']['copy_4B_per_lane']['GBs_read_plus_write']}
    for form, kern in (('streaming', 'degseq_wave_kernel<true, false, false>'), ('shipped_search', 'degseq_wave_kernel<true, false, true>')):
        c = p[kern]
        ms = t[form]['ms_per_launch']
        mem = c['FETCH_SIZE']['mean'] * 1024.0 * f_fetch + c['WRITE_SIZE']['mean'] * 1024.0 * f_write
        hit, miss = c['TCC_HIT_sum']['mean'], c['TCC_MISS_sum']['mean']
        out[form] = {'kernel': kern, 'ms_per_launch': ms,
                     'algorithmic_bytes_per_launch': t[form]['algorithmic_bytes_per_launch'],
                     'algorithmic_GBs': t[form]['achieved_GBs'], 'algorithmic_frac_of_8TBs': t[form]['frac_of_8TBs'],
                     'FETCH_SIZE_KB_reported': c['FETCH_SIZE']['mean'], 'WRITE_SIZE_KB_reported': c['WRITE_SIZE']['mean'],
                     'memory_side_bytes_per_launch': mem, 'memory_side_GBs': mem / ms / 1e6,
                     'memory_side_frac_of_8TBs': mem / ms / 1e6 / PEAK,
                     'traffic_over_algorithmic': mem / t[form]['algorithmic_bytes_per_launch'],
                     'l2_hit_rate': hit / (hit + miss)}
    return out


def main(d, tag):
    res = {
        'what': 'sgnn_degree_sequence (structure-channel CSR gather): time per launch (HIP events, 20 back-to-back launches, '
                'un-profiled run) and memory-side traffic per launch (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum '
                'TCC_MISS_sum, three separate passes, 3 profiled launches each), tools/run_hbm_probe.sh',
        'counter_note': 'FETCH_SIZE/WRITE_SIZE count the L2\'s memory-side requests (Infinity-Cache hits included); on gfx950 '
                        'FETCH_SIZE reports half the bytes of a streamed read -- the calibration copy below shows the same factor '
                        'for 4 and for 16 bytes per lane -- so memory_side_bytes = FETCH_SIZE x fetch_factor + WRITE_SIZE',
        'benchmark_graph': case(d, tag, 'bench'),
        'out_of_cache': {'bfs_sets': case(d, tag, 'bfs'), 'random_sets': case(d, tag, 'random')},
    }
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
