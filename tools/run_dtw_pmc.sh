#!/bin/bash
# Counter passes of the DTW kernel on the benchmark's external side (tools/dtw_side_probe.py): issue / wait split of the
# wave-cycles, instruction mix, LDS activity.  Separate passes (8 SQ counters per pass), no tracing.
#   bash tools/run_dtw_pmc.sh [tag]     -> gpurun_out/<tag>_dtw_pmc.json
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r03}
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf $O/${T}_dtwpmc_$i
  rocprofv3 --pmc $c --output-format csv -d $O/${T}_dtwpmc_$i -- python3 tools/dtw_side_probe.py external 3 > $O/${T}_dtwpmc_$i.log 2>&1
done
python tools/pmc_summary.py $O/${T}_dtwpmc_1 $O/${T}_dtwpmc_2 $O/${T}_dtwpmc_3 > $O/${T}_dtw_pmc_raw.json
python - "$T" <<'PY'
import json, sys
T = sys.argv[1]
d = json.load(open('gpurun_out/%s_dtw_pmc_raw.json' % T))
out = {'what': 'rocprofv3 --pmc passes (tools/run_dtw_pmc.sh) on tools/dtw_side_probe.py external 3: the external-side DTW launch of '
               'the benchmark (50k components x 210 anchor patches = 164 010 wavefront tasks of 64 pairs); per-dispatch averages'}
for k, v in d.items():
    if 'dtw_similarity' in k:
        c = {n: x['mean'] for n, x in v.items()}
        r = dict(c)
        wc = c.get('SQ_WAVE_CYCLES')
        if wc:
            r['frac_wave_cycles_issuing'] = c.get('SQ_ACTIVE_INST_ANY', 0) / wc
            r['frac_wave_cycles_waiting_waitcnt_or_barrier'] = c.get('SQ_WAIT_ANY', 0) / wc
            r['frac_wave_cycles_issue_stalled'] = c.get('SQ_WAIT_INST_ANY', 0) / wc
            r['frac_wave_cycles_valu_active'] = c.get('SQ_ACTIVE_INST_VALU', 0) / wc
        if c.get('SQ_WAVES') and c.get('SQ_INSTS_VALU'):
            r['valu_instructions_per_wave'] = c['SQ_INSTS_VALU'] / c['SQ_WAVES']
            r['salu_instructions_per_wave'] = c.get('SQ_INSTS_SALU', 0) / c['SQ_WAVES']
            r['lds_instructions_per_wave'] = c.get('SQ_INSTS_LDS', 0) / c['SQ_WAVES']
            tasks = 210 * ((50000 + 63) // 64)
            r['valu_instructions_per_64_pairs'] = c['SQ_INSTS_VALU'] / tasks
            r['salu_instructions_per_64_pairs'] = c.get('SQ_INSTS_SALU', 0) / tasks
        # occupancy of the shipped instantiation (DTW_MINB20 = 3 workgroups of 256 threads per CU = 3 wavefronts per SIMD)
        r['wavefronts_per_simd'] = 3
        if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_ACTIVE_INST_VALU'):
            # SQ_ACTIVE_INST_* count in units of 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            r['simd_valu_busy_from_grbm'] = 4 * c['SQ_ACTIVE_INST_VALU'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)
        out[k] = r
json.dump(out, open('gpurun_out/%s_dtw_pmc.json' % T, 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
