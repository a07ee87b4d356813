#!/bin/bash
# Counter passes (separate rocprofv3 --pmc runs, no tracing) of the kernels whose name contains <substring>, for a probe command:
#   bash tools/run_pmc.sh <tag> <kernel substring> python3 tools/khop_probe.py      -> gpurun_out/<tag>_pmc.json (per-dispatch averages)
export TMPDIR=/tmp
O=gpurun_out
T=$1; K=$2; shift 2
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf $O/${T}_pmc_$i
  rocprofv3 --pmc $c --output-format csv -d $O/${T}_pmc_$i -- "$@" > $O/${T}_pmc_$i.log 2>&1
done
python tools/pmc_summary.py $O/${T}_pmc_1 $O/${T}_pmc_2 $O/${T}_pmc_3 $O/${T}_pmc_4 > $O/${T}_pmc_raw.json
python - "$T" "$K" <<'PY'
import json, sys
T, K = sys.argv[1], sys.argv[2]
d = json.load(open('gpurun_out/%s_pmc_raw.json' % T))
out = {}
for k, v in d.items():
    if K in k:
        c = {n: x['mean'] for n, x in v.items()}
        r = dict(c)
        wc = c.get('SQ_WAVE_CYCLES')
        if wc:
            for nm, key in (('issuing', 'SQ_ACTIVE_INST_ANY'), ('waiting_waitcnt_or_barrier', 'SQ_WAIT_ANY'), ('issue_stalled', 'SQ_WAIT_INST_ANY'),
                            ('valu_active', 'SQ_ACTIVE_INST_VALU'), ('lds_active', 'SQ_ACTIVE_INST_LDS'), ('scalar_active', 'SQ_ACTIVE_INST_SCA')):
                if key in c:
                    r['frac_wave_cycles_' + nm] = c[key] / wc
        if c.get('SQ_WAVES'):
            for nm, key in (('valu', 'SQ_INSTS_VALU'), ('salu', 'SQ_INSTS_SALU'), ('lds', 'SQ_INSTS_LDS'), ('vmem_rd', 'SQ_INSTS_VMEM_RD'), ('branch', 'SQ_INSTS_BRANCH')):
                if key in c:
                    r[nm + '_instructions_per_wave'] = c[key] / c['SQ_WAVES']
        out[k] = r
json.dump(out, open('gpurun_out/%s_pmc.json' % T, 'w'), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
