export TMPDIR=/tmp
rebuild() { SGNN_HIPCC_FLAGS="$1" python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1; }
bash tools/run_dtw_pmc.sh r05k_full > /dev/null 2>&1
rebuild "-DDTW_PROBE_NO_FINEST"
python tools/dtw_side_probe.py external 5 2>/dev/null | tail -1
bash tools/run_dtw_pmc.sh r05k_nofinest > /dev/null 2>&1
rebuild ""
python - <<'PY'
import json
for t in ('r05k_full','r05k_nofinest'):
    d=json.load(open('gpurun_out/%s_dtw_pmc.json'%t))
    for k,v in d.items():
        if 'dtw_similarity' in k:
            print(t, {a:round(v[a],3) for a in ('valu_instructions_per_64_pairs','salu_instructions_per_64_pairs','frac_wave_cycles_issuing','frac_wave_cycles_waiting_waitcnt_or_barrier','frac_wave_cycles_issue_stalled','simd_valu_busy_from_grbm') if a in v}, 'lds/wave', round(v.get('lds_instructions_per_wave',0),1))
PY
