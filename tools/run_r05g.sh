export TMPDIR=/tmp
python -m pytest tests/test_gpu_hotpath.py -m gpu -x -q -k "pool_reuse or sparse_prepare" 2>&1 | tail -3
python tools/dtw_side_probe.py external 5 2>/dev/null | tail -1
python tools/dtw_side_probe.py internal 5 2>/dev/null | tail -1
SGNN_HIPCC_FLAGS=-DDTW_PROBE_COUNT python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1
python tools/dtw_budget.py external > gpurun_out/r05g_dtw_budget_external.json 2> gpurun_out/r05g_dtw_budget.err; cat gpurun_out/r05g_dtw_budget_external.json
python tools/dtw_budget.py internal > gpurun_out/r05g_dtw_budget_internal.json 2>> gpurun_out/r05g_dtw_budget.err
python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r05g_bench_default.json 2> gpurun_out/r05g_bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05g_bench_default.json'))
print(round(d['ms_per_step'],3), d.get('sequential_ms_per_step'), d.get('extra'), d['first_pass_ms'], d['warm_up_ms_at_construction'], d['first_pass_breakdown_ms'])
print({k:round(v,2) for k,v in d['stages_ms'].items()})
print('shard', d['shard6250']['ms_per_step'], d['shard6250']['stages_ms'])
print({k:(v.get('ms_per_step_replayed'), v.get('kernels_per_step')) for k,v in d['configs'].items() if isinstance(v,dict)})
PY
