export TMPDIR=/tmp
rebuild() { SGNN_HIPCC_FLAGS="$1" python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1; }
for f in "" "-DDTW_KEY_ORDER=1" "-DDTW_KEY_ORDER=2" "-DDTW_KEY_STEPS=4.f" "-DDTW_KEY_STEPS=16.f" "-DDTW_KEY_STEPS=4.f -DDTW_KEY_ORDER=1" "-DDTW_KEY_STEPS=2.f"; do
  rebuild "$f -DDTW_PROBE_COUNT"
  echo "flags [$f]"
  python tools/dtw_budget.py external 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('  union/own', [(k, v['cells_evaluated_per_wavefront(union)'], v['cells_needed_per_pair(own window)']) for k,v in d['levels'].items()])"
  rebuild "$f"
  python tools/dtw_side_probe.py external 5 2>/dev/null | tail -1
done
rebuild ""
