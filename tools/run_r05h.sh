export TMPDIR=/tmp
rebuild() { SGNN_HIPCC_FLAGS="$1" python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1; }
for f in "-DDTW_PROBE_NO_FINEST" "-DDTW_PROBE_NO_COARSE" "-DDTW_PROBE_NO_BACKTRACK" ""; do
  rebuild "$f"; echo "flags [$f]"; python tools/dtw_side_probe.py external 5 2>/dev/null | tail -1
done
