#!/bin/bash
# Rebuild dtw.hip with different settings and time one side of the benchmark's DTW call (tools/dtw_side_probe.py).
#   bash tools/tune_dtw.sh "-DDTW_PROBE_NO_FINEST" "-DDTW_PROBE_NO_COARSE" ...
for flags in "" "$@"; do
  touch subgnn_amd/csrc/dtw.hip
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build > /dev/null 2>&1
  echo "[$flags]: $(python tools/dtw_side_probe.py external 5 2>&1 | grep -E '^external') | $(python tools/dtw_side_probe.py internal 5 2>&1 | grep -E '^internal')"
done
touch subgnn_amd/csrc/dtw.hip
python -m subgnn_amd.build > /dev/null 2>&1
