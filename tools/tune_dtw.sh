#!/bin/bash
# Rebuild similarity.hip with different settings and time the benchmark's DTW launches (tools/dtw_probe.py).
for flags in "" "$@"; do
  touch subgnn_amd/csrc/similarity.hip
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build > /dev/null 2>&1
  echo "[$flags]: $(python tools/dtw_probe.py 3 20 2>&1 | grep -E '^internal|^external' | awk '{print $1, $2}' | tr '\n' ' ')"
done
touch subgnn_amd/csrc/similarity.hip
python -m subgnn_amd.build > /dev/null 2>&1
