#!/bin/bash
# Rebuild graph_sets.hip with different border-BFS settings and time the border stage (tools/khop_probe.py).
#   bash tools/tune_khop.sh "-DK1_DEBUG_SKIP_DRAW" ...     (K1_DEBUG_* switches compile parts out: timing only, wrong results)
for flags in "" "$@"; do
  touch subgnn_amd/csrc/graph_sets.hip
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build > /dev/null 2>&1
  echo "[$flags]: $(python tools/khop_probe.py 2>&1 | grep -v amdgpu.ids | grep '43 slots' | head -1)"
done
touch subgnn_amd/csrc/graph_sets.hip
python -m subgnn_amd.build > /dev/null 2>&1
