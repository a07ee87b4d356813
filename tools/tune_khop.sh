#!/bin/bash
# Rebuild with different border-BFS settings and time the border stage (tools/khop_probe.py).
for flags in "" "$@"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "[$flags]: $(python tools/khop_probe.py 2>&1 | grep -v amdgpu.ids | grep 'count-only, LDS\|43 slots' | tr '\n' ' ')"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
