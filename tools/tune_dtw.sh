#!/bin/bash
# Rebuild the library with different compile-time DTW settings on the GPU box and time the
# DTW calls of tools/dtw_probe.py for 12-, 20- and 32-node subgraphs.  Usage: bash tools/tune_dtw.sh
for flags in "-DDTW_BRANCHLESS_ROWS=0" "-DDTW_BRANCHLESS_ROWS=1" "-DDTW_BRANCHLESS_ROWS=0 -DDTW_REG_BLOCKS=512" "-DDTW_BRANCHLESS_ROWS=0 -DDTW_REG_BLOCKS=2048" "-DDTW_BRANCHLESS_ROWS=0 -DDTW_MINB20=3"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  for nx in 12 20 32; do
    echo "$flags nx=$nx: $(python tools/dtw_probe.py 3 $nx 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
python -m subgnn_amd.build --force > /dev/null 2>&1
