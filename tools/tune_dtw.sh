#!/bin/bash
# Rebuild the library with different compile-time DTW settings on the GPU box and time the
# DTW calls of tools/dtw_probe.py for 12-, 20- and 32-node subgraphs.  Usage: bash tools/tune_dtw.sh
for flags in "-DDTW_REG_BLOCKS=2048" "-DDTW_REG_BLOCKS=4096" "-DDTW_REG_BLOCKS=8192" "-DDTW_REG_BLOCKS=16384" "-DDTW_REG_BLOCKS=1024"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  for nx in 20; do
    echo "$flags nx=$nx: $(python tools/dtw_probe.py 3 $nx 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' ')"
  done
done
python -m subgnn_amd.build --force > /dev/null 2>&1
