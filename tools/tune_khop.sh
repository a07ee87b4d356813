#!/bin/bash
# Rebuild with different numbers of 64-edge chunks in flight per wavefront and time the border stage.
for flags in "-DKB_INFLIGHT=2" "-DKB_INFLIGHT=4" "-DKB_INFLIGHT=8"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/khop_probe.py 2>&1 | grep -v amdgpu.ids | grep 'count-only, LDS\|43 slots' | tr '\n' ' ')"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
