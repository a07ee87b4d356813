#!/bin/bash
# rebuild libsubgnn_hip.so with different tuning macros on the GPU box and time the CSR gather
for flags in "-DDS_GRID_CAP=1048576" "-DDS_GRID_CAP=16384" "-DDS_GRID_CAP=8192" "-DDS_GRID_CAP=5120" "-DDS_GRID_CAP=25000"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/degseq_probe.py 20 ordered 2>&1 | tail -1)"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
