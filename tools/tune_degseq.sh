#!/bin/bash
# rebuild libsubgnn_hip.so with different tuning macros on the GPU box and time the CSR gather
for flags in "-DDS_MIN_WAVES=5" "-DDS_MIN_WAVES=6" "-DDS_MIN_WAVES=7" "-DDS_MIN_WAVES=8" "-DDS_MIN_WAVES=4"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/degseq_probe.py 20 ordered 2>&1 | tail -1)"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
