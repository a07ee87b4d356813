#!/bin/bash
# rebuild libsubgnn_hip.so with different tuning macros on the GPU box and time the CSR gather
for flags in "-DDS_PIPELINE=0" "-DDS_PIPELINE=1" "-DDS_PIPELINE=1 -DDS_BIG=128" "-DDS_PIPELINE=1 -DDS_BIG=32" "-DDS_PIPELINE=1 -DDS_WAVES=2"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/degseq_probe.py 20 2>&1 | tail -1)"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
