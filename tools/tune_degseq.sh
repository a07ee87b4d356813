#!/bin/bash
# rebuild libsubgnn_hip.so with different tuning macros on the GPU box and time the CSR gather
for flags in "-DDS_SEARCH=1024" "-DDS_SEARCH=512" "-DDS_SEARCH=256" "-DDS_SEARCH=2048"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/degseq_probe.py 20 search 2>&1 | tail -1)"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
