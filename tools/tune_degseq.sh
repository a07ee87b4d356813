#!/bin/bash
# rebuild libsubgnn_hip.so with different tuning macros on the GPU box and time the CSR gather
for flags in "-DDS_INFLIGHT=8" "-DDS_INFLIGHT=16" "-DDS_INFLIGHT=32" "-DDS_INFLIGHT=4"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/degseq_probe.py 20 ordered 2>&1 | tail -1) $(python tools/degseq_probe.py 5 heavy 2>&1 | tail -1)"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
