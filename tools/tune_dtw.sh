#!/bin/bash
for flags in "-DDTW_BLOCKS=256" "-DDTW_BLOCKS=512" "-DDTW_BLOCKS=1024" "-DDTW_BLOCKS=2048"; do
  SGNN_HIPCC_FLAGS="$flags" python -m subgnn_amd.build --force > /dev/null 2>&1
  echo "$flags: $(python tools/dtw_probe.py 3 | tr '\n' ' ')"
done
python -m subgnn_amd.build --force > /dev/null 2>&1
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_dtw3 -- python3 tools/dtw_probe.py 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_dtw4 -- python3 tools/dtw_probe.py 1 > /dev/null 2>&1
