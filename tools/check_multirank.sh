#!/bin/bash
# Functional check of bench.py's multi-rank path on ONE GPU: 2 ranks share the device, collectives over
# gloo (host memory).  Checks that both scaling modes run, that the loss is finite and that the sharded
# pass reports the global subgraph count.  Not a measurement.
export SGNN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in weak strong; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
    bench.py --gpus 2 --steps 2 --warmup 1 --nodes 200000 --subgraphs 4000 --scaling $mode --no-cpu-baseline 2> gpurun_out/r02_multirank_$mode.err | tail -1 > gpurun_out/r02_multirank_$mode.json
  python - <<PY
import json
d = json.load(open('gpurun_out/r02_multirank_$mode.json'))
print('$mode', d['n_gpus'], d['scaling'], d['config']['subgraphs_total'], d['ms_per_step'], d['loss'], d['stages_ms'])
PY
done
# the same two modes on one rank under the launcher (RCCL, world size 1)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
  bench.py --gpus 1 --steps 2 --warmup 1 --nodes 200000 --subgraphs 4000 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nccl world 1', d['ms_per_step'], d['loss'])"
