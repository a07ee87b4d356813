#!/bin/bash
# Functional check of bench.py's multi-rank path on ONE GPU: 2 ranks share the device, collectives over
# gloo (host memory).  Checks that both scaling modes run with both forms of the head (--head sharded | replicated), that the loss is finite and that the sharded
# pass reports the global subgraph count.  Not a measurement.
export SGNN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for head in sharded replicated; do
for mode in weak strong; do
  tag=${mode}_${head}
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
    bench.py --gpus 2 --steps 2 --warmup 1 --nodes 200000 --subgraphs 4000 --scaling $mode --head $head --pipeline-multi --no-cpu-baseline --no-extras 2> gpurun_out/${TAG:-r05}_multirank_$tag.err | tail -1 > gpurun_out/${TAG:-r05}_multirank_$tag.json
  python - <<PY
import json
d = json.load(open('gpurun_out/${TAG:-r05}_multirank_$tag.json'))
print('$tag', d['n_gpus'], d['scaling'], d['config']['subgraphs_total'], d['ms_per_step'], d['loss'], d['stages_ms'], d['collectives'])
PY
done
done
# the same two modes on one rank under the launcher (RCCL, world size 1)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
  bench.py --gpus 1 --steps 2 --warmup 1 --nodes 200000 --subgraphs 4000 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('nccl world 1', d['ms_per_step'], d['loss'])"

# gradient exchange, exactly: without dropout the loss AFTER an update (step 2) of the strong 2-rank runs must equal the
# single-rank loss -- the same 4000 subgraphs, the same draws (global tape items), gradients averaged / summed over ranks
export SGNN_BENCH_HP='{"lin_dropout": 0.0}'
for head in sharded replicated; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
    bench.py --gpus 2 --steps 3 --warmup 0 --nodes 200000 --subgraphs 4000 --scaling strong --head $head --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('no dropout, strong, $head: loss after 5 updates', d['loss'])"
done
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 0 --nodes 200000 --subgraphs 4000 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('no dropout, single rank:             loss after 5 updates', d['loss'])"
