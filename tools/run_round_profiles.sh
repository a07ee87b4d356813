#!/bin/bash
# Everything profiles/ holds for a round, in one go on the GPU box.  usage: tools/run_round_profiles.sh r02
T=$1
O=gpurun_out
python bench.py > $O/${T}_final_bench_default.json 2> $O/${T}_final_bench_default.err
# the sequential schedule on one stream (no pass pipelining, no two-stream preparation): stage times add up to the step
SGNN_OVERLAP_STREAMS=0 python bench.py --no-pipeline --no-cpu-baseline --no-extras --steps 10 > $O/${T}_final_bench_sequential.json 2>/dev/null
python bench.py --no-pipeline --no-cpu-baseline --no-extras --steps 10 > $O/${T}_final_bench_two_streams.json 2>/dev/null
python bench.py --graph train --no-cpu-baseline --no-extras --steps 10 > $O/${T}_final_bench_graph.json 2>/dev/null
python bench.py --graph both --no-cpu-baseline --no-extras --steps 10 > $O/${T}_final_bench_graph_both.json 2>/dev/null
bash tools/profile_bench.sh ${T}_final > /dev/null 2>&1
bash tools/pass_kernels.sh > /dev/null 2>&1; python tools/pass_train_half.py gpurun_out/pk_pass_kernels.txt > $O/${T}_training_half_kernels.txt 2>/dev/null; cp gpurun_out/pk_pass_kernels.txt $O/${T}_pass_kernels.txt
python tools/bench_repeat.py 8 --steps 12 --warmup 0 > $O/${T}_bench_repeat.txt 2>&1
python bench.py --subgraphs 6250 --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $O/${T}_bench_shard6250.json 2>/dev/null
python bench.py --subgraphs 6250 --no-cpu-baseline --no-extras --steps 20 --warmup 3 --graph off > $O/${T}_bench_shard6250_eager.json 2>/dev/null
python bench.py --subgraphs 6250 --no-cpu-baseline --no-extras --steps 20 --warmup 3 --graph both > $O/${T}_bench_shard6250_graph_both.json 2>/dev/null
for c in density_n ppi_bp hpo_metab em_user; do
  python tools/bench_standin.py --config $c --epochs 9 > $O/${T}_bench_standin_$c.json 2> $O/${T}_bench_standin_$c.err
done
bash tools/check_multirank.sh > $O/${T}_multirank_check.txt 2>&1
ls -la $O | grep ${T}_ | head -40
