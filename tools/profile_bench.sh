#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run; per-stage breakdown of one pass.
# (sequential schedule on one stream: the breakdown cuts a pass at kernel names, which needs the passes' kernels in order)
# usage: tools/profile_bench.sh TAG   -> gpurun_out/TAG_kernel_stats.csv, TAG_pass_breakdown.txt, TAG_bench_profiled.json
export TMPDIR=/tmp
T=$1
D=gpurun_out/${T}_prof
rm -rf $D
SGNN_OVERLAP_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline > gpurun_out/${T}_bench_profiled.json 2> gpurun_out/${T}_prof.err
KT=$(find $D -name "*kernel_trace.csv" | head -1)
KS=$(find $D -name "*kernel_stats.csv" | head -1)
cp $KS gpurun_out/${T}_kernel_stats.csv
python tools/pass_breakdown.py $KT 60 14 > gpurun_out/${T}_pass_breakdown.txt
rm -rf $D
cat gpurun_out/${T}_pass_breakdown.txt
