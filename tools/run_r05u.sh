# one sequential pass at the strong-scaling shard size, kernel by kernel (where the 6250-shard's preparation spends its time)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
D=gpurun_out/pk_prof
rm -rf $D
SGNN_OVERLAP_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --subgraphs 6250 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline --graph off > gpurun_out/pk6250_bench.json 2> gpurun_out/pk6250.err
KT=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/pass_kernels.py $KT > gpurun_out/r05_pass_kernels_6250.txt
rm -rf $D
wc -l gpurun_out/r05_pass_kernels_6250.txt
