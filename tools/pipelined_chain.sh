export TMPDIR=/tmp
D=gpurun_out/pc_prof
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pc_bench.json 2> gpurun_out/pc.err
KT=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/pipelined_chain.py $KT > gpurun_out/pc_chain.txt
rm -rf $D
tail -12 gpurun_out/pc_chain.txt
