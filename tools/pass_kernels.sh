export TMPDIR=/tmp
D=gpurun_out/pk_prof
rm -rf $D
SGNN_OVERLAP_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline > gpurun_out/pk_bench.json 2> gpurun_out/pk.err
KT=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/pass_kernels.py $KT > gpurun_out/pk_pass_kernels.txt
rm -rf $D
wc -l gpurun_out/pk_pass_kernels.txt
