export TMPDIR=/tmp
python -m pytest tests/test_gpu_integer.py tests/test_gpu_fullsize.py -m gpu -x -q -k "bfs" 2>&1 | tail -3
python tools/bfs_probe.py 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05d_bfs_prof -- python3 tools/bfs_probe.py --reps 2 > gpurun_out/r05d_bfs_prof.log 2>&1
for c in ppi_bp hpo_metab; do python tools/step_kernels.py --config $c --out gpurun_out/r05d_step_kernels_$c.txt > /dev/null 2> gpurun_out/r05d_step_kernels_$c.err; tail -1 gpurun_out/r05d_step_kernels_$c.txt; done
