export TMPDIR=/tmp
python -m pytest tests/test_gpu_integer.py tests/test_gpu_fullsize.py tests/test_gpu_hotpath.py tests/test_gpu_configs.py -m gpu -x -q -k "bfs or sparse or position or prepare" 2>&1 | tail -3
python tools/bfs_probe.py 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05i_bfs_prof -- python3 tools/bfs_probe.py --reps 2 > gpurun_out/r05i_bfs_prof.log 2>&1
