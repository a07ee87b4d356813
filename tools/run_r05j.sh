export TMPDIR=/tmp
python -m pytest tests/test_gpu_integer.py tests/test_gpu_fullsize.py tests/test_gpu_hotpath.py -m gpu -x -q -k "dtw or DTW or similar or fullsize or structure or sparse_prepare or tie" 2>&1 | tail -4
python tools/dtw_side_probe.py external 5 2>/dev/null | tail -1
python tools/dtw_side_probe.py internal 5 2>/dev/null | tail -1
python tools/dtw_full_check.py 2>/dev/null | tail -3
