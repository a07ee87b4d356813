export TMPDIR=/tmp
python -m pytest tests/test_gpu_configs.py tests/test_gpu_integer.py -m gpu -x -q 2>&1 | tail -2
python tools/bench_standin.py --config em_user 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('em_user', round(d['ms_per_step'],3), round(d['eager']['ms_per_step'],3), d['kernels_per_step'], d['loss'], d['loss_graph'])"
python tools/dtw_side_probe.py external 5 2>/dev/null | tail -1
