export TMPDIR=/tmp
python -m pytest tests/test_gpu_float.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -x -q -k "lstm or LSTM or training_step or g11 or forward" 2>&1 | tail -3
for c in ppi_bp hpo_metab; do python tools/bench_standin.py --config $c 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$c', round(d['ms_per_step'],3), round(d['eager']['ms_per_step'],3), d['kernels_per_step'], 'atomics', d['atomics'] and (round(d['atomics']['ms_per_step'],3), d['atomics']['kernels_per_step']))"; done
