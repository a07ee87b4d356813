export TMPDIR=/tmp
python -m pytest tests/test_gpu_float.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_graph_step.py tests/test_gpu_train_driver.py -m gpu -x -q > gpurun_out/r05f_gputests.log 2>&1; tail -3 gpurun_out/r05f_gputests.log
for c in ppi_bp hpo_metab em_user density_n; do python tools/bench_standin.py --config $c > gpurun_out/r05f_standin_$c.json 2> gpurun_out/r05f_standin_$c.err; python -c "
import json; d=json.load(open('gpurun_out/r05f_standin_$c.json')); print('$c', round(d['ms_per_step'],3), round(d['eager']['ms_per_step'],3), d['kernels_per_step'])"; done
python tools/cold_pass_probe.py > gpurun_out/r05f_cold_pass.txt 2>&1; grep -v "^/opt" gpurun_out/r05f_cold_pass.txt | head -40
