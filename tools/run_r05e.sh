export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r05e_gputests.log 2>&1; tail -3 gpurun_out/r05e_gputests.log
for c in ppi_bp hpo_metab em_user density_n; do python tools/bench_standin.py --config $c > gpurun_out/r05e_standin_$c.json 2> gpurun_out/r05e_standin_$c.err; python -c "
import json; d=json.load(open('gpurun_out/r05e_standin_$c.json')); print('$c', round(d['ms_per_step'],3), round(d['eager']['ms_per_step'],3), d['kernels_per_step'])"; done
python tools/step_kernels.py --config hpo_metab --out gpurun_out/r05e_step_kernels_hpo_metab.txt > /dev/null 2>&1; tail -1 gpurun_out/r05e_step_kernels_hpo_metab.txt
