export TMPDIR=/tmp
export TAG=r05
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests.log 2>&1; tail -2 gpurun_out/r05_gputests.log
bash tools/run_round_profiles.sh r05 > gpurun_out/r05_round_profiles.log 2>&1
bash tools/run_dtw_pmc.sh r05 > /dev/null 2>&1
bash tools/run_hbm_probe.sh r05 > gpurun_out/r05_hbm.log 2>&1
python tools/make_traffic_profile.py gpurun_out r05 > gpurun_out/r05_degseq_traffic.json 2> gpurun_out/r05_traffic.err
python tools/khop1_time.py > gpurun_out/r05_khop1_time.txt 2>/dev/null
python tools/bfs_probe.py > gpurun_out/r05_bfs_probe.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_bfs_prof -- python3 tools/bfs_probe.py --reps 2 > gpurun_out/r05_bfs_prof.log 2>&1
SGNN_HIPCC_FLAGS=-DDTW_PROBE_COUNT python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1
python tools/dtw_budget.py external > gpurun_out/r05_dtw_budget_external.json 2>/dev/null
python tools/dtw_budget.py internal > gpurun_out/r05_dtw_budget_internal.json 2>/dev/null
python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'dtw.hip')); build.build(verbose=False)" > /dev/null 2>&1
for c in density_n ppi_bp hpo_metab; do python tools/step_kernels.py --config $c --out gpurun_out/r05_step_kernels_$c.txt > /dev/null 2>&1; tail -1 gpurun_out/r05_step_kernels_$c.txt; done
bash tools/run_r05u.sh > /dev/null 2>&1
ls gpurun_out | grep "^r05_" | wc -l
