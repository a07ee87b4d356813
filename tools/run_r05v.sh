# the round's measured set again after the last changes (everything of tools/run_round_profiles.sh + step lists + the 6250-shard
# trace; the counter passes, the traffic probe and the DTW budget are not repeated: their kernels did not change)
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests.log 2>&1; tail -2 gpurun_out/r05_gputests.log
bash tools/run_round_profiles.sh r05 > gpurun_out/r05_round_profiles.log 2>&1
for c in density_n ppi_bp hpo_metab; do python tools/step_kernels.py --config $c --out gpurun_out/r05_step_kernels_$c.txt > /dev/null 2>&1; tail -1 gpurun_out/r05_step_kernels_$c.txt; done
bash tools/run_r05u.sh > /dev/null 2>&1
python tools/khop1_time.py > gpurun_out/r05_khop1_time.txt 2>/dev/null
python tools/bfs_probe.py > gpurun_out/r05_bfs_probe.json 2>/dev/null
ls gpurun_out | grep "^r05_" | wc -l
