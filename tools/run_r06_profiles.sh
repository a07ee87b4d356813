#!/bin/bash
# Everything profiles/r06_* holds, in one go on the GPU box:   bash tools/run_r06_profiles.sh
export TMPDIR=/tmp
export TAG=r06
O=gpurun_out
mkdir -p $O
bash tools/run_round_profiles.sh r06 > $O/r06_round_profiles.log 2>&1
for c in density_n ppi_bp hpo_metab em_user; do
  python tools/step_kernels.py --config $c --out $O/r06_step_kernels_$c.txt > /dev/null 2>&1
done
python tools/train_half_probe.py 50000 > $O/r06_training_half_probe_50k.json 2>/dev/null
python tools/train_half_probe.py 6250 > $O/r06_training_half_probe_6250.json 2>/dev/null
python tools/warm_probe.py > $O/r06_warm_up_split.txt 2>&1
bash tools/run_dtw_pmc.sh r06 > /dev/null 2>&1
python tools/khop1_time.py 20 2>&1 | grep -v amdgpu.ids > $O/r06_khop1_time.txt
DTW_PROBE_DUPS=1 python tools/dtw_side_probe.py external 5 2>&1 | grep -v amdgpu.ids > $O/r06_dtw_side_probe.txt
DTW_PROBE_DUPS=1 python tools/dtw_side_probe.py internal 5 2>&1 | grep -v amdgpu.ids >> $O/r06_dtw_side_probe.txt
python tools/dtw_full_check.py 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/r06_dtw_side_probe.txt
python tools/bfs_probe.py > $O/r06_bfs_probe.json 2>/dev/null
bash tools/run_hbm_probe.sh r06 > /dev/null 2>&1; python tools/make_traffic_profile.py $O r06 > $O/r06_degseq_traffic.json
python tools/degseq_probe.py 20 search 2>&1 | grep -v amdgpu.ids > $O/r06_degseq_probe.txt
python tools/epoch_stall_probe.py ppi_bp 12 2>&1 | grep -v amdgpu.ids > $O/r06_epoch_stall_probe.txt
python tools/device_stall_probe.py 4 2>&1 | grep -v amdgpu.ids > $O/r06_device_stall_probe.txt
python tools/device_stall_probe.py 4 2>&1 | grep -v amdgpu.ids >> $O/r06_device_stall_probe.txt
python -m pytest tests -m gpu -q > $O/r06_gpu_tests.log 2>&1
tail -3 $O/r06_gpu_tests.log
ls $O | grep "^r06_" | wc -l
