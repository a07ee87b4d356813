#!/bin/bash
# Out-of-Infinity-Cache datapoint of the CSR gather: timing runs, then the rocprofv3 counter passes
# (separate passes per counter group, kernel-trace-free), then per-kernel counter averages.
# usage: tools/run_hbm_probe.sh r04 ; python tools/make_traffic_profile.py gpurun_out r04 > profiles/r04_degseq_traffic.json
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r04}
for fam in bfs random bench; do
  extra=""; f=$fam
  if [ $fam = bench ]; then extra="--benchmark-graph"; f=bfs; fi
  python tools/degseq_hbm_probe.py --family $f $extra > $O/${T}_hbm_$fam.json 2> $O/${T}_hbm_$fam.err
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/${T}_pmc_${fam}_$(echo $c | tr ' ' '_')
    rm -rf $d
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/degseq_hbm_probe.py --family $f $extra --reps 3 > $d.log 2>&1
  done
  python tools/pmc_summary.py $O/${T}_pmc_${fam}_FETCH_SIZE $O/${T}_pmc_${fam}_WRITE_SIZE $O/${T}_pmc_${fam}_TCC_HIT_sum_TCC_MISS_sum > $O/${T}_pmc_$fam.json
done
cat $O/${T}_hbm_bfs.json $O/${T}_hbm_random.json $O/${T}_hbm_bench.json
