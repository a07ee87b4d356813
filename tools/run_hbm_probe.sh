#!/bin/bash
# Out-of-Infinity-Cache datapoint of the CSR gather: timing runs, then the rocprofv3 counter passes
# (separate passes per counter group, kernel-trace-free), then per-kernel counter averages.
export TMPDIR=/tmp
O=gpurun_out
for fam in bfs random bench; do
  extra=""; f=$fam
  if [ $fam = bench ]; then extra="--benchmark-graph"; f=bfs; fi
  python tools/degseq_hbm_probe.py --family $f $extra > $O/r02_hbm_$fam.json 2> $O/r02_hbm_$fam.err
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/r02_pmc_${fam}_$(echo $c | tr ' ' '_')
    rm -rf $d
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/degseq_hbm_probe.py --family $f $extra --reps 3 > $d.log 2>&1
  done
  python tools/pmc_summary.py $O/r02_pmc_${fam}_FETCH_SIZE $O/r02_pmc_${fam}_WRITE_SIZE $O/r02_pmc_${fam}_TCC_HIT_sum_TCC_MISS_sum > $O/r02_pmc_$fam.json
done
cat $O/r02_hbm_bfs.json $O/r02_hbm_random.json $O/r02_hbm_bench.json
