export TMPDIR=/tmp
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  d=gpurun_out/r05m_bfs_pmc_$(echo $c | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bfs_probe.py --reps 1 > $d.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r05m_bfs_pmc_*/')):
    f = glob.glob(d + '*/*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    rows = list(csv.DictReader(open(f[0])))
    # dispatches of msbfs_level_kernel in order; group by counter
    per = collections.defaultdict(list)
    for r in rows:
        if 'msbfs_level' in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        # one search = (levels+3) level launches; print the first hinted search's levels: take a window of 10 after the first 32
        print(d.split('/')[-2], k, [round(x / 1e3, 1) for x in v[32:42]])
PY
