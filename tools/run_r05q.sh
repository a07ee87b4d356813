export TMPDIR=/tmp
for c in ppi_bp hpo_metab em_user density_n; do python tools/bench_standin.py --config $c --atomics 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$c atomics', round(d['ms_per_step'],3), round(d['eager']['ms_per_step'],3), d['kernels_per_step'])"; done
