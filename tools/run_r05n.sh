export TMPDIR=/tmp
rebuild() { SGNN_HIPCC_FLAGS="$1" python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'degree_sequence.hip')); build.build(verbose=False)" > /dev/null 2>&1; }
for k in 2 4 8 16; do rebuild "-DDS_KARY=$k"; echo "KARY $k"; python tools/degseq_hbm_probe.py --family bfs --benchmark-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  shipped ms', round(d['shipped_search']['ms_per_launch'],4), 'streaming', round(d['streaming']['ms_per_launch'],4), 'agree', d['forms_agree'])"; done
rebuild ""
python -m pytest tests/test_gpu_integer.py tests/test_gpu_fullsize.py -m gpu -x -q -k "degree or degseq or fullsize" 2>&1 | tail -2
