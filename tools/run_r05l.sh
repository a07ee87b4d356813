export TMPDIR=/tmp
rebuild() { SGNN_HIPCC_FLAGS="$1" python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'graph_sets.hip')); build.build(verbose=False)" > /dev/null 2>&1; }
python -m pytest tests/test_gpu_integer.py tests/test_gpu_fullsize.py -m gpu -x -q -k "khop or border or anchor or sampled or fullsize" 2>&1 | tail -3
for f in "-DK1_CHUNK_DEAL=0" "-DK1_DEAL=8" "-DK1_DEAL=16" ""; do rebuild "$f"; echo "flags [$f]"; python tools/khop1_time.py 2>/dev/null | tail -1; done
