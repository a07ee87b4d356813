export TMPDIR=/tmp
rebuild() { SGNN_HIPCC_FLAGS="$1" python -c "
import os
from subgnn_amd import build
os.utime(os.path.join(build.CSRC,'lstm.hip')); build.build(verbose=False)" > /dev/null 2>&1; }
for f in "-DLSTM_ACC=1" "-DLSTM_ACC=2" ""; do rebuild "$f"; echo "flags [$f]"; python tools/step_kernels.py --config ppi_bp 2>/dev/null | grep "lstm_fwd\|lstm_bwd\|^kernels" | cut -c1-60; done
python -m pytest tests/test_gpu_float.py -m gpu -x -q -k "lstm or LSTM" 2>&1 | tail -2
