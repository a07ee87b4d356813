"""Constants of the hot path (mirrors the reference's top-level config.py:6,9)."""
import os
from pathlib import Path

# directory every dataset path is relative to (reference config.py:6); override with the
# SUBGNN_PROJECT_ROOT environment variable or by assigning subgnn_amd.config.PROJECT_ROOT.
PROJECT_ROOT = Path(os.environ.get('SUBGNN_PROJECT_ROOT', '.'))

# node ids are 1-based so that 0 can pad (reference config.py:9, SubGNN/SubGNN.py:554-559)
PAD_VALUE = 0

# fastdtw's predecessor rule on ties (oracle/fastdtw_restate.py describes the three rules the kernels implement).  The
# reference pins fastdtw==0.3.4 (SubGNN.yml:109), which ships a pure-Python module AND a compiled one that takes precedence when
# it was built.  SubGNN.py:808-815 hands gamma.calc_dtw the EMPTY degree sequence of every padded component row; the pure-Python
# ``__dtw`` walks its path back from D[0, len_y] -- a ``(inf,)`` default entry -- and raises IndexError on ``[1]``.  Every
# dataset with a subgraph of fewer components than the widest one (all of the reference's) therefore ran the COMPILED variant:
# the default is the rule of the shape a compiled loop has (2: the three predecessor costs compared with <=, diagonal first),
# not the rule of the module that provably did not produce the reference's numbers (0).  The compiled source is absent from the
# reference tree and from this image, so the choice stays a hyper-parameter (hparams['dtw_tie_order'] in {0, 1, 2}) and the
# DTW values stay "parity unpinned" (DESIGN.md section 5).
DTW_TIE_ORDER = 2
