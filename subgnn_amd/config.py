"""Constants of the hot path (mirrors the reference's top-level config.py:6,9)."""
import os
from pathlib import Path

# directory every dataset path is relative to (reference config.py:6); override with the
# SUBGNN_PROJECT_ROOT environment variable or by assigning subgnn_amd.config.PROJECT_ROOT.
PROJECT_ROOT = Path(os.environ.get('SUBGNN_PROJECT_ROOT', '.'))

# node ids are 1-based so that 0 can pad (reference config.py:9, SubGNN/SubGNN.py:554-559)
PAD_VALUE = 0
