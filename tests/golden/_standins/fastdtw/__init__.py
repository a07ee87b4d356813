"""Forwards to the repo's restatement (the real fastdtw==0.3.4 is absent: provisional).  The predecessor rule is the restated
pure-Python module's (0) unless SGNN_STANDIN_FASTDTW_TIE names another one (tests/golden/make_goldens_ties.py)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..', '..')))
from oracle import fastdtw_restate as _fd  # noqa: E402


def fastdtw(x, y, radius=1, dist=None):
    return _fd.fastdtw(x, y, radius=radius, dist=dist, tie_order=int(os.environ.get('SGNN_STANDIN_FASTDTW_TIE', '0')))
