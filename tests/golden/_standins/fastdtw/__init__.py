"""Forwards to the repo's restatement (the real fastdtw==0.3.4 is absent: provisional)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..', '..')))
from oracle.fastdtw_restate import fastdtw  # noqa: E402,F401
