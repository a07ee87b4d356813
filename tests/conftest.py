import os
import sys
import json

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """Lazy view on one tests/golden/<name>.npz fixture."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
        self.hp = json.loads(str(self.z['base_hparams']))
        # the fixtures were generated through the restated pure-Python fastdtw (predecessor rule 0): a model that is compared
        # with g7 / g11 states that rule; the product's default is config.DTW_TIE_ORDER (tests/golden/ties.npz pins 1 and 2)
        self.hp.setdefault('dtw_tie_order', 0)
        self.seed = int(self.z['seed'])
        self.has_ego = bool(self.z['has_ego'])

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    @property
    def files(self):
        return self.z.files

    def ragged(self, k, fill):
        return [[int(v) for v in row if v != fill] for row in self.z[k]]


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(params=['tiny', 'tiny_ego', 'density'])
def golden(request):
    return load_golden(request.param)


@pytest.fixture
def tiny():
    return load_golden('tiny')
