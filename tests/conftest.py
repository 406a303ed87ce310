import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def manifest():
    import json
    with open(os.path.join(GOLD, 'manifest.json')) as f:
        return json.load(f)


def load_gold(name):
    import numpy as np
    return np.load(os.path.join(GOLD, name + '.npz'))
