import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_names():
    """model fixtures (one rolling / static sequence each)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and not f.startswith('loss_'))


def loss_golden_names():
    """targets / loss fixtures (reference models/loss.py)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and f.startswith('loss_'))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN_DIR
