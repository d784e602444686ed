import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


_OTHER_KINDS = ('loss_', 'infer_', 'init_', 'dense_', 'chunk_')


def golden_names():
    """model fixtures (one rolling / static train-pattern sequence each)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and not f.startswith(_OTHER_KINDS))


def infer_golden_names():
    """inference-loop fixtures (reference infer.py:48-87: update_graph(mode='test') + decode_tracks)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and f.startswith('infer_'))


def loss_golden_names():
    """targets / loss fixtures (reference models/loss.py)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and f.startswith('loss_'))


def chunk_golden_names():
    """whole training chunks (reference train.py:54-135 with models/loss.py's targets and losses)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and f.startswith('chunk_'))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN_DIR
