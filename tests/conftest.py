"""pytest configuration: the `gpu` marker and shared helpers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
  with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as z:
    return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def golden():
  return load_golden
