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


def pytest_collection_modifyitems(config, items):
  """A module may list node-id fragments in SLOW_UNDER_EMULATION: those cases are skipped there
  (tests/test_simt_emulated.py re-runs the GPU tests on the CPU emulation and leaves the minutes-long ones out)."""
  for item in items:
    slow = getattr(getattr(item, 'module', None), 'SLOW_UNDER_EMULATION', ())
    if slow and any(fragment in item.nodeid for fragment in slow):
      item.add_marker(pytest.mark.skip(reason='left to the GPU run: minutes under the CPU emulation'))


def load_golden(name):
  with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as z:
    return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def golden():
  return load_golden
