"""The portable half of the parity pin: regenerate every fixture of make_golden.py under REAL TensorFlow.

    pip install 'tensorflow<=2.11' numpy            # the reference's own pin (setup.py:56); CPU wheels are enough
    DDSP_REFERENCE_ROOT=/path/to/magenta-ddsp python tests/golden/make_golden_tf.py
    python -m pytest tests/test_golden_tf_pin.py -q

writes tests/golden/NAME.tf.npz beside each committed NAME.npz - the same cases, the same seeds, the same reference source files
(ddsp/core.py, synths.py, processors.py, effects.py, losses.py, spectral_ops.py, imported UNMODIFIED from DDSP_REFERENCE_ROOT), but
tf.signal / tf.image / tf.cumsum are TensorFlow's, not the numpy stand-in of tf_numpy_shim.py.  tests/test_golden_tf_pin.py then
diffs the two sets: inputs bit for bit, outputs within the fp32 tolerances the GPU tests hold the kernels to.

Why this exists ("missing" #1): TensorFlow cannot be installed in the build container (no network), so the committed
fixtures come from the reference's source run on a stand-in whose op semantics are this repository's reading of TensorFlow
(SURVEY.md Appendix A).  A misreading shared by the stand-in and the oracle is invisible to every test here - one was found by
inspection in round 6 (tf.signal.hann_window of an ODD length is the symmetric window: window_ops.py, `n = window_length +
periodic * even - 1`), another by fuzzing in round 5 (legacy NEAREST resize rounds with roundf).  Whoever has TensorFlow closes
the loop with the three commands above; the fixtures it writes are not committed by default (`*.tf.npz` may be, once made).

Only TensorFlow itself and numpy are needed: `import ddsp` is avoided (its __init__ pulls in crepe, librosa, tensorflow_probability,
gin and the training stack); the six modules are imported from a namespace package, and gin / crepe / librosa / tfp are stubbed
exactly as far as the synthesis path touches them (decorators that return their argument; librosa's two published formulas) when
they are not installed.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))


def _identity_decorator(*args, **kwargs):
  if len(args) == 1 and callable(args[0]) and not kwargs:
    return args[0]
  return lambda f: f


def install(reference_root):
  import tensorflow as tf                                  # the real one, or an ImportError that says what is missing
  major, minor = (int(v) for v in tf.__version__.split('.')[:2])
  if (major, minor) > (2, 11):
    print('warning: the reference pins tensorflow<=2.11 (setup.py:56); this is', tf.__version__, file=sys.stderr)
  try:
    import gin                                             # noqa: F401  (the real gin registers the configurables: harmless)
  except ImportError:
    gin = types.ModuleType('gin')
    gin.register = _identity_decorator
    gin.configurable = _identity_decorator
    sys.modules['gin'] = gin
  sys.path.insert(0, HERE)
  import tf_numpy_shim                                     # (only for librosa's two formulas, restated there)
  for name in ('crepe', 'tensorflow_probability', 'librosa'):
    try:
      __import__(name)
    except ImportError:
      stub = types.ModuleType(name)
      stub.distributions = types.SimpleNamespace(HiddenMarkovModel=object)
      if name == 'librosa':
        stub.fft_frequencies = tf_numpy_shim.librosa_fft_frequencies
        stub.A_weighting = tf_numpy_shim.librosa_a_weighting
      sys.modules[name] = stub
  pkg = types.ModuleType('ddsp')
  pkg.__path__ = [os.path.join(reference_root, 'ddsp')]    # namespace only: ddsp/__init__.py (the training stack) is not run
  sys.modules['ddsp'] = pkg
  return tf


if __name__ == '__main__':
  os.environ['DDSP_GOLDEN_BACKEND'] = 'tf'
  os.environ.setdefault('DDSP_REFERENCE_ROOT', '/root/reference')
  sys.path.insert(0, HERE)
  import make_golden
  make_golden.main()
