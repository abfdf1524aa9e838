"""CPU pins of the wavetable method behind harm_table_kernel: the committed coefficient header is what the
generator produces, its aliasing error is what DESIGN.md states, and the numpy model of the kernel (same
index arithmetic, fp32) reproduces the oracle's Harmonic within a small multiple of fp32 round-off."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ddsp_oracle as O
from wavetable_model import COEFFS, harmonic_table_model, load_coeffs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
  spec = importlib.util.spec_from_file_location('gen_wavetable_coeffs', os.path.join(ROOT, 'tools', 'gen_wavetable_coeffs.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_committed_header_is_current():
  assert open(COEFFS).read() == _generator().render()


@pytest.mark.parametrize('w,k,bound', [(6, 60, 4e-6), (6, 100, 7e-6), (8, 100, 1e-7), (8, 128, 7e-6)])
def test_window_aliasing_error(w, k, bound):
  g = _generator()
  c = load_coeffs()
  de, do = c['kWtDegE%d' % w], c['kWtDegO%d' % w]
  ce = c['kWtE%d' % w].reshape(w // 2, de + 1)
  co = c['kWtO%d' % w].reshape(w // 2, do + 1)
  assert g.aliasing_error(w, ce, co, k, 512) <= bound
  # the tabulated 1 / psi_hat is the transform of exactly this window
  ks = np.arange(0, 129)
  np.testing.assert_allclose(c['kWtInvPsi%d_T512' % w], 1.0 / g.psi_hat(w, ce, co, ks / 512.0), rtol=2e-7)
  np.testing.assert_allclose(c['kWtInvPsi%d_T512' % w] * c['kWtPsi%d_T512' % w], 1.0, rtol=3e-7)


@pytest.mark.parametrize('f0_center,spread', [(70.0, 1.0), (200.0, 1.0), (400.0, 150.0), (3000.0, 2000.0)])
@pytest.mark.parametrize('w,k', [(6, 100), (8, 128), (6, 60)])
def test_model_matches_oracle(f0_center, spread, w, k):
  rng = np.random.default_rng(int(f0_center) + k)
  b, f, n, sr = 2, 40, 2560, 16000
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = np.abs(f0_center + spread * rng.standard_normal((b, f, 1))).astype(np.float32)
  truth = O.harmonic(amps, hd, f0, n, sr, dtype=np.float64)
  out = harmonic_table_model(amps, hd, f0, n, sr, W=w)
  assert np.abs(out - truth).max() <= 1e-5 * 2.0


def test_model_linear_envelope_and_long_hop():
  rng = np.random.default_rng(5)
  b, f, k, hop, sr = 1, 12, 100, 192, 16000
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = (75 + 5 * rng.standard_normal((b, f, 1))).astype(np.float32)
  truth = O.harmonic(amps, hd, f0, f * hop, sr, amp_resample_method='linear', dtype=np.float64)
  out = harmonic_table_model(amps, hd, f0, f * hop, sr, W=6, amp_linear=True)
  assert np.abs(out - truth).max() <= 1e-5 * 2.0
