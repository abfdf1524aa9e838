"""CPU tests of the oracle: golden vectors made by the reference's own source files,
the reference's known-answer tests re-expressed, and the closed
forms the HIP kernels rely on."""
import numpy as np
import pytest
import scipy.signal

from oracle import ddsp_oracle as O
from conftest import load_golden

HARMONIC_CASES = ['harmonic_window_cumsum', 'harmonic_window_angular',
                  'harmonic_linear_cumsum', 'harmonic_k100_live',
                  'harmonic_hop192_angular', 'harmonic_noscale_nonorm']
NOISE_CASES = ['noise_m65_w257', 'noise_m65_w0', 'noise_m33_w17', 'noise_m17_w16_even',
               'noise_ragged']


# ---- golden vectors (reference source on the numpy TF stand-in) -----------------
@pytest.mark.parametrize('name', HARMONIC_CASES)
def test_harmonic_matches_reference_source(name):
  g = load_golden(name)
  scale_fn = O.exp_sigmoid if int(g['scale']) else None
  c = O.harmonic_get_controls(g['amplitudes'], g['harmonic_distribution'], g['f0_hz'],
                              int(g['sample_rate']), scale_fn, bool(g['normalize']))
  np.testing.assert_allclose(c['amplitudes'], g['ctl_amplitudes'], rtol=1e-6, atol=1e-9)
  np.testing.assert_allclose(c['harmonic_distribution'], g['ctl_harmonic_distribution'],
                             rtol=1e-6, atol=1e-9)
  sig = O.harmonic_get_signal(c['amplitudes'], c['harmonic_distribution'], c['f0_hz'],
                              int(g['n_samples']), int(g['sample_rate']),
                              str(g['amp_method']), bool(g['angular']))
  assert sig.dtype == np.float32 and sig.shape == g['signal'].shape
  np.testing.assert_allclose(sig, g['signal'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('name', NOISE_CASES)
def test_filtered_noise_matches_reference_source(name):
  g = load_golden(name)
  scale_fn = O.exp_sigmoid if int(g['scale']) else None
  c = O.filtered_noise_get_controls(g['magnitudes'], scale_fn)
  np.testing.assert_allclose(c['magnitudes'], g['ctl_magnitudes'], rtol=1e-6, atol=1e-9)
  ir = O.frequency_impulse_response(c['magnitudes'], int(g['window_size']))
  assert ir.shape == g['impulse_response'].shape
  np.testing.assert_allclose(ir, g['impulse_response'], rtol=0, atol=1e-7)
  sig = O.filtered_noise_get_signal(c['magnitudes'], g['noise'], int(g['window_size']))
  np.testing.assert_allclose(sig, g['signal'], rtol=0, atol=1e-6)


def test_resample_matches_reference_source():
  g = load_golden('resample')
  np.testing.assert_array_equal(O.resample(g['x'], 576, 'window'), g['window_576'])
  for n in (576, 1728, 900):
    np.testing.assert_array_equal(O.resample(g['x'], n, 'linear'), g['linear_%d' % n])


def test_add_matches_reference_source():
  g = load_golden('add')
  np.testing.assert_array_equal(O.add(g['signal_one'], g['signal_two']), g['signal'])


# ---- closed forms the kernels evaluate -------------------------------
def test_window_upsample_closed_form():
  rng = np.random.default_rng(0)
  x = rng.standard_normal((2, 11, 5))
  lit = O.upsample_with_windows(x, 11 * 48, dtype=np.float64)
  cf = O.upsample_with_windows_closed_form(x, 11 * 48, dtype=np.float64)
  np.testing.assert_allclose(cf, lit, rtol=0, atol=1e-14)


@pytest.mark.parametrize('n,f,m,ws', [(3200, 50, 65, 0), (640, 10, 33, 17), (100, 7, 9, 0)])
def test_fft_convolve_equals_direct_time_varying_fir(n, f, m, ws):
  rng = np.random.default_rng(1)
  mags = np.abs(rng.standard_normal((2, f, m)))
  noise = rng.uniform(-1, 1, (2, n))
  ir = O.frequency_impulse_response(mags, ws, dtype=np.float64)
  lit = O.fft_convolve(noise, ir, dtype=np.float64)
  direct = O.time_varying_fir_direct(noise, ir)
  np.testing.assert_allclose(direct, lit, rtol=0, atol=1e-12)


def test_all_ones_magnitudes_is_a_two_sample_delay():
  """M=65, window 0: L=128, peak 64, crop 62 -> the reference delays by 2 (F7c)."""
  x = np.random.default_rng(2).uniform(-1, 1, (1, 640))
  y = O.frequency_filter(x, np.ones((1, 10, 65)), window_size=0, dtype=np.float64)
  np.testing.assert_allclose(y[:, 2:], x[:, :-2], atol=1e-12)
  np.testing.assert_allclose(y[:, :2], 0.0, atol=1e-12)


def test_constant_f0_single_harmonic_known_answer():
  """cumsum is inclusive: sample t has phase 2*pi*f*(t+1)/sr."""
  sr, n, f = 16000, 640, 440.0
  sig = O.harmonic_get_signal(np.ones((1, 10, 1)), np.ones((1, 10, 1)),
                              np.full((1, 10, 1), f), n, sr, dtype=np.float64)
  t = np.arange(n)
  np.testing.assert_allclose(sig[0], np.sin(2 * np.pi * f * (t + 1) / sr), atol=1e-9)


# ---- the reference's own known-answer tests, re-expressed -------------------------
@pytest.mark.parametrize('audio_size,ir_size', [(1000, 10), (10, 100)])
def test_fft_convolve_is_accurate(audio_size, ir_size):            # core_test.py:730-757
  audio, ir = np.ones([1, audio_size], np.float32), np.ones([1, 1, ir_size], np.float32)
  out = O.fft_convolve(audio, ir, padding='valid', delay_compensation=0)[0]
  target = scipy.signal.fftconvolve(audio[0], ir[0, 0], mode='full')
  assert np.abs(target - out).mean() <= 1e-3


@pytest.mark.parametrize('gain', [1.0, 0.1])
def test_delay_compensation_corrects_group_delay(gain):             # core_test.py:759-785
  n = 4 * 1024
  audio = np.sin(np.linspace(0, 200.0, n))[None, :].astype(np.float32)
  mags = gain * np.ones([1, 1025], np.float32)
  out = O.frequency_filter(audio, mags, window_size=257)
  assert np.abs(out - gain * audio).mean() <= 1e-3


@pytest.mark.parametrize('fft_size,window_size', [(2048, 257), (2048, 256), (2048, 0),
                                                  (1024, 1025 + 100)])
def test_frequency_impulse_response_gives_correct_size(fft_size, window_size):  # :825-855
  mags = np.ones([1, 5, fft_size // 2 + 1], np.float32)
  ir = O.frequency_impulse_response(mags, window_size)
  if window_size <= 0 or window_size > fft_size:
    target = fft_size
  else:
    target = window_size if window_size % 2 else window_size - 1
  assert ir.shape[-1] == target


@pytest.mark.parametrize('sr', [4000, 16000, 44100])
def test_silent_above_nyquist(sr):                                 # core_test.py:484-503
  n = 1000
  for ratio in (1.1, 1.5, 2.0):
    f = np.full((2, n, 3), ratio * sr / 2.0, np.float32)
    audio = O.oscillator_bank(f, np.ones_like(f), sample_rate=sr)
    assert np.all(audio == 0.0)


@pytest.mark.parametrize('method', ['linear', 'window'])
def test_upsample_accuracy(method):                                # core_test.py:242-267
  x = np.array([0.0, 1.0, 0.5, -0.3, 0.8], np.float32)[None, :, None]
  n = 16000
  y = O.resample(x, n, method=method)
  idx = np.arange(5) * (n // 5)
  np.testing.assert_allclose(y[0, idx, 0], x[0, :, 0], atol=1e-3)


def test_value_errors():                                           # core_test.py:295-381, 787-886
  with pytest.raises(ValueError):
    O.upsample_with_windows(np.ones((2, 10)), 100)                  # not 3-D
  with pytest.raises(ValueError):
    O.upsample_with_windows(np.ones((1, 10, 1)), 5)                 # downsampling
  with pytest.raises(ValueError):
    O.upsample_with_windows(np.ones((1, 10, 1)), 105)               # not divisible
  with pytest.raises(ValueError):
    O.resample(np.ones((1, 10, 1)), 100, method='bogus')
  with pytest.raises(ValueError):
    O.fft_convolve(np.ones((2, 100)), np.ones((3, 10, 5)))          # batch mismatch
  with pytest.raises(ValueError):
    O.fft_convolve(np.ones((1, 100)), np.ones((1, 30, 5)))          # frame mismatch
  with pytest.raises(ValueError):
    O.fft_convolve(np.ones((1, 100)), np.ones((1, 10, 5)), padding='bogus')


# ---- fp32 drift facts that define the parity contract ------------------
def test_angular_cumsum_tracks_fp64_truth_and_plain_cumsum_drifts():
  rng = np.random.default_rng(3)
  n, f, k = 32000, 500, 30
  amps = np.ones((1, f, 1)); hd = np.ones((1, f, k)) / k
  f0 = (220 + 3 * rng.standard_normal((1, f, 1)))
  truth = O.harmonic_get_signal(amps, hd, f0, n, dtype=np.float64)
  ang = O.harmonic_get_signal(amps, hd, f0, n, use_angular_cumsum=True)
  seq = O.harmonic_get_signal(amps, hd, f0, n, use_angular_cumsum=False)
  assert np.abs(ang - truth).max() < 2e-2
  assert np.abs(seq - truth).max() > 5 * np.abs(ang - truth).max()


def test_device_noise_restatement_is_uniform():
  x = O.device_uniform_noise(4, 4096, seed=7)
  assert x.dtype == np.float32 and x.min() >= -1.0 and x.max() < 1.0
  assert abs(x.mean()) < 0.03 and abs(x.var() - 1 / 3) < 0.02
  # counter-based: rows and offsets are independent of batch tiling
  y = O.device_uniform_noise(2, 4096, seed=7, batch_offset=2)
  np.testing.assert_array_equal(x[2:], y)
  # Philox4x32 known answers (Random123 kat_vectors): 10 rounds (the paper's default, what the device generator runs)
  # and 7 rounds (measured in round 3, not adopted), ctr = 0 / key = 0 and ctr = key = all ones
  w = O.philox4x32_10(0, 0, 0, 0, 0, 0)
  assert [int(v) for v in w] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
  assert O.NOISE_ROUNDS == 10
  w = O.philox4x32(0, 0, 0, 0, 0, 0, 7)
  assert [int(v) for v in w] == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
  ones = 0xffffffff
  w = O.philox4x32(ones, ones, ones, ones, ones, ones, 7)
  assert [int(v) for v in w] == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]


# ---- effects.Reverb ----------------------------------------------
REVERB_CASES = ['reverb_b2_dry', 'reverb_b2_wet_rank3', 'reverb_trainable']


@pytest.mark.parametrize('name', REVERB_CASES)
def test_reverb_matches_reference_source(name):
  g = load_golden(name)
  sig = O.reverb(g['audio'], g['ir'], add_dry=bool(g['add_dry']))
  assert sig.dtype == np.float32 and sig.shape == g['signal'].shape
  np.testing.assert_allclose(sig, g['signal'], rtol=0, atol=1e-6)
  # and the fp64 direct form the HIP path is checked against at sizes the FFT form is slow for
  truth = O.reverb_direct(g['audio'], g['ir'], add_dry=bool(g['add_dry']))
  np.testing.assert_allclose(sig, truth, rtol=0, atol=2e-5 * np.abs(truth).max())


def test_reverb_masks_the_dry_tap_and_is_causal():                 # effects.py:50-60, 113-117
  x = np.zeros((1, 64), np.float32)
  x[0, 5] = 1.0
  ir = np.zeros((1, 16), np.float32)
  ir[0, 0], ir[0, 3] = 7.0, 0.5                                    # tap 0 must be ignored
  wet = O.reverb(x, ir, add_dry=False)
  expect = np.zeros_like(x)
  expect[0, 8] = 0.5
  np.testing.assert_allclose(wet, expect, atol=1e-6)
  np.testing.assert_allclose(O.reverb(x, ir, add_dry=True), expect + x, atol=1e-6)


# ---- losses.SpectralLoss forward -------------------------------------
def test_spectral_loss_matches_reference_source():
  g = load_golden('spectral_loss')
  t, a = g['target_audio'], g['audio']
  np.testing.assert_allclose(O.compute_mag(a, 256), g['mag_256'], rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(O.spectral_loss(t, a), g['loss_default'], rtol=1e-5)
  np.testing.assert_allclose(O.spectral_loss(t, a, logmag_weight=1.0), g['loss_ae_gin'], rtol=1e-5)
  np.testing.assert_allclose(O.spectral_loss(t, a, fft_sizes=(512, 64), logmag_weight=0.5),
                             g['loss_two_scales'], rtol=1e-5)
  # fp64 truth is what the GPU path is held to
  np.testing.assert_allclose(O.spectral_loss(t, a, logmag_weight=1.0, dtype=np.float64),
                             g['loss_ae_gin'], rtol=2e-5)


def test_spectral_loss_is_zero_for_identical_signals_and_counts_padded_frames():
  x = np.random.default_rng(1).standard_normal((1, 1000)).astype(np.float32)
  assert O.spectral_loss(x, x, logmag_weight=1.0) == 0.0
  assert O.compute_mag(x, 64).shape == (1, 63, 33)                 # ceil(1000/16) frames, 33 bins
  assert O.compute_mag(x, 2048).shape == (1, 2, 1025)              # frames longer than the clip


# ---- core.streaming_harmonic_synthesis ---------------------------------
STREAMING_CASES = ['streaming_2frames_linear', 'streaming_2frames_window', 'streaming_nyquist_crossing',
                   'streaming_no_distribution', 'streaming_cubic', 'streaming_nearest_ragged', 'streaming_linear_ragged']


@pytest.mark.parametrize('name', STREAMING_CASES)
def test_streaming_synthesis_matches_reference_source(name):
  g = load_golden(name)
  audio, final_phase = O.streaming_harmonic_synthesis(
      g['f0_hz'], g['amplitudes'], g.get('harmonic_distribution'), g['initial_phase'],
      int(g['n_samples']), int(g['sample_rate']), str(g['amp_method']))
  assert audio.dtype == np.float32 and audio.shape == g['audio'].shape
  np.testing.assert_allclose(audio, g['audio'], rtol=0, atol=2e-6)
  np.testing.assert_allclose(final_phase, g['final_phase'], rtol=0, atol=1e-6)
  # fp64 truth (what the HIP path is held to) stays close on these short chunks
  a64, p64 = O.streaming_harmonic_synthesis(
      g['f0_hz'], g['amplitudes'], g.get('harmonic_distribution'), g['initial_phase'],
      int(g['n_samples']), int(g['sample_rate']), str(g['amp_method']), dtype=np.float64)
  assert np.abs(a64 - g['audio']).max() < 2e-3
  assert np.abs(p64 - g['final_phase']).max() < 5e-4      # the reference's own fp32 phase error


# ---- backward pass of Harmonic: analytic fp64 vs finite differences ----
@pytest.mark.parametrize('method,scale,normalize,n', [('window', True, True, 40), ('linear', True, True, 40),
                                                       ('window', False, False, 40),
                                                       # the materialised chain's argument space (round 5): other envelopes, ragged lengths
                                                       ('cubic', True, True, 40), ('nearest', True, True, 40),
                                                       ('linear', True, True, 43), ('cubic', False, False, 37)])
def test_harmonic_backward_matches_finite_differences(method, scale, normalize, n):
  rng = np.random.default_rng(3)
  b, f, k, sr = 1, 5, 6, 16000
  amps = rng.standard_normal((b, f, 1))
  hd = rng.standard_normal((b, f, k))
  if not scale:
    amps, hd = np.abs(amps) + 0.1, np.abs(hd) + 0.05
  f0 = rng.uniform(900.0, 1500.0, (b, f, 1))                      # harmonics 6.. cross Nyquist: masks are exercised
  g = rng.standard_normal((b, n))
  scale_fn = O.exp_sigmoid if scale else None

  def loss(a, h):
    y = O.harmonic(a, h, f0, n, sr, scale_fn, normalize, method, dtype=np.float64)
    return float(np.sum(y * g))
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, scale_fn, normalize, method)
  eps = 1e-6
  for idx in [(0, 0, 0), (0, 2, 0), (0, 4, 0)]:
    d = np.zeros_like(amps); d[idx] = eps
    fd = (loss(amps + d, hd) - loss(amps - d, hd)) / (2 * eps)
    np.testing.assert_allclose(ga[idx], fd, rtol=1e-5, atol=1e-8)
  for idx in [(0, 0, 0), (0, 1, 3), (0, 3, 5), (0, 4, 2), (0, 2, 1)]:
    d = np.zeros_like(hd); d[idx] = eps
    fd = (loss(amps, hd + d) - loss(amps, hd - d)) / (2 * eps)
    np.testing.assert_allclose(gh[idx], fd, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('m,window_size,n_frames,n,scale', [(9, 0, 4, 64, True), (65, 0, 3, 192, True),
                                                             (9, 5, 4, 50, False)])
def test_filtered_noise_backward_matches_finite_differences(m, window_size, n_frames, n, scale):
  rng = np.random.default_rng(5)
  mags = rng.standard_normal((2, n_frames, m)) + (4.0 if scale else 0.0)
  noise = rng.uniform(-1, 1, (2, n))
  g = rng.standard_normal((2, n))
  scale_fn = O.exp_sigmoid if scale else None

  def loss(mm):
    return float(np.sum(O.filtered_noise(mm, noise, window_size, scale_fn, dtype=np.float64) * g))
  gm = O.filtered_noise_backward(mags, noise, g, window_size, scale_fn)
  eps = 1e-6
  for idx in [(0, 0, 0), (1, n_frames - 1, m - 1), (0, 1, m // 2), (1, 2, 1)]:
    d = np.zeros_like(mags); d[idx] = eps
    fd = (loss(mags + d) - loss(mags - d)) / (2 * eps)
    np.testing.assert_allclose(gm[idx], fd, rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('b,bir,n,l,add_dry', [(2, 2, 30, 12, True), (3, 1, 25, 40, False)])
def test_reverb_backward_matches_finite_differences(b, bir, n, l, add_dry):
  rng = np.random.default_rng(9)
  x = rng.standard_normal((b, n))
  h = rng.standard_normal((bir, l))
  g = rng.standard_normal((b, n))

  def loss(xx, hh):
    return float(np.sum(O.reverb(xx, hh, add_dry, dtype=np.float64) * g))
  dx, dh = O.reverb_backward(x, h, g, add_dry)
  assert dx.shape == x.shape and dh.shape == h.shape
  eps = 1e-6
  for idx in [(0, 0), (b - 1, n - 1), (0, n // 2)]:
    d = np.zeros_like(x); d[idx] = eps
    np.testing.assert_allclose(dx[idx], (loss(x + d, h) - loss(x - d, h)) / (2 * eps), rtol=1e-6, atol=1e-8)
  for idx in [(0, 0), (0, 1), (bir - 1, l - 1), (0, l // 2)]:
    d = np.zeros_like(h); d[idx] = eps
    np.testing.assert_allclose(dh[idx], (loss(x, h + d) - loss(x, h - d)) / (2 * eps), rtol=1e-6, atol=1e-8)


def test_spectral_loss_backward_matches_finite_differences():
  rng = np.random.default_rng(2)
  t = 0.3 * rng.standard_normal((2, 300))
  a = t * 0.8 + 0.05 * rng.standard_normal((2, 300))
  kw = dict(fft_sizes=(128, 64, 16), mag_weight=1.0, logmag_weight=0.7)
  g = O.spectral_loss_backward(t, a, **kw)
  eps = 1e-6
  for idx in [(0, 0), (0, 150), (1, 299), (1, 37), (0, 291)]:
    d = np.zeros_like(a); d[idx] = eps
    fd = (O.spectral_loss(t, a + d, dtype=np.float64, **kw) - O.spectral_loss(t, a - d, dtype=np.float64, **kw)) / (2 * eps)
    np.testing.assert_allclose(g[idx], fd, rtol=1e-4, atol=1e-9)


def test_spectral_loss_backward_with_frames_that_are_not_powers_of_two():
  """gin/models/vst/vst_48k.gin:56 asks for frames of 6144, 3072, .. 192 samples: tf.signal.stft (fft_length=None) transforms the
  ENCLOSING power of two - 3 * 2^k samples zero-padded to 2^(k+2), 2^(k+1) + 1 bins.  The gradient restatement took the frame
  size for the transform's length until round 5 (it had only ever met powers of two); against central differences."""
  rng = np.random.default_rng(3)
  t = 0.3 * rng.standard_normal((2, 400))
  a = t * 0.8 + 0.05 * rng.standard_normal((2, 400))
  assert O.stft(a, 48).shape == (2, -(-400 // 12), 33) and O.stft(a, 192).shape[-1] == 129
  kw = dict(fft_sizes=(192, 48, 64), mag_weight=1.0, logmag_weight=0.7)
  g = O.spectral_loss_backward(t, a, **kw)
  eps = 1e-6
  for idx in [(0, 0), (0, 201), (1, 399), (1, 47), (0, 350)]:
    d = np.zeros_like(a); d[idx] = eps
    fd = (O.spectral_loss(t, a + d, dtype=np.float64, **kw) - O.spectral_loss(t, a - d, dtype=np.float64, **kw)) / (2 * eps)
    np.testing.assert_allclose(g[idx], fd, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize('method', ['window', 'linear'])
def test_harmonic_backward_f0_matches_finite_differences(method):
  rng = np.random.default_rng(13)
  b, f, k, n, sr = 1, 6, 5, 48, 16000
  amps = rng.standard_normal((b, f, 1))
  hd = rng.standard_normal((b, f, k))
  f0 = rng.uniform(300.0, 700.0, (b, f, 1))                       # 5 * 700 < 8000: no mask flips under the perturbation
  g = rng.standard_normal((b, n))

  def loss(ff):
    return float(np.sum(O.harmonic(amps, hd, ff, n, sr, O.exp_sigmoid, True, method, dtype=np.float64) * g))
  _, _, gf = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid, True, method, with_f0=True)
  eps = 1e-4
  for j in range(f):
    d = np.zeros_like(f0); d[0, j, 0] = eps
    fd = (loss(f0 + d) - loss(f0 - d)) / (2 * eps)
    np.testing.assert_allclose(gf[0, j, 0], fd, rtol=1e-5, atol=1e-9)


# ---- the rest of the path's argument space (golden: reference source on the TF stand-in) -------------
def _resample_golden_entries(g):
  for key in g:
    parts = key.split('_')
    if parts[0] in ('up', 'small', 'ragged', 'down', 'x4d') and len(parts) == 3:
      yield key, parts[0], parts[1], parts[2] == 'endpoint'


def test_resample_every_method_matches_reference_source():         # core.py:573-642
  g = load_golden('resample_methods')
  seen = 0
  for key, kind, method, add_endpoint in _resample_golden_entries(g):
    n = g[key].shape[1]
    src = {'up': g['x'], 'ragged': g['x'], 'small': g['x_small'], 'x4d': g['x_4d']}.get(kind)
    if kind == 'down':
      src = g['up_%s_%s' % (method, 'endpoint' if add_endpoint else 'noendpoint')]
    out = O.resample(src, n, method=method, add_endpoint=add_endpoint)
    assert out.shape == g[key].shape and out.dtype == np.float32
    np.testing.assert_array_equal(out, g[key], err_msg=key)
    seen += 1
  assert seen == 4 * 2 * 2 + 3 * 2 * 3


@pytest.mark.parametrize('method', ['nearest', 'linear', 'cubic', 'window'])
@pytest.mark.parametrize('add_endpoint', [True, False])
def test_upsample_and_downsample_accuracy_every_method(method, add_endpoint):      # core_test.py:219-293
  n_small, n_large = 5, 16000
  n_total = int(n_large / n_small * (n_small - 1)) if add_endpoint else n_large - 1
  idx = np.linspace(0, n_total, n_small).astype(int)
  before = (1.0 - np.sin(np.linspace(0, np.pi, n_small)))[None, :, None]
  if method == 'window' and not add_endpoint:
    after = O.resample(before, n_large, method=method, add_endpoint=add_endpoint)   # 16000 % 4 == 0
  else:
    after = O.resample(before, n_large, method=method, add_endpoint=add_endpoint)
  np.testing.assert_allclose(after[0, idx, 0], before[0, :, 0], atol=1e-3)
  if method != 'window':
    big = (1.0 - np.sin(np.linspace(0, np.pi, n_large)))[None, :, None]
    small = O.resample(big, n_small, method=method, add_endpoint=add_endpoint)
    np.testing.assert_allclose(big[0, idx, 0], small[0, :, 0], atol=1e-3)
  with pytest.raises(ValueError):
    O.resample(np.ones((2, 5, 3, 2)), 100, method='window')         # 4-D: core_test.py:175-198


def test_bicubic_weights_are_a_partition_of_unity_and_interpolate():
  near, far = O._cubic_coeffs_table()
  off = np.arange(1025)
  total = far[off].astype(np.float64) + near[off] + near[1024 - off] + far[1024 - off]
  np.testing.assert_allclose(total, 1.0, atol=3e-7)
  assert (far[0], near[0], near[1024], far[1024]) == (0.0, 1.0, 0.0, 0.0)      # exact at the frame points
  # (the A = -0.75 kernel reproduces constants, not straight lines: only Keys' A = -0.5 does)
  flat = np.full((1, 8, 1), 3.25, np.float32)
  np.testing.assert_allclose(O.resample(flat, 64, method='cubic'), 3.25, atol=1e-6)
  x = np.random.default_rng(0).standard_normal((1, 8, 1)).astype(np.float32)
  np.testing.assert_array_equal(O.resample(x, 64, method='cubic')[0, ::8, 0], x[0, :, 0])


def test_fft_convolve_crops_match_reference_source():               # core.py:1338-1379
  g = load_golden('fft_convolve_crops')
  for key, ir, padding, delay in [('valid_d0', 'ir', 'valid', 0), ('valid_d5', 'ir', 'valid', 5),
                                  ('valid_auto', 'ir', 'valid', -1), ('same_d40', 'ir', 'same', 40),
                                  ('one_valid_d0', 'ir_one', 'valid', 0), ('one_valid_auto', 'ir_one', 'valid', -1)]:
    out = O.fft_convolve(g['audio'], g[ir], padding=padding, delay_compensation=delay)
    assert out.shape == g[key].shape, key
    np.testing.assert_allclose(out, g[key], rtol=0, atol=1e-6, err_msg=key)
    direct = O.time_varying_fir_direct(g['audio'], np.broadcast_to(g[ir], (2,) + g[ir].shape[1:]),
                                       delay_compensation=delay, n_out=out.shape[1])
    np.testing.assert_allclose(direct, g[key], rtol=0, atol=2e-6, err_msg=key)


SYNTHESIS_CASES = ['synthesis_shifts_window', 'synthesis_shifts_only', 'synthesis_cubic',
                   'synthesis_nearest_angular', 'synthesis_linear_ragged']


@pytest.mark.parametrize('name', SYNTHESIS_CASES)
def test_harmonic_synthesis_argument_space_matches_reference_source(name):     # core.py:1048-1111
  g = load_golden(name)
  out = O.harmonic_synthesis(g['f0_hz'], g['amplitudes'], g.get('harmonic_shifts'), g.get('harmonic_distribution'),
                             n_samples=int(g['n_samples']), sample_rate=int(g['sample_rate']),
                             amp_resample_method=str(g['amp_method']), use_angular_cumsum=bool(g['angular']))
  assert out.shape == g['audio'].shape
  np.testing.assert_allclose(out, g['audio'], rtol=0, atol=2e-6)


def test_harmonic_with_cubic_envelope_matches_reference_source():
  g = load_golden('harmonic_cubic_amp')
  c = O.harmonic_get_controls(g['amplitudes'], g['harmonic_distribution'], g['f0_hz'], int(g['sample_rate']))
  sig = O.harmonic_get_signal(c['amplitudes'], c['harmonic_distribution'], c['f0_hz'], int(g['n_samples']),
                              int(g['sample_rate']), str(g['amp_method']), bool(g['angular']))
  np.testing.assert_allclose(sig, g['signal'], rtol=0, atol=2e-6)


def test_exp_decay_ir_backward_matches_finite_differences():        # effects.py:144-151
  rng = np.random.default_rng(17)
  b, l = 2, 300
  gain, decay = rng.standard_normal((b, 1)), rng.uniform(-1.0, 2.0, (b, 1))
  noise, g = rng.uniform(-1, 1, (1, l)), rng.standard_normal((b, l))
  loss = lambda ga, de: float(np.sum(O.exp_decay_ir(ga, de, noise, dtype=np.float64) * g))
  gg, gd = O.exp_decay_ir_backward(gain, decay, noise, g)
  eps = 1e-6
  for i in range(b):
    e = np.zeros((b, 1)); e[i] = eps
    assert abs((loss(gain + e, decay) - loss(gain - e, decay)) / (2 * eps) - gg[i, 0]) <= 1e-6 * max(1, abs(gg[i, 0]))
    assert abs((loss(gain, decay + e) - loss(gain, decay - e)) / (2 * eps) - gd[i, 0]) <= 1e-6 * max(1, abs(gd[i, 0]))
  ir = O.exp_decay_ir(np.full((1, 1), 2.0), np.full((1, 1), 4.0), np.ones((1, l)))      # the trainable initial values
  assert ir.shape == (1, l) and ir[0, 0] > ir[0, 1] > ir[0, -1] > 0                      # effects.py:158-168


@pytest.mark.parametrize('name', ['exp_decay_reverb_b3', 'exp_decay_reverb_trainable'])
def test_exp_decay_reverb_matches_reference_source(name):          # effects.py:120-199
  g = load_golden(name)
  ir = O.exp_decay_ir(g['gain'].reshape(-1, 1), g['decay'].reshape(-1, 1), g['noise'])
  np.testing.assert_allclose(np.broadcast_to(ir, g['ir'].shape), g['ir'], rtol=2e-6, atol=1e-9)
  sig = O.reverb(g['audio'], ir, add_dry=bool(g['add_dry']))
  np.testing.assert_allclose(sig, g['signal'], rtol=0, atol=1e-5)


def test_harmonic_oscillator_bank_matches_reference_source():
  """core.harmonic_oscillator_bank on audio-rate inputs (core.py:966-1025): the restatement against the reference's
  own source run on the TF stand-in, both phase accumulations; and the fp64 run stays within the drift the fp32
  sequential sum is known for."""
  g = load_golden('harmonic_oscillator_bank')
  for mode, angular, phase0 in (('angular', True, g['initial_phase']), ('cumsum', False, None)):
    audio, final_phase = O.harmonic_oscillator_bank(g['frequency'], g['amplitude_envelopes'], phase0,
                                                    int(g['sample_rate']), use_angular_cumsum=angular)
    np.testing.assert_allclose(audio, g['audio_' + mode], rtol=0, atol=2e-5)
    np.testing.assert_allclose(final_phase, g['final_phase_' + mode], rtol=0, atol=2e-5)
    a64, _ = O.harmonic_oscillator_bank(g['frequency'].astype(np.float64), g['amplitude_envelopes'].astype(np.float64),
                                        None if phase0 is None else phase0.astype(np.float64), int(g['sample_rate']),
                                        use_angular_cumsum=angular)
    assert np.abs(a64 - g['audio_' + mode]).max() < 5e-2


def test_spectral_loss_every_term_matches_reference_source():
  """losses.SpectralLoss with every term, loss type and the weights mask (losses.py:102-128, 199-236): the restatement
  against the reference's own source run on the TF stand-in."""
  g = load_golden('spectral_loss_terms')
  t, a = g['target_audio'], g['audio']
  sizes = tuple(int(v) for v in g['fft_sizes'])
  kw = {k: float(g[k]) for k in ('mag_weight', 'delta_time_weight', 'delta_freq_weight', 'cumsum_freq_weight',
                                 'logmag_weight')}
  for key, loss_type, w in (('l1', 'L1', None), ('l2', 'L2', None), ('cosine', 'COSINE', None),
                            ('l1_weighted', 'L1', g['weights']), ('l2_weighted', 'L2', g['weights']),
                            ('cosine_weighted', 'COSINE', g['weights'])):
    np.testing.assert_allclose(O.spectral_loss(t, a, sizes, loss_type, weights=w, **kw), np.ravel(g[key])[0], rtol=1e-6)
  np.testing.assert_allclose(O.spectral_loss(t, a, (256,), 'L2', mag_weight=0.0, delta_time_weight=1.0),
                             np.ravel(g['l2_delta_time_only'])[0], rtol=1e-6)
  np.testing.assert_allclose(O.spectral_loss(t, a, (2048, 64), 'L1', mag_weight=0.0, cumsum_freq_weight=1.0),
                             np.ravel(g['l1_cumsum_only'])[0], rtol=1e-6)
  # core.diff / cumsum by hand on a tiny spectrogram
  x = np.arange(24, dtype=np.float32).reshape(1, 4, 6) ** 2
  np.testing.assert_array_equal(O.diff(x, 1), x[:, 1:] - x[:, :-1])
  np.testing.assert_array_equal(O.diff(x, 2), x[:, :, 1:] - x[:, :, :-1])


def test_nearest_resize_rounds_like_roundf_not_like_a_rounded_sum():
  """tf.compat.v1.image.resize(NEAREST, align_corners=True) takes roundf(out * scale) (resize_nearest_neighbor_op.cc).  With two
  frames and 111 samples the position of sample 55 is 55 * fp32(1 / 110) = 0.49999997: roundf gives frame 0; floor(pos + 0.5) in
  fp32 - the sum rounds to 1.0 - gave frame 1 (the restatement until round 5; tools/fuzz_parity.py found the kernel, which calls
  roundf, and this oracle one whole frame apart)."""
  pos = O._legacy_resize_positions(2, 111, True)
  assert pos[55] < np.float32(0.5) and np.float32(pos[55] + np.float32(0.5)) == np.float32(1.0)
  x = np.array([[[1.0], [2.0]]], np.float32)
  out = O.resize_nearest_legacy(x, 111, True)[0, :, 0]
  np.testing.assert_array_equal(out[:56], 1.0)
  np.testing.assert_array_equal(out[56:], 2.0)


def test_loudness_matches_the_reference_source_and_its_gradient_finite_differences():
  """spectral_ops.compute_loudness as SpectralLoss calls it (n_fft = 2048): the restatement against the golden vectors (the
  reference's own source with librosa's fft_frequencies / A_weighting restated from its published formula), the analytic
  gradient against central differences - through the dB floor too (a silent stretch)."""
  g = load_golden('spectral_loss_loudness')
  t, a = g['target_audio'], g['audio']
  np.testing.assert_allclose(O.compute_loudness(a, dtype=np.float32), g['loudness_audio'], rtol=0, atol=2e-4)
  assert O.compute_loudness(a).shape == (2, 1 + 3000 // 64)
  for key, kw in (('l1_only', dict(mag_weight=0.0, loudness_weight=1.0)), ('l2_only', dict(loss_type='L2', mag_weight=0.0, loudness_weight=1.0)),
                  ('cosine_only', dict(loss_type='COSINE', mag_weight=0.0, loudness_weight=1.0))):
    np.testing.assert_allclose(float(O.spectral_loss(t, a, dtype=np.float32, **kw)), float(np.ravel(g[key])[0]), rtol=2e-5)
  w = O.a_weighting_db(16000, 2048)
  assert w.shape == (1025,) and w[0] == -80.0 and abs(w[128] - 0.0) < 0.01                     # 0 dB at 1 kHz (bin 128 of 2048 at 16 kHz)
  rng = np.random.default_rng(1)
  a64 = a.astype(np.float64)
  gl = rng.standard_normal(O.compute_loudness(a64, dtype=np.float64).shape)
  gb = O.compute_loudness_backward(a64, gl)
  eps = 1e-6
  for idx in [(0, 5), (1, 700), (0, 2500), (1, 2999), (1, 1400)]:
    d = np.zeros_like(a64); d[idx] = eps
    fd = ((O.compute_loudness(a64 + d, dtype=np.float64) - O.compute_loudness(a64 - d, dtype=np.float64)) * gl).sum() / (2 * eps)
    np.testing.assert_allclose(gb[idx], fd, rtol=1e-4, atol=1e-9)


def test_hann_window_is_tensorflows_not_the_textbook_periodic_one():
  """tf.signal.hann_window(n) (tensorflow/python/ops/signal/window_ops.py, _raised_cosine_window; periodic=True is what
  core.py:698, 1505 and tf.signal.stft pass): n' = window_length + periodic * even - 1 with even = 1 - window_length % 2 -
  periodic for EVEN lengths, the SYMMETRIC window for odd ones (the constructor's window_size=257 on a response longer than
  that: core_test.py:775), ones([1]) for a single sample.  Known answers worked out from that definition; rounds 1-5
  divided by n whatever its parity, in the oracle AND in the TF stand-in the fixtures are made with."""
  np.testing.assert_allclose(O.hann_window_periodic(4, np.float64), [0.0, 0.5, 1.0, 0.5], atol=1e-15)        # periodic: d = 4
  np.testing.assert_allclose(O.hann_window_periodic(5, np.float64), [0.0, 0.5, 1.0, 0.5, 0.0], atol=1e-15)   # symmetric: d = 4
  np.testing.assert_array_equal(O.hann_window_periodic(1), [1.0])
  w = O.hann_window_periodic(257, np.float64)
  assert w[0] == 0.0 and abs(w[256]) < 1e-30 and w[128] == 1.0 and np.allclose(w, w[::-1], atol=1e-15)
  w = O.hann_window_periodic(128, np.float64)
  assert w[0] == 0.0 and w[64] == 1.0 and np.allclose(w[1:], w[:0:-1], atol=1e-15) and w[127] > 0.0
  # an impulse response of one sample comes back as it is (a window of [1.0]; core.py:1477-1531)
  ir = np.array([[[0.75]]], np.float32)
  np.testing.assert_array_equal(O.apply_window_to_impulse_response(ir, 0), ir)


def test_the_subdifferential_check_of_the_loss_gradient_is_not_vacuous():
  """tests/test_gpu_parity.py::check_loss_case holds SpectralLoss's gradient to the subdifferential the oracle returns with
  fp32_envelope (round 6: the criterion tools/fuzz_parity.py's campaigns end at zero failures with).  Here, without a GPU: the
  exact gradient passes with room, and each of three structural mistakes a kernel could make - a frame missing from the
  overlap-add, a stretch of samples at half their value, one scale of the loss forgotten - fails it."""
  import test_gpu_parity as P
  for seed, (n, sizes, mw, lw) in enumerate([(3000, (1024, 64, 384), 1.0, 0.5), (12345, (2048, 16), 0.5, 1.0), (1025, (16, 1024, 768, 2048), 1.0, 0.0)]):
    rng = np.random.default_rng(700 + seed)
    t = (0.3 * rng.standard_normal((3, n))).astype(np.float32)
    a = (0.8 * t + 0.05 * rng.standard_normal((3, n))).astype(np.float32)
    a[0, n // 2: n // 2 + n // 8] = 0.0
    exact = O.spectral_loss_backward(t, a, sizes, mw, lw)
    ref, env = O.spectral_loss_backward(t, a, sizes, mw, lw, fp32_envelope=5e-6)

    def verdict(g):
      atol, err = P.loss_gradient_excess(g, ref, env)
      return float(np.quantile(err, 0.999)) <= atol and float(err.max()) <= 30.0 * atol

    assert verdict(exact)
    assert verdict(exact.astype(np.float32).astype(np.float64))              # ... and its rounding to fp32
    size = max(sizes)
    broken = exact.copy()                                                     # one frame of the largest size left out of one row
    one = O.spectral_loss_backward(t[1:2, :min(n, size)], a[1:2, :min(n, size)], (size,), mw, lw)
    broken[1, :one.shape[1]] -= one[0] * (one.size and 1.0) * 0.5
    assert not verdict(broken)
    halved = exact.copy()
    halved[2, n // 3: n // 3 + 16] *= 0.5                                     # sixteen samples at half their value
    assert not verdict(halved)
    assert not verdict(O.spectral_loss_backward(t, a, sizes[:-1], mw, lw))    # a scale forgotten
    assert float((env[1:] <= 0.05 * np.abs(ref).max()).mean()) >= 0.5 or n < 4 * size


def test_exact_resize_positions_is_a_checker_switch_and_never_the_default():
  """oracle.exact_resize_positions (round 6; tools/fuzz_parity.py's streaming family): inside the context the bilinear legacy resize
  takes t n_in / n_out in exact arithmetic; outside - before, after, and after an exception inside - TF's fl32(t fl32(n_in / n_out)).
  Frame sizes that are powers of two: the two agree to the last bit.  Frames of 252 samples: they differ, by no more than the
  frame-to-frame step times pos 2^-23."""
  rng = np.random.default_rng(5)
  x = rng.uniform(60.0, 600.0, (2, 5, 1))
  tf_256 = O.resample(x, 5 * 256, method='linear', dtype=np.float64)
  tf_252 = O.resample(x, 5 * 252, method='linear', dtype=np.float64)
  with O.exact_resize_positions():
    ex_256 = O.resample(x, 5 * 256, method='linear', dtype=np.float64)
    ex_252 = O.resample(x, 5 * 252, method='linear', dtype=np.float64)
  np.testing.assert_array_equal(ex_256, tf_256)
  d = np.abs(ex_252 - tf_252).max()
  step = np.abs(np.diff(x, axis=1)).max()
  assert 0.0 < d <= step * 5 * 2.0 ** -22
  # exact positions against the definition: frame j + r / 252 of a linear ramp between frames
  t = np.arange(5 * 252) / 252.0
  j = np.minimum(np.floor(t).astype(int), 4); j1 = np.minimum(j + 1, 4)
  want = x[:, j, 0] + (x[:, j1, 0] - x[:, j, 0]) * (t - j)[None]
  np.testing.assert_allclose(ex_252[:, :, 0], want, rtol=0, atol=1e-10)
  with pytest.raises(RuntimeError):
    with O.exact_resize_positions():
      raise RuntimeError('inside')
  np.testing.assert_array_equal(O.resample(x, 5 * 252, method='linear', dtype=np.float64), tf_252)       # the default is back


def test_spectral_loss_value_bounds_are_tight_without_a_null_and_open_only_upwards_with_one():
  """oracle.spectral_loss_value_bounds (round 6, fuzz seed loss:107037044).  Ordinary signals: no null bin, and the interval is the
  fp64 value +- 2.5e-4 of it (every magnitude moved by what fp32 knows it to, all in the same direction).  A frame made to cancel at Nyquist (its
  windowed samples alternate to 1e-7 of their size): one null per signal planted, the interval's lower end moves by that bin's
  exact term only, its upper end by 110 nats over the number of terms - and the reference's own fp32 op order lands inside."""
  rng = np.random.default_rng(3)
  sizes, n = (256, 64), 1024
  t = (0.3 * rng.standard_normal((2, n))).astype(np.float32)
  a = (0.8 * t + 0.05 * rng.standard_normal((2, n))).astype(np.float32)
  v = float(O.spectral_loss(t, a, sizes, mag_weight=1.0, logmag_weight=1.0, dtype=np.float64))
  lo, hi, n_null = O.spectral_loss_value_bounds(t, a, sizes, 1.0, 1.0)
  assert n_null == 0 and lo <= v <= hi and hi - lo <= 5e-4 * v
  # plant the null: make the Nyquist bin of frame 0 (size 256) of row 0 of `a` cancel - move one sample by what is left of it
  w = O.hann_window_periodic(256, np.float64)
  alt = (-1.0) ** np.arange(256)
  a2 = a.astype(np.float64)
  a2[0, 100] -= float((a2[0, :256] * w * alt).sum()) / (w[100] * alt[100])
  a2 = a2.astype(np.float32)
  ny = abs(float((a2[0, :256].astype(np.float64) * w * alt).sum()))
  assert 0.0 < ny < 1e-6                                       # fp32 storage leaves ~1e-8 .. 1e-7 of it
  v2 = float(O.spectral_loss(t, a2, sizes, mag_weight=1.0, logmag_weight=1.0, dtype=np.float64))
  lo2, hi2, n_null2 = O.spectral_loss_value_bounds(t, a2, sizes, 1.0, 1.0)
  count = float(np.abs(O.stft(t, 256, dtype=np.float64)).size)
  assert n_null2 == 1 and lo2 <= v2 <= hi2
  assert 110.0 / count <= hi2 - lo2 <= (110.0 + 25.0) / count + 5e-4 * v2     # the null's exact term (<= 25 nats) + 110, once
  v32 = float(O.spectral_loss(t, a2, sizes, mag_weight=1.0, logmag_weight=1.0, dtype=np.float32))
  assert lo2 - 1e-4 * v2 <= v32 <= hi2 + 1e-4 * v2
