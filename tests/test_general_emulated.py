"""CPU checks of the plain-HIP kernels in ddsp_amd/csrc/general.hip, compiled for the host by
tests/hip_emu (TEST INFRASTRUCTURE: a kernel launch is a serial loop there) and called through the same
C ABI the product uses, against the oracle.  Index arithmetic, argument checks and launch geometry are
what this exercises; tests/test_gpu_general.py repeats the comparisons on the MI355X."""
import numpy as np
import pytest
from scipy import signal

from oracle import ddsp_oracle as oracle
from tests.hip_emu import emu
from ddsp_amd import _lib

METHODS = _lib.RESAMPLE_METHODS


@pytest.fixture(scope='module')
def lib():
  return emu.load()


def run_resample(lib, x, n, method, add_endpoint):
  x = emu.f32(x)
  b, f, c = x.shape
  out = np.full((b, n, c), np.nan, np.float32)
  rc = lib.ddsp_resample_ex_f32(emu.ptr(x), emu.ptr(out), b, f, n, c, METHODS[method], int(add_endpoint), None)
  return rc, out


@pytest.mark.parametrize('method', ['nearest', 'linear', 'cubic', 'window'])
@pytest.mark.parametrize('add_endpoint', [True, False])
@pytest.mark.parametrize('f,n,c', [(5, 16000, 1), (9, 640, 3), (17, 1024, 2), (64, 4096, 5)])
def test_resample_matches_oracle_upsampling(lib, method, add_endpoint, f, n, c):
  rng = np.random.default_rng(f * 31 + n)
  x = rng.standard_normal((2, f, c)).astype(np.float32)
  if method == 'window' and n % (f if add_endpoint else f - 1):
    with pytest.raises(ValueError):
      oracle.resample(x, n, method=method, add_endpoint=add_endpoint)
    rc, _ = run_resample(lib, x, n, method, add_endpoint)
    assert rc == -2
    return
  rc, out = run_resample(lib, x, n, method, add_endpoint)
  assert rc == 0
  ref = oracle.resample(x, n, method=method, add_endpoint=add_endpoint)
  # same fp32 operations in the same order for the three resize kernels; the window form is the closed form
  # of the overlap-add: rounding differs
  tol = 2e-6 if method == 'window' else 0.0
  np.testing.assert_allclose(out, ref, rtol=0, atol=tol)


@pytest.mark.parametrize('method', ['nearest', 'linear', 'cubic'])
@pytest.mark.parametrize('add_endpoint', [True, False])
def test_resample_downsampling_and_reference_accuracy_test(lib, method, add_endpoint):     # core_test.py:219-293
  n_small, n_large = 5, 16000
  big = (1.0 - np.sin(np.linspace(0, np.pi, n_large)))[None, :, None].astype(np.float32)
  rc, small = run_resample(lib, big, n_small, method, add_endpoint)
  assert rc == 0
  np.testing.assert_array_equal(small, oracle.resample(big, n_small, method=method, add_endpoint=add_endpoint))
  n_total = int(n_large / n_small * (n_small - 1)) if add_endpoint else n_large - 1
  idx = np.linspace(0, n_total, n_small).astype(int)
  np.testing.assert_allclose(big[0, idx, 0], small[0, :, 0], atol=1e-3)
  before = (1.0 - np.sin(np.linspace(0, np.pi, n_small)))[None, :, None].astype(np.float32)
  rc, after = run_resample(lib, before, n_large, method, add_endpoint)
  assert rc == 0
  np.testing.assert_allclose(after[0, idx, 0], before[0, :, 0], atol=1e-3)


def test_resample_argument_checks(lib):
  x = np.zeros((1, 4, 1), np.float32)
  out = np.zeros((1, 16, 1), np.float32)
  assert lib.ddsp_resample_ex_f32(None, emu.ptr(out), 1, 4, 16, 1, 1, 1, None) == -1
  assert lib.ddsp_resample_ex_f32(emu.ptr(x), emu.ptr(out), 1, 4, 16, 1, 7, 1, None) == -2
  assert lib.ddsp_resample_ex_f32(emu.ptr(x), emu.ptr(out), 1, 4, 3, 1, 3, 1, None) == -2      # window: downsampling
  assert lib.ddsp_resample_ex_f32(emu.ptr(x), emu.ptr(out), 1, 4, 15, 1, 3, 0, None) == 0      # 15 % (4-1) == 0
  assert lib.ddsp_resample_ex_f32(emu.ptr(x), emu.ptr(out), 1, 4, 16, 1, 3, 0, None) == -2     # 16 % 3 != 0


# ---- core.fft_convolve, any crop ------------------------------------------------------------------------
def python_crop(total, start, end):
  """The index range audio[:, start:-end] selects (crop_and_compensate_delay, core.py:1371-1379)."""
  return range(total)[start:-end]


def run_fir(lib, audio, ir, n_out, start):
  audio, ir = emu.f32(audio), emu.f32(ir)
  b, n = audio.shape
  bir, f, l = ir.shape
  out = np.full((b, n_out), np.nan, np.float32)
  rc = lib.ddsp_fft_convolve_f32(emu.ptr(audio), emu.ptr(ir), emu.ptr(out), b, bir, f, l, n, n_out, start, None)
  return rc, out


@pytest.mark.parametrize('audio_size,ir_size', [(1000, 10), (10, 100)])             # core_test.py:730-757
def test_fft_convolve_valid_is_accurate_reference_test(lib, audio_size, ir_size):
  audio = np.ones([1, audio_size], np.float32)
  ir = np.ones([1, 1, ir_size], np.float32)
  rc, out = run_fir(lib, audio, ir, ir_size + audio_size - 1, 0)
  assert rc == 0
  ref = signal.fftconvolve(audio[0], ir[0, 0])
  assert np.abs(ref - out[0]).mean() <= 1e-3
  np.testing.assert_allclose(out, oracle.fft_convolve(audio, ir, padding='valid', delay_compensation=0), atol=1e-3)


@pytest.mark.parametrize('padding', ['valid', 'same'])
@pytest.mark.parametrize('delay', [-1, 0, 5])
@pytest.mark.parametrize('b,bir,n,f,l', [(2, 2, 640, 10, 33), (3, 1, 1000, 8, 64), (1, 1, 250, 25, 7),
                                         (2, 2, 96, 1, 200), (2, 1, 777, 7, 16)])
def test_fft_convolve_any_crop_matches_oracle(lib, padding, delay, b, bir, n, f, l):
  rng = np.random.default_rng(n + l)
  audio = rng.standard_normal((b, n)).astype(np.float32)
  ir = rng.standard_normal((bir, f, l)).astype(np.float32) / np.sqrt(l)
  ref = oracle.fft_convolve(audio.astype(np.float64), ir.astype(np.float64), padding=padding,
                            delay_compensation=delay, dtype=np.float64)
  start = (l - 1) // 2 - 1 if delay < 0 else delay
  if ref.shape[1] == 0:
    # python's audio[:, start:-end] with end <= 0 (the FFT size leaves no slack to crop): the reference
    # returns an empty tensor; the host layer does the same without a launch
    assert padding == 'valid'
    return
  rc, out = run_fir(lib, audio, ir, ref.shape[1], start)
  assert rc == 0
  assert out.shape == ref.shape
  np.testing.assert_allclose(out, ref, atol=2e-5)


def test_fft_convolve_argument_checks(lib):
  a = np.zeros((2, 100), np.float32)
  h = np.zeros((2, 3, 8), np.float32)
  o = np.zeros((2, 100), np.float32)
  # 100 samples / 3 frames: frame_size 34, ceil(100 / 34) == 3 frames: accepted
  assert lib.ddsp_fft_convolve_f32(emu.ptr(a), emu.ptr(h), emu.ptr(o), 2, 2, 3, 8, 100, 100, 3, None) == 0
  # 100 samples / 7 frames: frame_size 15 gives 7 frames; / 9 frames: frame_size 12 gives 9 - take 11 (frame 10 -> 10)
  assert lib.ddsp_fft_convolve_f32(emu.ptr(a), emu.ptr(h), emu.ptr(o), 2, 2, 11, 8, 100, 100, 3, None) == -2
  assert lib.ddsp_fft_convolve_f32(emu.ptr(a), emu.ptr(h), emu.ptr(o), 2, 3, 3, 8, 100, 100, 3, None) == -2
  assert lib.ddsp_fft_convolve_f32(emu.ptr(a), None, emu.ptr(o), 2, 2, 3, 8, 100, 100, 3, None) == -1


# ---- the frame-rate tensors of core.harmonic_synthesis ---------------------------------------------------
@pytest.mark.parametrize('with_hd,with_shifts', [(True, True), (True, False), (False, True)])
def test_harmonic_envelopes(lib, with_hd, with_shifts):
  rng = np.random.default_rng(3)
  b, f, k = 2, 13, 11
  amp = rng.uniform(0.1, 1.0, (b, f, 1)).astype(np.float32)
  hd = rng.uniform(0.0, 1.0, (b, f, k)).astype(np.float32) if with_hd else None
  f0 = rng.uniform(50.0, 900.0, (b, f, 1)).astype(np.float32)
  shifts = (0.05 * rng.standard_normal((b, f, k))).astype(np.float32) if with_shifts else None
  freq = np.full((b, f, k), np.nan, np.float32)
  amps = np.full((b, f, k), np.nan, np.float32)
  rc = lib.ddsp_harmonic_envelopes_f32(emu.ptr(amp), emu.ptr(hd), emu.ptr(f0), emu.ptr(shifts), emu.ptr(freq),
                                       emu.ptr(amps), b, f, k, None)
  assert rc == 0
  ref_f = oracle.get_harmonic_frequencies(f0, k)                       # fp32, core.py:1028-1045
  if with_shifts:
    ref_f = ref_f * (np.float32(1.0) + shifts)                         # core.py:1089-1090
  np.testing.assert_array_equal(freq, ref_f)
  np.testing.assert_array_equal(amps, amp * hd if with_hd else np.broadcast_to(amp, (b, f, k)))


# ---- dL/d f0_hz of Harmonic ------------------------------------------------------------------------------
def run_f0_grad(lib, ctl_amp, ctl_hd, f0, g, sample_rate, linear):
  ctl_amp, ctl_hd, f0, g = emu.f32(ctl_amp), emu.f32(ctl_hd), emu.f32(f0), emu.f32(g)
  b, f, k = ctl_hd.shape
  n = g.shape[1]
  nbytes = lib.ddsp_harmonic_f0_grad_workspace_bytes(b, f, k, n)
  ws = np.zeros(nbytes // 8 + 2, np.float64)            # 16-byte aligned start is not guaranteed by numpy: pick one
  off = (-ws.ctypes.data) % 16
  grad = np.full((b, f, 1), np.nan, np.float32)
  rc = lib.ddsp_harmonic_f0_grad_f32(emu.ptr(ctl_amp), emu.ptr(ctl_hd), emu.ptr(f0), emu.ptr(g), emu.ptr(grad),
                                     ws.ctypes.data + off, nbytes, b, f, k, n, sample_rate,
                                     _lib.HARM_AMP_LINEAR if linear else 0, None)
  return rc, grad


@pytest.mark.parametrize('method', ['window', 'linear'])
@pytest.mark.parametrize('b,f,k,hop,sr,f_lo,f_hi', [
    (2, 12, 8, 64, 16000, 100.0, 400.0),
    (1, 9, 20, 32, 16000, 300.0, 1200.0),       # harmonics cross Nyquist inside frames
    (2, 6, 5, 50, 8000, 60.0, 90.0),
])
def test_f0_grad_matches_analytic_oracle(lib, method, b, f, k, hop, sr, f_lo, f_hi):
  rng = np.random.default_rng(k * 7 + hop)
  n = f * hop
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = rng.uniform(f_lo, f_hi, (b, f, 1)).astype(np.float32)
  g = rng.standard_normal((b, n)).astype(np.float32)
  ctl = oracle.harmonic_get_controls(amps, hd, f0, sample_rate=sr, dtype=np.float64)
  rc, grad = run_f0_grad(lib, ctl['amplitudes'], ctl['harmonic_distribution'], f0, g, sr, method == 'linear')
  assert rc == 0
  ref = oracle.harmonic_backward(amps, hd, f0, g, n_samples=n, sample_rate=sr, amp_resample_method=method,
                                 with_f0=True)[2]
  scale = np.abs(ref).max()
  assert scale > 0
  np.testing.assert_allclose(grad, ref, rtol=0, atol=2e-4 * scale)


def test_f0_grad_argument_checks(lib):
  z = np.zeros(64, np.float32)
  ws = np.zeros(4096, np.float64)
  assert lib.ddsp_harmonic_f0_grad_workspace_bytes(2, 4, 3, 64) == 2 * 4 * 8 + 2 * 4 * 3 * 8 + 2 * 64 * 4
  args = (emu.ptr(z), emu.ptr(z), emu.ptr(z), emu.ptr(z), emu.ptr(z))
  off = (-ws.ctypes.data) % 16
  assert lib.ddsp_harmonic_f0_grad_f32(*args, ws.ctypes.data + off, 16, 2, 4, 3, 64, 16000, 0, None) == -4
  assert lib.ddsp_harmonic_f0_grad_f32(*args, ws.ctypes.data + off, 4096, 2, 4, 3, 66, 16000, 0, None) == -3
  assert lib.ddsp_harmonic_f0_grad_f32(*args, None, 4096, 2, 4, 3, 64, 16000, 0, None) == -1


# ---- the same entry points against the golden vectors (reference source on the TF stand-in) ------------------
from conftest import load_golden   # noqa: E402


def test_resample_golden(lib):
  g = load_golden('resample_methods')
  seen = 0
  for key in g:
    parts = key.split('_')
    if parts[0] not in ('up', 'small', 'ragged', 'down', 'x4d') or len(parts) != 3:
      continue
    kind, method, add_endpoint = parts[0], parts[1], parts[2] == 'endpoint'
    src = {'up': g['x'], 'ragged': g['x'], 'small': g['x_small']}.get(kind)
    if kind == 'down':
      src = g['up_%s_%s' % (method, parts[2])]
    if kind == 'x4d':                                 # [B,F,n_freq,C]: the host layer flattens the last two axes
      src = g['x_4d'].reshape(2, 6, 6)
    n = g[key].shape[1]
    rc, out = run_resample(lib, src, n, method, add_endpoint)
    assert rc == 0, key
    np.testing.assert_allclose(out.reshape(g[key].shape), g[key], rtol=0, atol=2e-6 if method == 'window' else 0.0,
                               err_msg=key)
    seen += 1
  assert seen == 34


def test_fft_convolve_golden(lib):
  g = load_golden('fft_convolve_crops')
  for key, ir, delay in [('valid_d0', 'ir', 0), ('valid_d5', 'ir', 5), ('valid_auto', 'ir', -1), ('same_d40', 'ir', 40),
                         ('one_valid_d0', 'ir_one', 0), ('one_valid_auto', 'ir_one', -1)]:
    l = g[ir].shape[2]
    start = (l - 1) // 2 - 1 if delay < 0 else delay
    rc, out = run_fir(lib, g['audio'], g[ir], g[key].shape[1], start)
    assert rc == 0, key
    np.testing.assert_allclose(out, g[key], rtol=0, atol=2e-6, err_msg=key)


# ---- effects.ExpDecayReverb: impulse response and its gradient ---------------------------------------------
@pytest.mark.parametrize('b,l,scale', [(3, 1000, True), (1, 4801, True), (2, 257, False), (2, 1, True)])
def test_exp_decay_ir_and_backward(lib, b, l, scale):
  rng = np.random.default_rng(b * 100 + l)
  gain = rng.standard_normal((b, 1)).astype(np.float32) + (0.0 if scale else 2.0)
  decay = rng.uniform(-1.0, 3.0, (b, 1)).astype(np.float32)
  noise = rng.uniform(-1.0, 1.0, (1, l)).astype(np.float32)
  ir = np.full((b, l), np.nan, np.float32)
  flags = _lib.DECAY_SCALE_EXP_SIGMOID if scale else 0
  assert lib.ddsp_exp_decay_ir_f32(emu.ptr(gain), emu.ptr(decay), emu.ptr(noise), emu.ptr(ir), b, l, flags, None) == 0
  scale_fn = oracle.exp_sigmoid if scale else None
  ref = oracle.exp_decay_ir(gain, decay, noise, scale_fn, dtype=np.float64)
  np.testing.assert_allclose(ir, ref, rtol=2e-5, atol=1e-7)
  g = rng.standard_normal((b, l)).astype(np.float32)
  nbytes = lib.ddsp_exp_decay_ir_backward_workspace_bytes(b, l)
  ws = np.zeros(nbytes // 8 + 2, np.float64)
  off = (-ws.ctypes.data) % 16
  gg, gd = np.full((b, 1), np.nan, np.float32), np.full((b, 1), np.nan, np.float32)
  rc = lib.ddsp_exp_decay_ir_backward_f32(emu.ptr(gain), emu.ptr(decay), emu.ptr(noise), emu.ptr(g), emu.ptr(gg),
                                          emu.ptr(gd), ws.ctypes.data + off, nbytes, b, l, flags, None)
  assert rc == 0
  ref_g, ref_d = oracle.exp_decay_ir_backward(gain, decay, noise, g, scale_fn)
  np.testing.assert_allclose(gg, ref_g, rtol=1e-4, atol=1e-5 * np.abs(ref_g).max() + 1e-12)
  np.testing.assert_allclose(gd, ref_d, rtol=1e-4, atol=1e-5 * np.abs(ref_d).max() + 1e-12)


@pytest.mark.parametrize('name', ['exp_decay_reverb_b3', 'exp_decay_reverb_trainable'])
def test_exp_decay_ir_golden(lib, name):
  g = load_golden(name)
  gain, decay, noise = emu.f32(g['gain'].reshape(-1)), emu.f32(g['decay'].reshape(-1)), emu.f32(g['noise'])
  b, l = gain.size, noise.shape[1]
  ir = np.full((b, l), np.nan, np.float32)
  assert lib.ddsp_exp_decay_ir_f32(emu.ptr(gain), emu.ptr(decay), emu.ptr(noise), emu.ptr(ir), b, l,
                                   _lib.DECAY_SCALE_EXP_SIGMOID, None) == 0
  np.testing.assert_allclose(np.broadcast_to(ir, g['ir'].shape), g['ir'], rtol=2e-5, atol=1e-7)


# ---- processors.Mix ------------------------------------------------------------------------------------------
def test_sigmoid_and_mix_kernels(lib):
  rng = np.random.default_rng(6)
  x = np.concatenate([rng.standard_normal(1000) * 5, [-100.0, 100.0, 0.0]]).astype(np.float32)
  y = np.full_like(x, np.nan)
  assert lib.ddsp_sigmoid_f32(emu.ptr(x), emu.ptr(y), x.size, None) == 0
  np.testing.assert_allclose(y, oracle.sigmoid(x.astype(np.float64)), rtol=3e-7, atol=1e-38)
  rows, c = 2 * 100, 3
  one, two = rng.standard_normal((rows, c)).astype(np.float32), rng.standard_normal((rows, c)).astype(np.float32)
  m = rng.uniform(0.0, 1.0, rows).astype(np.float32)
  out = np.full((rows, c), np.nan, np.float32)
  assert lib.ddsp_mix_f32(emu.ptr(one), emu.ptr(two), emu.ptr(m), emu.ptr(out), rows, c, None) == 0
  ref = np.sqrt(np.abs(m))[:, None] * one + (np.float32(1.0) - np.sqrt(np.abs(m - np.float32(1.0))))[:, None] * two
  np.testing.assert_array_equal(out, ref)                             # processors.py:231-233, fp32 op for op
  # the adjoints (round 6): against the analytic derivatives in fp64
  g = rng.standard_normal((rows, c)).astype(np.float32)
  g1, g2, gl = (np.full((rows, c), np.nan, np.float32), np.full((rows, c), np.nan, np.float32), np.full(rows, np.nan, np.float32))
  assert lib.ddsp_mix_backward_f32(emu.ptr(one), emu.ptr(two), emu.ptr(m), emu.ptr(g), emu.ptr(g1), emu.ptr(g2), emu.ptr(gl), rows, c,
                                   None) == 0
  m64 = m.astype(np.float64)
  np.testing.assert_allclose(g1, np.sqrt(m64)[:, None] * g, rtol=2e-7)
  np.testing.assert_allclose(g2, (1.0 - np.sqrt(1.0 - m64))[:, None] * g, rtol=2e-6, atol=1e-7)
  ref_l = (g * (one * (0.5 / np.sqrt(m64))[:, None] + two * (0.5 / np.sqrt(1.0 - m64))[:, None])).sum(axis=1)
  np.testing.assert_allclose(gl, ref_l, rtol=2e-5, atol=1e-5 * np.abs(ref_l).max())
  gl_only = np.full(rows, np.nan, np.float32)                          # any output may be left out
  assert lib.ddsp_mix_backward_f32(emu.ptr(one), emu.ptr(two), emu.ptr(m), emu.ptr(g), None, None, emu.ptr(gl_only), rows, c, None) == 0
  np.testing.assert_array_equal(gl_only, gl)
  gx = np.full_like(x, np.nan)
  gy = rng.standard_normal(x.size).astype(np.float32)
  assert lib.ddsp_sigmoid_backward_f32(emu.ptr(x), emu.ptr(gy), emu.ptr(gx), x.size, None) == 0
  s64 = oracle.sigmoid(x.astype(np.float64))
  e64 = np.exp(-np.abs(x.astype(np.float64)))
  np.testing.assert_allclose(gx, gy * e64 / (1.0 + e64) ** 2, rtol=2e-6, atol=1e-44)
