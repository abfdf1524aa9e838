"""CPU checks of the PYTHON GLUE in ddsp_amd/core.py / synths.py around the general-shape entry points:
argument order of the ctypes calls, output shapes, routing between the fast and the general entries, the
autograd node's f0 branch.  The host layer is pointed at host memory for the duration of a test: tensors
stay on the CPU, the general kernels come from the host build of general.hip (tests/hip_emu), and the two
entries that build cannot provide (harmonic controls, oscillator bank) are TEST DOUBLES that fill their
output buffers from the oracle.  Nothing here says anything about the GPU kernels themselves; the product
never runs this way (ddsp_amd has no CPU path: see test_host_api.test_no_gpu_fails_loudly_not_silently)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden
from ddsp_amd import _lib, core, synths
from oracle import ddsp_oracle as O
from tests.hip_emu import emu


def _view(ptr, shape, dtype=np.float32):
  n = int(np.prod(shape))
  ctype = ctypes.c_float if dtype == np.float32 else ctypes.c_double
  return np.ctypeslib.as_array((ctype * n).from_address(ptr)).reshape(shape)


class _HostLib:
  """emu entry points + oracle-backed doubles, behind the attribute names core.py calls."""

  def __init__(self):
    self._emu = emu.load()
    self.calls = []

  def __getattr__(self, name):
    if name in emu.ENTRY_POINTS:
      fn = getattr(self._emu, name)

      def traced(*args):
        self.calls.append(name)
        return fn(*args)
      return traced
    raise AttributeError('no host stand-in for ' + name)

  def ddsp_harmonic_controls_f32(self, amp, hd, f0, ctl_amp, ctl_hd, b, f, k, sample_rate, flags, stream):
    self.calls.append('ddsp_harmonic_controls_f32')
    scale = O.exp_sigmoid if flags & _lib.HARM_SCALE_EXP_SIGMOID else None
    c = O.harmonic_get_controls(_view(amp, (b, f, 1)), _view(hd, (b, f, k)), _view(f0, (b, f, 1)), sample_rate,
                                scale, bool(flags & _lib.HARM_NORMALIZE_NYQUIST))
    _view(ctl_amp, (b, f, 1))[:] = c['amplitudes']
    _view(ctl_hd, (b, f, k))[:] = c['harmonic_distribution']
    return 0

  def ddsp_oscillator_bank_workspace_bytes(self, b, n, k):
    return 64

  def ddsp_oscillator_bank_f32(self, freq, amp, out, ws, ws_bytes, b, n, k, sample_rate, sum_sinusoids, stream):
    self.calls.append('ddsp_oscillator_bank_f32')
    audio = O.oscillator_bank(_view(freq, (b, n, k)).astype(np.float64), _view(amp, (b, n, k)).astype(np.float64),
                              sample_rate, sum_sinusoids=bool(sum_sinusoids))
    _view(out, audio.shape)[:] = audio
    return 0

  def ddsp_fft_convolve_same_f32(self, audio, ir, out, b, bir, f, l, n, delay, stream):
    self.calls.append('ddsp_fft_convolve_same_f32')
    start = (l - 1) // 2 - 1 if delay < 0 else delay
    return self._emu.ddsp_fft_convolve_f32(audio, ir, out, b, bir, f, l, n, n, max(start, 0), stream)

  def ddsp_uniform_noise_f32(self, out, b, n, seed, batch_offset, stream):
    self.calls.append('ddsp_uniform_noise_f32')
    _view(out, (b, n))[:] = O.device_uniform_noise(b, n, seed=seed, batch_offset=batch_offset, noise_bits=11)
    return 0

  def ddsp_uniform_noise_ex_f32(self, out, b, n, seed, batch_offset, noise_bits, stream):
    self.calls.append('ddsp_uniform_noise_f32')
    _view(out, (b, n))[:] = O.device_uniform_noise(b, n, seed=seed, batch_offset=batch_offset, noise_bits=noise_bits)
    return 0

  def ddsp_fft_convolve_long_ex_workspace_bytes(self, b, bir, n, l, n_out, delay):
    return 64

  def ddsp_fft_convolve_long_ex_f32(self, audio, ir, out, ws, ws_bytes, b, bir, n, l, n_out, delay, flags, stream):
    """out[b][k] = y[k + delay], y = the full linear convolution (include/ddsp_amd.h), in fp64."""
    self.calls.append('ddsp_fft_convolve_long_ex_f32')
    a = _view(audio, (b, n)).astype(np.float64)
    h = np.broadcast_to(_view(ir, (bir, l)).astype(np.float64), (b, l)).copy()
    if flags & _lib.CONV_REVERSE_AUDIO:
      a = a[:, ::-1]
    if flags & _lib.CONV_REVERSE_IR:
      h = h[:, ::-1]
    if flags & _lib.CONV_MASK_TAP0:
      h[:, 0] = 0.0
    o = _view(out, (b, n_out))
    for row in range(b):
      y = np.concatenate([np.convolve(a[row], h[row]), np.zeros(delay + n_out)])[delay:delay + n_out]
      if flags & _lib.CONV_ADD_DRY:
        y = y + a[row]
      if flags & _lib.CONV_ZERO_OUT0:
        y[0] = 0.0
      o[row] = y[::-1] if flags & _lib.CONV_REVERSE_OUT else y
    return 0

  def ddsp_sum_rows_f32(self, x, out, b, l, zero_first, stream):
    self.calls.append('ddsp_sum_rows_f32')
    v = _view(x, (b, l)).astype(np.float64).sum(axis=0)
    if zero_first:
      v[0] = 0.0
    _view(out, (1, l))[:] = v
    return 0

  def ddsp_resample_f32(self, x, out, b, f, n, c, window, stream):
    self.calls.append('ddsp_resample_f32')
    return self._emu.ddsp_resample_ex_f32(x, out, b, f, n, c, 3 if window else 1, 1, stream)


@pytest.fixture
def host(monkeypatch):
  lib = _HostLib()
  monkeypatch.setattr(_lib, 'load', lambda: lib)
  monkeypatch.setattr(core, '_device', lambda: torch.device('cpu'))
  monkeypatch.setattr(core, '_stream', lambda: None)
  monkeypatch.setattr(core, '_ws_bytes_cache', {})
  return lib


def npy(t):
  return t.detach().numpy()


def test_resample_glue_every_method(host):
  g = load_golden('resample_methods')
  for key in g:
    parts = key.split('_')
    if parts[0] not in ('up', 'small', 'ragged', 'down', 'x4d') or len(parts) != 3:
      continue
    kind, method, add_endpoint = parts[0], parts[1], parts[2] == 'endpoint'
    src = {'up': g['x'], 'ragged': g['x'], 'small': g['x_small'], 'x4d': g['x_4d']}.get(kind)
    if kind == 'down':
      src = g['up_%s_%s' % (method, parts[2])]
    host.calls.clear()
    out = npy(core.resample(src, g[key].shape[1], method=method, add_endpoint=add_endpoint))
    assert out.shape == g[key].shape, key
    np.testing.assert_allclose(out, g[key], rtol=0, atol=2e-6 if method == 'window' else 0.0, err_msg=key)
    fast = add_endpoint and method in ('linear', 'window')
    assert host.calls == ['ddsp_resample_f32' if fast else 'ddsp_resample_ex_f32'], key
  # 1-D and 2-D inputs come back 1-D and 2-D (core_test.py:153-173)
  assert tuple(core.resample(np.ones(5, np.float32), 160, method='cubic').shape) == (160,)
  assert tuple(core.resample(np.ones((5, 5), np.float32), 160, method='nearest', add_endpoint=False).shape) == (5, 160)
  with pytest.raises(ValueError, match='3 dimensions'):
    core.resample(np.ones((5, 5, 5, 5), np.float32), 160, method='window')
  with pytest.raises(ValueError, match='is invalid'):
    core.resample(np.ones((1, 5, 1), np.float32), 160, method='bogus')
  with pytest.raises(ValueError, match=r'frames - 1'):
    core.upsample_with_windows(np.ones((1, 5, 1), np.float32), 161, add_endpoint=False)


def test_fft_convolve_glue_crops_and_routing(host):
  g = load_golden('fft_convolve_crops')
  for key, ir, padding, delay, entry in [
      ('valid_d0', 'ir', 'valid', 0, 'ddsp_fft_convolve_f32'), ('valid_d5', 'ir', 'valid', 5, 'ddsp_fft_convolve_f32'),
      ('valid_auto', 'ir', 'valid', -1, 'ddsp_fft_convolve_f32'), ('same_d40', 'ir', 'same', 40, 'ddsp_fft_convolve_same_f32'),
      ('one_valid_d0', 'ir_one', 'valid', 0, 'ddsp_fft_convolve_f32'),
      ('one_valid_auto', 'ir_one', 'valid', -1, 'ddsp_fft_convolve_f32')]:
    host.calls.clear()
    out = npy(core.fft_convolve(g['audio'], g[ir], padding=padding, delay_compensation=delay))
    assert out.shape == g[key].shape, key
    np.testing.assert_allclose(out, g[key], rtol=0, atol=2e-6, err_msg=key)
    assert host.calls == [entry], key
  host.calls.clear()
  empty = core.fft_convolve(np.ones((1, 250), np.float32), np.ones((1, 25, 7), np.float32), padding='valid',
                            delay_compensation=0)
  assert tuple(empty.shape) == (1, 0) and host.calls == []
  with pytest.raises(ValueError, match='Padding'):
    core.fft_convolve(g['audio'], g['ir'], padding='bogus')
  with pytest.raises(ValueError, match='do not match'):
    core.fft_convolve(np.ones((1, 100), np.float32), np.ones((1, 30, 5), np.float32), padding='valid')


@pytest.mark.parametrize('name', ['synthesis_shifts_window', 'synthesis_shifts_only', 'synthesis_cubic',
                                  'synthesis_nearest_angular', 'synthesis_linear_ragged'])
def test_harmonic_synthesis_glue_materialised_chain(host, name):
  g = load_golden(name)
  out = npy(core.harmonic_synthesis(
      frequencies=g['f0_hz'], amplitudes=g['amplitudes'], harmonic_shifts=g.get('harmonic_shifts'),
      harmonic_distribution=g.get('harmonic_distribution'), n_samples=int(g['n_samples']),
      sample_rate=int(g['sample_rate']), amp_resample_method=str(g['amp_method']),
      use_angular_cumsum=bool(g['angular'])))
  assert out.shape == g['audio'].shape
  amp_sum = float(np.abs(g['amplitudes']).max()) * (float(np.abs(g['harmonic_distribution']).sum(-1).max())
                                                     if 'harmonic_distribution' in g else g['harmonic_shifts'].shape[-1])
  np.testing.assert_allclose(out, g['audio'], rtol=0, atol=2e-3 * max(1.0, amp_sum))
  assert host.calls[0] == 'ddsp_harmonic_envelopes_f32' and host.calls[-1] == 'ddsp_oscillator_bank_f32'
  assert len(host.calls) == 4                                 # envelopes, two resamples, oscillator bank


def test_harmonic_processor_glue_cubic_envelope(host):
  g = load_golden('harmonic_cubic_amp')
  synth = synths.Harmonic(n_samples=int(g['n_samples']), sample_rate=int(g['sample_rate']),
                          amp_resample_method='cubic')
  out = synth(g['amplitudes'], g['harmonic_distribution'], g['f0_hz'], return_outputs_dict=True)
  np.testing.assert_allclose(npy(out['controls']['harmonic_distribution']), g['ctl_harmonic_distribution'],
                             rtol=2e-5, atol=1e-9)
  np.testing.assert_allclose(npy(out['signal']), g['signal'], rtol=0, atol=2e-3)


def test_harmonic_f0_gradient_glue(host):
  rng = np.random.default_rng(9)
  b, f, k, hop, sr = 2, 12, 8, 64, 16000
  n = f * hop
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = rng.uniform(100.0, 400.0, (b, f, 1)).astype(np.float32)
  g = rng.standard_normal((b, n)).astype(np.float32)
  for method in ('window', 'linear'):
    synth = synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)
    grad_f0 = npy(synth._backward_f0(torch.tensor(amps), torch.tensor(hd), torch.tensor(f0), True, torch.tensor(g)))
    ref = O.harmonic_backward(amps, hd, f0, g, n_samples=n, sample_rate=sr, amp_resample_method=method,
                              with_f0=True)[2]
    assert grad_f0.shape == (b, f, 1)
    np.testing.assert_allclose(grad_f0, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
  # the autograd node routes needs_input_grad[2] to it and returns None for the inputs that need nothing
  class _Ctx:
    needs_input_grad = (False, False, True, False, False, False)
    saved_tensors = (torch.tensor(amps), torch.tensor(hd), torch.tensor(f0))
    fuse = True
  _Ctx.synth = synth
  grads = synths._HarmonicFunction.backward(_Ctx, torch.tensor(g), None, None)
  assert grads[0] is None and grads[1] is None and grads[3:] == (None, None, None)
  np.testing.assert_allclose(npy(grads[2]), ref, rtol=0, atol=2e-4 * np.abs(ref).max())


# ---- effects.ExpDecayReverb (ddsp/effects.py:120-199; effects_test.py:96-110) --------------------------------
def test_exp_decay_reverb_glue(host):
  from ddsp_amd import effects
  rng = np.random.default_rng(4)
  b, n, l = 3, 400, 100
  audio = rng.standard_normal((b, n)).astype(np.float32)
  gain = rng.standard_normal((b, 1)).astype(np.float32)
  decay = rng.uniform(-1.0, 2.0, (b, 1)).astype(np.float32)
  noise = rng.uniform(-1.0, 1.0, (1, l)).astype(np.float32)
  rev = effects.ExpDecayReverb(trainable=False, reverb_length=l)
  with pytest.raises(ValueError, match='gain'):                      # test_non_trainable_raises_value_error
    rev(audio)
  out = rev(audio, gain, decay, noise=noise, return_outputs_dict=True)
  assert sorted(out['controls']) == ['audio', 'ir']                  # test_get_controls_returns_correct_keys
  ir_ref = O.exp_decay_ir(gain, decay, noise, dtype=np.float64)
  np.testing.assert_allclose(npy(out['controls']['ir']), ir_ref, rtol=2e-5, atol=1e-7)
  np.testing.assert_allclose(npy(out['signal']), O.reverb(audio, ir_ref, add_dry=True, dtype=np.float64),
                             rtol=0, atol=2e-5)
  # generated burst: one Philox row of reverb_length samples, a new one per call
  host.calls.clear()
  y1, y2 = rev(audio, gain, decay), rev(audio, gain, decay)
  assert tuple(y1.shape) == (b, n) and host.calls.count('ddsp_uniform_noise_f32') == 2
  assert float((y1 - y2).abs().max()) > 0
  burst = O.device_uniform_noise(1, l, seed=0 | (0 << 32))
  ir0 = O.exp_decay_ir(gain, decay, burst, dtype=np.float64)
  np.testing.assert_allclose(npy(y1), O.reverb(audio, ir0, add_dry=True, dtype=np.float64), rtol=0, atol=2e-5)
  # trainable: gain 2.0 / decay 4.0 (effects.py:158-168), one IR tiled over the batch
  trev = effects.ExpDecayReverb(trainable=True, reverb_length=l, add_dry=False)
  ctl = trev.get_controls(torch.tensor(audio), noise=noise)
  assert tuple(ctl['ir'].shape) == (1, l) and float(trev._gain) == 2.0 and float(trev._decay) == 4.0
  ir_t = O.exp_decay_ir(np.full((1, 1), 2.0), np.full((1, 1), 4.0), noise, dtype=np.float64)
  np.testing.assert_allclose(npy(trev.get_signal(**ctl)), O.reverb(audio, ir_t, add_dry=False, dtype=np.float64),
                             rtol=0, atol=2e-5)
  # gradients with respect to gain and decay, through the reverb's own autograd node
  tg = torch.tensor(gain, requires_grad=True)
  td = torch.tensor(decay, requires_grad=True)
  g_out = rng.standard_normal((b, n)).astype(np.float32)
  rev(audio, tg, td, noise=noise).backward(torch.tensor(g_out))
  g_ir = O.reverb_backward(audio, ir_ref, g_out, add_dry=True)[1]
  ref_g, ref_d = O.exp_decay_ir_backward(gain, decay, noise, g_ir)
  np.testing.assert_allclose(npy(tg.grad), ref_g, rtol=1e-3, atol=1e-4 * np.abs(ref_g).max())
  np.testing.assert_allclose(npy(td.grad), ref_d, rtol=1e-3, atol=1e-4 * np.abs(ref_d).max())


# ---- processors.Mix, synths.TensorToAudio (processors_test.py:103-114, synths.py:23-52) --------------------------
def test_mix_and_tensor_to_audio_glue(host):
  from ddsp_amd import processors
  x1 = np.zeros((2, 100, 3), np.float32) + 1.0
  x2 = np.zeros((2, 100, 3), np.float32) + 2.0
  level = np.zeros((2, 100, 1), np.float32) + 0.1                     # will be passed to sigmoid
  out = processors.Mix(name='mix')(x1, x2, level)
  assert list(out.shape) == [2, 100, 3]                               # MixTest.test_output_shape_is_correct
  m = 1.0 / (1.0 + np.exp(-0.1))
  np.testing.assert_allclose(npy(out), np.sqrt(m) * 1.0 + (1.0 - np.sqrt(1.0 - m)) * 2.0, rtol=1e-6)
  # a frame-rate mix level is resampled to the signals' length (processors.py:208), 2-D signals come back 2-D
  rng = np.random.default_rng(2)
  s1, s2 = rng.standard_normal((2, 64)).astype(np.float32), rng.standard_normal((2, 64)).astype(np.float32)
  coarse = rng.standard_normal((2, 8, 1)).astype(np.float32)
  ctl = processors.Mix().get_controls(s1, s2, coarse)
  ml = O.resample(O.sigmoid(coarse), 64)
  np.testing.assert_allclose(npy(ctl['mix_level']), ml, rtol=1e-6, atol=1e-7)
  out2 = npy(processors.Mix().get_signal(**ctl))
  ref2 = np.sqrt(np.abs(ml[:, :, 0])) * s1 + (1.0 - np.sqrt(np.abs(ml[:, :, 0] - 1.0))) * s2
  assert out2.shape == (2, 64)
  np.testing.assert_allclose(out2, ref2, rtol=1e-5, atol=1e-6)
  with pytest.raises(ValueError, match='same length'):
    processors.Mix()(np.zeros((2, 100, 3), np.float32), np.zeros((2, 90, 3), np.float32), level)
  # with a mix level on the autograd tape the same numbers come out of the torch path
  t = torch.tensor(coarse, requires_grad=True)
  out3 = processors.Mix()(s1, s2, t)
  np.testing.assert_allclose(npy(out3), ref2, rtol=1e-5, atol=1e-6)
  wgt = rng.standard_normal((2, 64)).astype(np.float32)
  (out3 * torch.tensor(wgt)).sum().backward()
  assert t.grad is not None and float(t.grad.abs().max()) > 0

  def chain(cc):                                                      # the same chain in fp64 numpy (processors.py:207-233)
    mm = O.resample(O.sigmoid(cc.astype(np.float64)), 64, dtype=np.float64)[:, :, 0]
    return float(((np.sqrt(np.abs(mm)) * s1 + (1.0 - np.sqrt(np.abs(mm - 1.0))) * s2) * wgt).sum())
  fd = np.zeros_like(coarse, dtype=np.float64)
  for idx in np.ndindex(coarse.shape):
    d = np.zeros_like(coarse, dtype=np.float64); d[idx] = 1e-4
    fd[idx] = (chain(coarse + d) - chain(coarse - d)) / 2e-4
  np.testing.assert_allclose(npy(t.grad), fd, rtol=2e-3, atol=2e-4 * np.abs(fd).max())
  # ... and of the two signals (Mix.get_signal on its own node)
  t1 = torch.tensor(s1, requires_grad=True)
  processors.Mix().get_signal(t1, s2, ctl['mix_level']).backward(torch.tensor(wgt))
  np.testing.assert_allclose(npy(t1.grad), np.sqrt(np.abs(ml[:, :, 0])) * wgt, rtol=1e-5, atol=1e-7)
  assert {'ddsp_mix_backward_f32', 'ddsp_sigmoid_backward_f32', 'ddsp_resample_ex_backward_f32'} <= set(host.calls)
  samples = rng.standard_normal((2, 50, 1)).astype(np.float32)
  np.testing.assert_array_equal(npy(synths.TensorToAudio()(samples)), samples[:, :, 0])
