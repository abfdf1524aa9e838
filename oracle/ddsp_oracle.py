"""CPU oracle for the DDSP additive-synthesis hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement, op for op, of the reference's
`ddsp/core.py`, `ddsp/synths.py` and `ddsp/processors.py` for the one path this
repository accelerates (synths.Harmonic + synths.FilteredNoise + processors.Add),
plus `ddsp/effects.py:27-117` (effects.Reverb) and the forward pass of
`ddsp/losses.py:131-243` (losses.SpectralLoss, rank 2).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product package `ddsp_amd` never does (it fails loudly when
the HIP library is missing).

Pinning status
--------------
* The reference itself cannot run here (TensorFlow / gin are not installed), so
  there is no output of real TensorFlow to compare with.
* What IS pinned: `tests/golden/*.npz` are produced by executing the reference's
  own `ddsp/core.py` / `ddsp/synths.py` / `ddsp/effects.py` source files, unmodified, on top of a
  numpy stand-in for the dozen TensorFlow ops they call
  (`tests/golden/make_golden.py`, `tests/golden/tf_numpy_shim.py`).  Those vectors
  pin this restatement's index math, crops, windows, masks and op order against
  the reference code; the TF op semantics themselves come from SURVEY.md
  Appendix A (TF <= 2.11 behaviour) and are restated in the shim and here.
* The reference's own known-answer tests for the path (fft_convolve vs scipy,
  group-delay identity, IR sizes, Nyquist silence, resample sub-sampling,
  ValueErrors) are re-expressed in `tests/test_oracle.py`.
* Oscillator numerics vs real TensorFlow: PARITY UNPINNED (the reference's three
  "is_accurate" oscillator tests compare empty slices).

Two arithmetic modes share one code path (`dtype` argument):
  np.float32  "faithful": same op order TF executes, sequential cumsum, no FMA.
  np.float64  "truth":    same formulas in double precision.
All tensors are row-major [batch, time, channel], as in the reference.
"""

import contextlib

import numpy as np

TWO_PI = 2.0 * np.pi


# ----------------------------------------------------------------------------
# Utilities / scaling   (ddsp/core.py:31-36, 207-210, 386-404)
# ----------------------------------------------------------------------------
def as_float(x, dtype=np.float32):
  """core.tf_float32 (core.py:31-36): convert anything to a float array."""
  return np.asarray(x, dtype=dtype)


def safe_divide(numerator, denominator, eps=1e-7):
  """core.safe_divide (core.py:207-210): zero denominators become eps."""
  dt = numerator.dtype
  safe_denominator = np.where(denominator == 0.0, dt.type(eps), denominator)
  return numerator / safe_denominator


def sigmoid(x):
  """tf.nn.sigmoid: 1/(1+exp(-x)), computed in x's dtype (stable both tails)."""
  dt = x.dtype
  e = np.exp(-np.abs(x))
  return np.where(x >= 0, dt.type(1) / (dt.type(1) + e), e / (dt.type(1) + e))


def exp_sigmoid(x, exponent=10.0, max_value=2.0, threshold=1e-7, dtype=np.float32):
  """core.exp_sigmoid (core.py:386-404): max_value*sigmoid(x)**log(exponent)+thr."""
  x = as_float(x, dtype)
  dt = x.dtype.type
  return dt(max_value) * sigmoid(x)**dt(np.log(exponent)) + dt(threshold)


# ----------------------------------------------------------------------------
# Resampling   (ddsp/core.py:573-714)
# ----------------------------------------------------------------------------
_EXACT_POSITIONS = [False]


@contextlib.contextmanager
def exact_resize_positions():
  """Inside this context the BILINEAR legacy resize (what core.resample(..., 'linear') and therefore every frequency envelope
  goes through) takes its source coordinate t n_in / n_out in exact arithmetic instead of TF's fp32 product with an fp32 scale
  (up to pos 2^-23 frames off when n_out / n_in is not a power of two).  For checkers that want to know how far the reference's
  own position rounding moves a result - tools/fuzz_parity.py's streaming family allows the kernels, which take r / hop on clips
  that are whole frames, that much (known limits).  Never the default: the fixtures pin TF's arithmetic."""
  _EXACT_POSITIONS[0] = True
  try:
    yield
  finally:
    _EXACT_POSITIONS[0] = False


def resize_bilinear_legacy(x, n_out, align_corners=False):
  """tf.compat.v1.image.resize(BILINEAR) on the time axis of x[B, F, C].

  Legacy kernel, no half-pixel centres: scale is computed
  in fp32, pos = t*scale (fp32), lo=floor, hi=min(ceil, F-1), lerp=pos-lo,
  out = top + (bottom - top) * lerp.  Index math is fp32 in both modes (that is
  what TF does); only the value arithmetic follows x.dtype.
  """
  dt = x.dtype.type
  n_in = x.shape[1]
  if _EXACT_POSITIONS[0]:
    # (checkers only, see exact_resize_positions below: t n_in / n_out in fp64 instead of TF's fl32(t fl32(n_in / n_out)))
    num, den = (n_in - 1, n_out - 1) if (align_corners and n_out > 1) else (n_in, n_out)
    pos = np.arange(n_out, dtype=np.float64) * num / den
    lo = np.floor(pos)
    hi = np.minimum(np.ceil(pos), float(n_in - 1))
    lo_i, hi_i = lo.astype(np.int64), hi.astype(np.int64)
    top, bottom = x[:, lo_i, :], x[:, hi_i, :]
    return top + (bottom - top) * (pos - lo).astype(x.dtype)[None, :, None]
  if align_corners and n_out > 1:
    scale = np.float32(n_in - 1) / np.float32(n_out - 1)
  else:
    scale = np.float32(n_in) / np.float32(n_out)
  pos = np.arange(n_out, dtype=np.float32) * scale            # fp32 product
  lo = np.floor(pos)
  hi = np.minimum(np.ceil(pos), np.float32(n_in - 1))
  lerp = (pos - lo).astype(np.float32)
  lo_i, hi_i = lo.astype(np.int64), hi.astype(np.int64)
  top, bottom = x[:, lo_i, :], x[:, hi_i, :]
  return top + (bottom - top) * lerp.astype(x.dtype)[None, :, None].astype(dt)


def _legacy_resize_positions(n_in, n_out, align_corners):
  """Source coordinate of every output index, fp32: out * scale (LegacyScaler), scale as
  CalculateResizeScale computes it (TF <= 2.11 image_resizer_state.h)."""
  if align_corners and n_out > 1:
    scale = np.float32(n_in - 1) / np.float32(n_out - 1)
  else:
    scale = np.float32(n_in) / np.float32(n_out)
  return np.arange(n_out, dtype=np.float32) * scale


def resize_nearest_legacy(x, n_out, align_corners=False):
  """tf.compat.v1.image.resize(NEAREST_NEIGHBOR) on the time axis of x[B, F, C] (legacy kernel):
  source = min(floor(pos), F-1), or min(round(pos), F-1) with align_corners (roundf: half away from 0)."""
  n_in = x.shape[1]
  pos = _legacy_resize_positions(n_in, n_out, align_corners)
  # roundf(pos): half away from zero.  In fp64: pos + 0.5 is exact there - in fp32 the sum itself rounds (0.49999997 + 0.5 -> 1.0)
  # and floor() lands one frame too far (found by tools/fuzz_parity.py against the kernel's roundf: frames = 2, n = 111)
  src = np.floor(pos.astype(np.float64) + 0.5) if align_corners else np.floor(pos)
  return x[:, np.minimum(src.astype(np.int64), n_in - 1), :]


_CUBIC_TABLE_SIZE = 1024


def _cubic_coeffs_table():
  """TF's bicubic coefficient table, A = -0.75 (resize_bicubic_op.cc InitCoeffsTable): entry 2i is the
  |x| <= 1 polynomial at x = i/1024, entry 2i+1 the 1 <= |x| <= 2 polynomial at x + 1; evaluated in
  double on the float abscissa and stored as float."""
  a = -0.75
  x = (np.arange(_CUBIC_TABLE_SIZE + 1, dtype=np.float64) / _CUBIC_TABLE_SIZE).astype(np.float32)
  near = ((a + 2.0) * x.astype(np.float64) - (a + 3.0)) * x.astype(np.float64)**2 + 1.0
  x1 = (x + np.float32(1.0)).astype(np.float64)
  far = ((a * x1 - 5.0 * a) * x1 + 8.0 * a) * x1 - 4.0 * a
  return near.astype(np.float32), far.astype(np.float32)


def resize_bicubic_legacy(x, n_out, align_corners=False):
  """tf.compat.v1.image.resize(BICUBIC) on the time axis of x[B, F, C] (legacy kernel, no half-pixel
  centres): pos = t*scale, the fractional part quantised to 1/1024 (lrintf), weights from the
  coefficient table, the four source indices clamped to [0, F-1], v0 w0 + v1 w1 + v2 w2 + v3 w3
  summed left to right in x.dtype.  The width axis (1 -> 1) has weights (0, 1, 0, 0): a copy."""
  n_in = x.shape[1]
  near, far = _cubic_coeffs_table()
  pos = _legacy_resize_positions(n_in, n_out, align_corners)
  lo = np.floor(pos)
  offset = np.rint((pos - lo).astype(np.float32) * np.float32(_CUBIC_TABLE_SIZE)).astype(np.int64)
  src = lo.astype(np.int64)
  w = [far[offset], near[offset], near[_CUBIC_TABLE_SIZE - offset], far[_CUBIC_TABLE_SIZE - offset]]
  out = None
  for tap in range(4):
    idx = np.clip(src - 1 + tap, 0, n_in - 1)
    term = x[:, idx, :] * w[tap].astype(x.dtype)[None, :, None]
    out = term if out is None else out + term
  return out


def hann_window_periodic(n, dtype=np.float32):
  """tf.signal.hann_window(n) (periodic=True) as TensorFlow computes it (tensorflow/python/ops/signal/window_ops.py,
  _raised_cosine_window): [1.0] for n == 1; else 0.5 - 0.5 cos(2 pi i / d) with d = n + periodic * even - 1, even = 1 - n % 2 -
  the periodic window (d = n) for EVEN n only, the SYMMETRIC one (d = n - 1) for odd n (core.py:1505 with the constructor's
  window_size=257 on a longer response).  Rounds 1-5 divided by n whatever its parity (and so did the TF stand-in of
  tests/golden: a shared misreading no test could see)."""
  n = int(n)
  if n == 1:
    return np.ones(1, dtype)
  d = n - 1 + (1 - n % 2)
  if np.dtype(dtype) == np.float32:
    # the faithful mode: TensorFlow's own op order in fp32 - cos_arg = constant(2 pi) * count / n, then a - b * cos(cos_arg) -,
    # which is where the first samples of a long window lose their relative accuracy (0.5 - 0.5 cos(x), x -> 0: 2e-4 at
    # sample 17 of 4096 points; the yardstick tools/fuzz_parity.py holds loss values on very short clips to)
    count = np.arange(n, dtype=np.float32)
    cos_arg = (np.float32(TWO_PI) * count) / np.float32(d)
    return (np.float32(0.5) - np.float32(0.5) * np.cos(cos_arg, dtype=np.float32)).astype(np.float32)
  i = np.arange(n, dtype=np.float64)
  return (0.5 - 0.5 * np.cos(TWO_PI * i / d)).astype(dtype)


def overlap_and_add(frames, step):
  """tf.signal.overlap_and_add(frames[..., F, L], step) -> [..., (F-1)*step+L]."""
  n_frames, length = frames.shape[-2], frames.shape[-1]
  out = np.zeros(frames.shape[:-2] + ((n_frames - 1) * step + length,), frames.dtype)
  for f in range(n_frames):  # plain summation in frame order
    out[..., f * step:f * step + length] += frames[..., f, :]
  return out


def upsample_with_windows(inputs, n_timesteps, add_endpoint=True, dtype=np.float32):
  """core.upsample_with_windows (core.py:645-714), literal Hann overlap-add."""
  inputs = as_float(inputs, dtype)
  if inputs.ndim != 3:
    raise ValueError('Upsample_with_windows() only supports 3 dimensions, '
                     'not {}.'.format(inputs.shape))
  if add_endpoint:
    inputs = np.concatenate([inputs, inputs[:, -1:, :]], axis=1)
  n_frames = int(inputs.shape[1])
  n_intervals = n_frames - 1
  if n_frames >= n_timesteps:
    raise ValueError('Upsample with windows cannot be used for downsampling'
                     'More input frames ({}) than output timesteps ({})'.format(
                         n_frames, n_timesteps))
  if n_timesteps % n_intervals != 0.0:
    minus_one = '' if add_endpoint else ' - 1'
    raise ValueError(
        'For upsampling, the target the number of timesteps must be divisible '
        'by the number of input frames{}. (timesteps:{}, frames:{}, '
        'add_endpoint={}).'.format(minus_one, n_timesteps, n_frames, add_endpoint))
  hop_size = n_timesteps // n_intervals
  window = hann_window_periodic(2 * hop_size, dtype)
  x = np.transpose(inputs, (0, 2, 1))                # [B, C, F]
  x_windowed = x[:, :, :, None] * window[None, None, None, :]
  x = overlap_and_add(x_windowed, hop_size)
  x = np.transpose(x, (0, 2, 1))                     # [B, T, C]
  return x[:, hop_size:-hop_size, :]


def upsample_with_windows_closed_form(inputs, n_timesteps, dtype=np.float32):
  """Raised-cosine interpolation identical to upsample_with_windows(add_endpoint=True).

  out[t] = x[j]*w[hop+r] + x[j+1]*w[r], j=t//hop, r=t%hop, x[F]=x[F-1]
  (SURVEY F7a).  This is the form the HIP kernel evaluates.
  """
  x = as_float(inputs, dtype)
  n_frames = x.shape[1]
  hop = n_timesteps // n_frames
  w = hann_window_periodic(2 * hop, dtype)
  xe = np.concatenate([x, x[:, -1:, :]], axis=1)
  t = np.arange(n_timesteps)
  j, r = t // hop, t % hop
  return xe[:, j, :] * w[hop + r][None, :, None] + xe[:, j + 1, :] * w[r][None, :, None]


def resample(inputs, n_timesteps, method='linear', add_endpoint=True, dtype=np.float32):
  """core.resample (core.py:573-642) for 1-D..4-D inputs, every method."""
  inputs = as_float(inputs, dtype)
  is_1d, is_2d, is_4d = inputs.ndim == 1, inputs.ndim == 2, inputs.ndim == 4
  if is_1d:
    inputs = inputs[None, :, None]
  elif is_2d:
    inputs = inputs[:, :, None]
  elif is_4d:
    if method == 'window':
      raise ValueError('Upsample_with_windows() only supports 3 dimensions, '
                       'not {}.'.format(inputs.shape))
    shape_4d = inputs.shape                       # [B, F, n_freq, C]: the width axis is resized 1:1
    inputs = inputs.reshape(shape_4d[0], shape_4d[1], -1)
  if method == 'linear':
    outputs = resize_bilinear_legacy(inputs, n_timesteps, align_corners=not add_endpoint)
  elif method == 'window':
    outputs = upsample_with_windows(inputs, n_timesteps, add_endpoint, dtype)
  elif method == 'nearest':
    outputs = resize_nearest_legacy(inputs, n_timesteps, align_corners=not add_endpoint)
  elif method == 'cubic':
    outputs = resize_bicubic_legacy(inputs, n_timesteps, align_corners=not add_endpoint)
  else:
    raise ValueError('Method ({}) is invalid. Must be one of {}.'.format(
        method, "['nearest', 'linear', 'cubic', 'window']"))
  if is_1d:
    outputs = outputs[0, :, 0]
  elif is_2d:
    outputs = outputs[:, :, 0]
  elif is_4d:
    outputs = outputs.reshape(shape_4d[0], n_timesteps, shape_4d[2], shape_4d[3])
  return outputs


# ----------------------------------------------------------------------------
# Oscillator bank   (ddsp/core.py:800-962, 1028-1111)
# ----------------------------------------------------------------------------
def angular_cumsum(angular_frequency, chunk_size=1000):
  """core.angular_cumsum (core.py:800-866): chunked cumsum with mod-2pi stitching."""
  x = angular_frequency
  dt = x.dtype.type
  n_batch, n_time = x.shape[0], x.shape[1]
  ch_shape = x.shape[2:]
  remainder = n_time % chunk_size
  if remainder:
    pad = [(0, 0), (0, chunk_size - remainder)] + [(0, 0)] * len(ch_shape)
    x = np.pad(x, pad)
  length = x.shape[1]
  n_chunks = length // chunk_size
  chunks = x.reshape((n_batch, n_chunks, chunk_size) + ch_shape)
  phase = np.cumsum(chunks, axis=2, dtype=x.dtype)
  two_pi = dt(TWO_PI)
  offsets = np.mod(phase[:, :, -1:, ...], two_pi)
  offsets = np.concatenate([np.zeros_like(offsets[:, :1]), offsets], axis=1)[:, :-1]
  offsets = np.mod(np.cumsum(offsets, axis=1, dtype=x.dtype), two_pi)
  phase = phase + offsets
  phase = np.mod(phase, two_pi)
  phase = phase.reshape((n_batch, length) + ch_shape)
  if remainder:
    phase = phase[:, :n_time]
  return phase


def remove_above_nyquist(frequency_envelopes, amplitude_envelopes, sample_rate=16000):
  """core.remove_above_nyquist (core.py:869-891): amp=0 where f >= sr/2."""
  dt = amplitude_envelopes.dtype.type
  return np.where(frequency_envelopes >= dt(sample_rate / 2.0),
                  np.zeros_like(amplitude_envelopes), amplitude_envelopes)


def get_harmonic_frequencies(frequencies, n_harmonics):
  """core.get_harmonic_frequencies (core.py:1028-1045): f0 * [1..K]."""
  f_ratios = np.linspace(1.0, float(n_harmonics), int(n_harmonics)).astype(frequencies.dtype)
  return frequencies * f_ratios[None, None, :]


def normalize_harmonics(harmonic_distribution, f0_hz=None, sample_rate=None):
  """core.normalize_harmonics (core.py:894-907)."""
  if sample_rate is not None and f0_hz is not None:
    n_harmonics = int(harmonic_distribution.shape[-1])
    harmonic_frequencies = get_harmonic_frequencies(f0_hz, n_harmonics)
    harmonic_distribution = remove_above_nyquist(
        harmonic_frequencies, harmonic_distribution, sample_rate)
  return safe_divide(harmonic_distribution,
                     np.sum(harmonic_distribution, axis=-1, keepdims=True,
                            dtype=harmonic_distribution.dtype))


def oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=16000,
                    sum_sinusoids=True, use_angular_cumsum=False):
  """core.oscillator_bank (core.py:912-962). dtype follows the inputs."""
  dt = frequency_envelopes.dtype.type
  amplitude_envelopes = remove_above_nyquist(frequency_envelopes, amplitude_envelopes,
                                             sample_rate)
  omegas = frequency_envelopes * dt(TWO_PI)
  omegas = omegas / dt(float(sample_rate))
  if use_angular_cumsum:
    phases = angular_cumsum(omegas)
  else:
    # numpy's cumsum along a non-contiguous axis is a strict sequential scan in
    # the array dtype: the TF-CPU (Eigen scan) order assumed in SURVEY Appendix A.
    phases = np.cumsum(omegas, axis=1, dtype=omegas.dtype)
  wavs = np.sin(phases)
  audio = amplitude_envelopes * wavs
  if sum_sinusoids:
    audio = np.sum(audio, axis=-1, dtype=audio.dtype)
  return audio


def harmonic_synthesis(frequencies, amplitudes, harmonic_shifts=None,
                       harmonic_distribution=None, n_samples=64000, sample_rate=16000,
                       amp_resample_method='window', use_angular_cumsum=False,
                       dtype=np.float32):
  """core.harmonic_synthesis (core.py:1048-1111)."""
  frequencies = as_float(frequencies, dtype)
  amplitudes = as_float(amplitudes, dtype)
  if harmonic_distribution is not None:
    harmonic_distribution = as_float(harmonic_distribution, dtype)
    n_harmonics = int(harmonic_distribution.shape[-1])
  elif harmonic_shifts is not None:
    harmonic_shifts = as_float(harmonic_shifts, dtype)
    n_harmonics = int(harmonic_shifts.shape[-1])
  else:
    n_harmonics = 1
  harmonic_frequencies = get_harmonic_frequencies(frequencies, n_harmonics)
  if harmonic_shifts is not None:
    harmonic_frequencies = harmonic_frequencies * (dtype(1.0) + harmonic_shifts)
  if harmonic_distribution is not None:
    harmonic_amplitudes = amplitudes * harmonic_distribution
  else:
    harmonic_amplitudes = amplitudes
  frequency_envelopes = resample(harmonic_frequencies, n_samples, dtype=dtype)
  amplitude_envelopes = resample(harmonic_amplitudes, n_samples,
                                 method=amp_resample_method, dtype=dtype)
  return oscillator_bank(frequency_envelopes, amplitude_envelopes,
                         sample_rate=sample_rate, use_angular_cumsum=use_angular_cumsum)


# ----------------------------------------------------------------------------
# Time-varying FIR   (ddsp/core.py:1317-1565, 1628-1655)
# ----------------------------------------------------------------------------
def harmonic_oscillator_bank(frequency, amplitude_envelopes, initial_phase=None,
                             sample_rate=16000, use_angular_cumsum=True):
  """core.harmonic_oscillator_bank (core.py:966-1025) -> (audio [B,N], final_phase [B,1,1])."""
  dtype = amplitude_envelopes.dtype
  omega = frequency * dtype.type(TWO_PI)
  omega = omega / dtype.type(sample_rate)
  phases = angular_cumsum(omega) if use_angular_cumsum else np.cumsum(omega, axis=1, dtype=dtype)
  if initial_phase is None:
    initial_phase = np.zeros([phases.shape[0], 1, 1], dtype)
  phases = phases + as_float(initial_phase, dtype)
  final_phase = phases[:, -1:, 0:1]
  n_harmonics = int(amplitude_envelopes.shape[-1])
  f_ratios = np.linspace(1.0, float(n_harmonics), n_harmonics).astype(dtype)[None, None, :]
  phases = phases * f_ratios
  audio = np.sum(amplitude_envelopes * np.sin(phases), axis=-1, dtype=dtype)
  return audio, final_phase


def streaming_harmonic_synthesis(frequencies, amplitudes, harmonic_distribution=None,
                                 initial_phase=None, n_samples=64000, sample_rate=16000,
                                 amp_resample_method='linear', dtype=np.float32):
  """core.streaming_harmonic_synthesis (core.py:1114-1164)."""
  frequencies, amplitudes = as_float(frequencies, dtype), as_float(amplitudes, dtype)
  if harmonic_distribution is not None:
    harmonic_distribution = normalize_harmonics(as_float(harmonic_distribution, dtype),
                                                frequencies, sample_rate)
    harmonic_amplitudes = amplitudes * harmonic_distribution
  else:
    harmonic_amplitudes = amplitudes
  frequencies = resample(frequencies, n_samples, dtype=dtype)
  amplitude_envelopes = resample(harmonic_amplitudes, n_samples, method=amp_resample_method,
                                 dtype=dtype)
  return harmonic_oscillator_bank(frequencies, amplitude_envelopes, initial_phase,
                                  sample_rate=sample_rate)


def get_fft_size(frame_size, ir_size, power_of_2=True):
  """core.get_fft_size (core.py:1317-1335), power-of-two branch only."""
  convolved_frame_size = ir_size + frame_size - 1
  if not power_of_2:
    raise NotImplementedError('the hot path always uses power_of_2=True')
  return int(2**np.ceil(np.log2(convolved_frame_size)))


def crop_and_compensate_delay(audio, audio_size, ir_size, padding, delay_compensation):
  """core.crop_and_compensate_delay (core.py:1338-1379)."""
  if padding == 'valid':
    crop_size = ir_size + audio_size - 1
  elif padding == 'same':
    crop_size = audio_size
  else:
    raise ValueError('Padding must be \'valid\' or \'same\', instead '
                     'of {}.'.format(padding))
  total_size = int(audio.shape[-1])
  crop = total_size - crop_size
  start = ((ir_size - 1) // 2 - 1 if delay_compensation < 0 else delay_compensation)
  end = crop - start
  return audio[:, start:-end]


def frame_pad_end(audio, frame_size, hop_size):
  """tf.signal.frame(audio[B,N], frame, hop, pad_end=True) -> [B, ceil(N/hop), frame]."""
  n = audio.shape[-1]
  n_frames = -(-n // hop_size)
  padded_len = (n_frames - 1) * hop_size + frame_size
  padded = np.zeros(audio.shape[:-1] + (padded_len,), audio.dtype)
  padded[..., :n] = audio
  idx = np.arange(n_frames)[:, None] * hop_size + np.arange(frame_size)[None, :]
  return padded[..., idx]


def fft_convolve(audio, impulse_response, padding='same', delay_compensation=-1,
                 dtype=np.float32):
  """core.fft_convolve (core.py:1382-1473): framed FFT convolution + overlap-add."""
  audio, impulse_response = as_float(audio, dtype), as_float(impulse_response, dtype)
  batch_size, audio_size = audio.shape
  ir_shape = impulse_response.shape
  if len(ir_shape) == 2:
    impulse_response = impulse_response[:, None, :]
  if ir_shape[0] == 1 and batch_size > 1:
    impulse_response = np.tile(impulse_response, [batch_size, 1, 1])
  batch_size_ir, n_ir_frames, ir_size = impulse_response.shape
  if batch_size != batch_size_ir:
    raise ValueError('Batch size of audio ({}) and impulse response ({}) must '
                     'be the same.'.format(batch_size, batch_size_ir))
  frame_size = int(np.ceil(audio_size / n_ir_frames))
  hop_size = frame_size
  audio_frames = frame_pad_end(audio, frame_size, hop_size)
  n_audio_frames = int(audio_frames.shape[1])
  if n_audio_frames != n_ir_frames:
    raise ValueError(
        'Number of Audio frames ({}) and impulse response frames ({}) do not '
        'match. For small hop size = ceil(audio_size / n_ir_frames), '
        'number of impulse response frames must be a multiple of the audio '
        'size.'.format(n_audio_frames, n_ir_frames))
  fft_size = get_fft_size(frame_size, ir_size, power_of_2=True)
  audio_fft = np.fft.rfft(audio_frames, fft_size)      # complex64 for fp32 input
  ir_fft = np.fft.rfft(impulse_response, fft_size)
  audio_ir_fft = audio_fft * ir_fft
  audio_frames_out = np.fft.irfft(audio_ir_fft, fft_size).astype(dtype)
  audio_out = overlap_and_add(audio_frames_out, hop_size)
  return crop_and_compensate_delay(audio_out, audio_size, ir_size, padding,
                                   delay_compensation)


def time_varying_fir_direct(audio, impulse_response, delay_compensation=-1,
                            dtype=np.float64, n_out=None):
  """Direct-form equivalent of fft_convolve: padding='same' by default, any crop with
  n_out (padding='valid' is n_out = ir_size + n - 1).

  z[m] = sum_k x[m-k] * h_{frame(m-k)}[k];  out[n] = z[n + start], zero beyond the support of z.
  The tap set is chosen by the frame of the INPUT sample.  This is the form the
  HIP kernel evaluates; used to cross-check fft_convolve on small cases.
  """
  audio, ir = as_float(audio, dtype), as_float(impulse_response, dtype)
  if ir.ndim == 2:
    ir = ir[:, None, :]
  b, n = audio.shape
  if ir.shape[0] == 1 and b > 1:
    ir = np.tile(ir, [b, 1, 1])
  n_frames, ir_size = ir.shape[1], ir.shape[2]
  frame_size = int(np.ceil(n / n_frames))
  z = np.zeros((b, n_frames * frame_size + ir_size - 1), dtype)
  for i in range(n):
    z[:, i:i + ir_size] += audio[:, i:i + 1] * ir[:, i // frame_size, :]
  start = ((ir_size - 1) // 2 - 1 if delay_compensation < 0 else delay_compensation)
  n_out = n if n_out is None else int(n_out)
  if start + n_out > z.shape[1]:
    z = np.concatenate([z, np.zeros((b, start + n_out - z.shape[1]), dtype)], axis=1)
  return z[:, start:start + n_out]


def apply_window_to_impulse_response(impulse_response, window_size=0, causal=False,
                                     dtype=np.float32):
  """core.apply_window_to_impulse_response (core.py:1477-1531)."""
  impulse_response = as_float(impulse_response, dtype)
  if causal:
    impulse_response = np.fft.fftshift(impulse_response, axes=-1)
  ir_size = int(impulse_response.shape[-1])
  if (window_size <= 0) or (window_size > ir_size):
    window_size = ir_size
  window = hann_window_periodic(window_size, dtype)
  padding = ir_size - window_size
  if padding > 0:
    half_idx = (window_size + 1) // 2
    window = np.concatenate([window[half_idx:], np.zeros([padding], dtype),
                             window[:half_idx]], axis=0)
  else:
    window = np.fft.fftshift(window, axes=-1)
  impulse_response = window * impulse_response
  if padding > 0:
    first_half_start = (ir_size - (half_idx - 1)) + 1
    second_half_end = half_idx + 1
    impulse_response = np.concatenate([impulse_response[..., first_half_start:],
                                       impulse_response[..., :second_half_end]], axis=-1)
  else:
    impulse_response = np.fft.fftshift(impulse_response, axes=-1)
  return impulse_response


def frequency_impulse_response(magnitudes, window_size=0, dtype=np.float32):
  """core.frequency_impulse_response (core.py:1534-1565)."""
  magnitudes = as_float(magnitudes, dtype)
  ctype = np.complex64 if dtype == np.float32 else np.complex128
  impulse_response = np.fft.irfft(magnitudes.astype(ctype)).astype(dtype)
  return apply_window_to_impulse_response(impulse_response, window_size, dtype=dtype)


def frequency_filter(audio, magnitudes, window_size=0, padding='same', dtype=np.float32):
  """core.frequency_filter (core.py:1628-1655)."""
  impulse_response = frequency_impulse_response(magnitudes, window_size, dtype)
  return fft_convolve(audio, impulse_response, padding=padding, dtype=dtype)


# ----------------------------------------------------------------------------
# Processors   (ddsp/synths.py:55-196, ddsp/processors.py:37-76, 162-176)
# ----------------------------------------------------------------------------
def harmonic_get_controls(amplitudes, harmonic_distribution, f0_hz, sample_rate=16000,
                          scale_fn=exp_sigmoid, normalize_below_nyquist=True,
                          dtype=np.float32):
  """synths.Harmonic.get_controls (synths.py:94-121)."""
  amplitudes = as_float(amplitudes, dtype)
  harmonic_distribution = as_float(harmonic_distribution, dtype)
  f0_hz = as_float(f0_hz, dtype)
  if scale_fn is not None:
    amplitudes = scale_fn(amplitudes, dtype=dtype)
    harmonic_distribution = scale_fn(harmonic_distribution, dtype=dtype)
  harmonic_distribution = normalize_harmonics(
      harmonic_distribution, f0_hz, sample_rate if normalize_below_nyquist else None)
  return {'amplitudes': amplitudes, 'harmonic_distribution': harmonic_distribution,
          'f0_hz': f0_hz}


def harmonic_get_signal(amplitudes, harmonic_distribution, f0_hz, n_samples=64000,
                        sample_rate=16000, amp_resample_method='window',
                        use_angular_cumsum=False, dtype=np.float32):
  """synths.Harmonic.get_signal (synths.py:123-146)."""
  return harmonic_synthesis(frequencies=f0_hz, amplitudes=amplitudes,
                            harmonic_distribution=harmonic_distribution,
                            n_samples=n_samples, sample_rate=sample_rate,
                            amp_resample_method=amp_resample_method,
                            use_angular_cumsum=use_angular_cumsum, dtype=dtype)


def harmonic(amplitudes, harmonic_distribution, f0_hz, n_samples=64000, sample_rate=16000,
             scale_fn=exp_sigmoid, normalize_below_nyquist=True,
             amp_resample_method='window', use_angular_cumsum=False, dtype=np.float32):
  """Processor.call for Harmonic (processors.py:53-68): get_signal(**get_controls())."""
  c = harmonic_get_controls(amplitudes, harmonic_distribution, f0_hz, sample_rate,
                            scale_fn, normalize_below_nyquist, dtype)
  return harmonic_get_signal(c['amplitudes'], c['harmonic_distribution'], c['f0_hz'],
                             n_samples, sample_rate, amp_resample_method,
                             use_angular_cumsum, dtype)


def filtered_noise_get_controls(magnitudes, scale_fn=exp_sigmoid, initial_bias=-5.0,
                                dtype=np.float32):
  """synths.FilteredNoise.get_controls (synths.py:165-179)."""
  magnitudes = as_float(magnitudes, dtype)
  if scale_fn is not None:
    magnitudes = scale_fn(magnitudes + dtype(initial_bias), dtype=dtype)
  return {'magnitudes': magnitudes}


def filtered_noise_get_signal(magnitudes, noise, window_size=257, dtype=np.float32):
  """synths.FilteredNoise.get_signal (synths.py:181-196) with the noise supplied.

  The reference draws tf.random.uniform([B, n_samples], -1, 1) from TF's stateful
  generator, which cannot be reproduced outside TF; parity is therefore defined
  with the same noise array handed to both implementations.
  """
  return frequency_filter(as_float(noise, dtype), magnitudes, window_size=window_size,
                          dtype=dtype)


def filtered_noise(magnitudes, noise, window_size=257, scale_fn=exp_sigmoid,
                   initial_bias=-5.0, dtype=np.float32):
  c = filtered_noise_get_controls(magnitudes, scale_fn, initial_bias, dtype)
  return filtered_noise_get_signal(c['magnitudes'], noise, window_size, dtype)


def reverb_mask_dry_ir(ir, dtype=np.float32):
  """effects.Reverb._mask_dry_ir (effects.py:50-60): 2-D IR with tap 0 zeroed."""
  ir = as_float(ir, dtype)
  if ir.ndim == 1:
    ir = ir[None, :]
  if ir.ndim == 3:
    ir = ir[:, :, 0]
  return np.concatenate([np.zeros((ir.shape[0], 1), dtype), ir[:, 1:]], axis=1)


def reverb(audio, ir, add_dry=True, dtype=np.float32):
  """effects.Reverb.get_signal (effects.py:100-117), ir given or a trainable Reverb's single IR.

  wet = fft_convolve(audio, mask_dry(ir), padding='same', delay_compensation=0): one frame,
  i.e. the first n_samples of the causal linear convolution; a 1-D / batch-1 IR is tiled over
  the batch (effects.py:62-69, core.py:1433-1434).
  """
  audio = as_float(audio, dtype)
  ir = reverb_mask_dry_ir(ir, dtype)
  wet = fft_convolve(audio, ir, padding='same', delay_compensation=0, dtype=dtype)
  return (wet + audio) if add_dry else wet


def exp_decay_ir(gain, decay, noise, scale_fn=exp_sigmoid, dtype=np.float32):
  """effects.ExpDecayReverb._get_ir (effects.py:144-151): gain, decay [B,1], noise [1,L] -> ir [B,L].

  ir = scale_fn(gain) * exp(-(2 + exp(decay)) * linspace(0, 1, L)) * noise.  The reference draws the noise
  from tf.random.uniform([1, L], -1, 1) inside; here it is an argument (the parity entry)."""
  gain, decay, noise = as_float(gain, dtype), as_float(decay, dtype), as_float(noise, dtype)
  if scale_fn is not None:
    gain = scale_fn(gain, dtype=dtype)
  decay_exponent = dtype(2.0) + np.exp(decay)
  length = noise.shape[-1]
  time = np.linspace(0.0, 1.0, length).astype(dtype)[None, :]
  return gain.reshape(-1, 1) * np.exp(-decay_exponent.reshape(-1, 1) * time) * noise.reshape(1, -1)


def exp_decay_ir_backward(gain, decay, noise, grad_ir, scale_fn=exp_sigmoid):
  """(dL/d gain [B,1], dL/d decay [B,1]) of exp_decay_ir given dL/d ir [B,L] (fp64 truth)."""
  gain, decay = as_float(gain, np.float64).reshape(-1, 1), as_float(decay, np.float64).reshape(-1, 1)
  noise, g = as_float(noise, np.float64).reshape(1, -1), as_float(grad_ir, np.float64)
  time = np.linspace(0.0, 1.0, noise.shape[-1])[None, :]
  envelope = np.exp(-(2.0 + np.exp(decay)) * time)
  scaled = scale_fn(gain, dtype=np.float64) if scale_fn is not None else gain
  d_scaled = exp_sigmoid_grad(gain) if scale_fn is not None else np.ones_like(gain)
  s0 = np.sum(g * envelope * noise, axis=1, keepdims=True)
  s1 = np.sum(g * envelope * noise * time, axis=1, keepdims=True)
  return d_scaled * s0, -np.exp(decay) * scaled * s1


def reverb_direct(audio, ir, add_dry=True):
  """fp64 direct-form truth for reverb(): y[n] = sum_{k>=1} ir[k] x[n-k] (+ x[n])."""
  audio = as_float(audio, np.float64)
  ir = reverb_mask_dry_ir(ir, np.float64)
  if ir.shape[0] == 1 and audio.shape[0] > 1:
    ir = np.tile(ir, [audio.shape[0], 1])
  n = audio.shape[1]
  wet = np.stack([np.convolve(audio[b], ir[b])[:n] for b in range(audio.shape[0])])
  return (wet + audio) if add_dry else wet


# ----------------------------------------------------------------------------
# Backward pass of synths.Harmonic.__call__ (what tf.GradientTape computes through
# ddsp/synths.py:94-146, trainers.py:162-171).  fp64 only: it is the truth
# the HIP gradients are compared with, itself checked against finite differences in tests/.
# ----------------------------------------------------------------------------
def exp_sigmoid_grad(x, exponent=10.0, max_value=2.0, threshold=1e-7):
  """d/dx of core.exp_sigmoid: log(exponent) * (y - threshold) * (1 - sigmoid(x))."""
  x = as_float(x, np.float64)
  y = exp_sigmoid(x, exponent, max_value, threshold, dtype=np.float64)
  return np.log(exponent) * (y - threshold) * (1.0 - sigmoid(x))


def harmonic_backward(amplitudes, harmonic_distribution, f0_hz, grad_audio, n_samples=64000,
                      sample_rate=16000, scale_fn=exp_sigmoid, normalize_below_nyquist=True,
                      amp_resample_method='window', with_f0=False):
  """(dL/d amplitudes [B,F,1], dL/d harmonic_distribution [B,F,K][, dL/d f0_hz [B,F,1]]) given
  dL/d audio [B,N].

  The masks (>= Nyquist) have zero gradient, as tf.where gives them (also with respect to f0).
  """
  amps_raw = as_float(amplitudes, np.float64)
  hd_raw = as_float(harmonic_distribution, np.float64)
  f0 = as_float(f0_hz, np.float64)
  g = as_float(grad_audio, np.float64)
  b, f, k = hd_raw.shape
  # ---- forward pieces ----
  amp_s = scale_fn(amps_raw, dtype=np.float64) if scale_fn is not None else amps_raw
  x = scale_fn(hd_raw, dtype=np.float64) if scale_fn is not None else hd_raw
  if normalize_below_nyquist:
    live = get_harmonic_frequencies(f0, k) < sample_rate / 2.0
    x = np.where(live, x, 0.0)
  else:
    live = np.ones_like(x, bool)
  den = np.sum(x, axis=-1, keepdims=True)
  den_safe = np.where(den == 0.0, 1e-7, den)
  hdn = x / den_safe
  # audio-rate pieces of harmonic_synthesis / oscillator_bank
  freq_env = resample(get_harmonic_frequencies(f0, k), n_samples, dtype=np.float64)     # [B,N,K]
  mask = freq_env < sample_rate / 2.0
  phases = np.cumsum(freq_env * (TWO_PI / float(sample_rate)), axis=1)
  gs = g[:, :, None] * np.where(mask, np.sin(phases), 0.0)                              # [B,N,K]
  # the amplitude upsampling is linear in its input: U [N,F] from the identity
  u = resample(np.eye(f)[None], n_samples, method=amp_resample_method, dtype=np.float64)[0]
  grad_a = np.einsum('nf,bnk->bfk', u, gs)                                              # dL/d(amp*hdn)
  # ---- frame-rate chain rule ----
  d_amp_s = np.sum(grad_a * hdn, axis=-1, keepdims=True)
  d_hdn = grad_a * amp_s
  d_x = (d_hdn - np.sum(d_hdn * hdn, axis=-1, keepdims=True)) / den_safe
  d_x = np.where(live & (den != 0.0), d_x, 0.0)
  out = ((d_amp_s * exp_sigmoid_grad(amps_raw), d_x * exp_sigmoid_grad(hd_raw)) if scale_fn is not None
         else (d_amp_s, d_x))
  if not with_f0:
    return out
  # dL/d f0: phase[n] = (2 pi / sr) k cumsum(f_env)[n]  ->  dL/d f_env[t] = (2 pi / sr) sum_{n >= t} c[n],
  # c[n] = g[n] sum_k k A_k[n] m_k[n] cos(phase_k[n]); f_env = U_f f0 (legacy bilinear resize, linear in f0)
  amp_env = resample(amp_s * hdn, n_samples, method=amp_resample_method, dtype=np.float64)   # [B,N,K]
  kk = np.arange(1, k + 1, dtype=np.float64)
  c = g * np.sum(kk * amp_env * np.where(mask, np.cos(phases), 0.0), axis=-1)                # [B,N]
  r_suffix = np.cumsum(c[:, ::-1], axis=1)[:, ::-1]
  u_f = resample(np.eye(f)[None], n_samples, dtype=np.float64)[0]                             # [N,F]
  d_f0 = (TWO_PI / float(sample_rate)) * np.einsum('nf,bn->bf', u_f, r_suffix)[:, :, None]
  return out + (d_f0,)


def filtered_noise_backward(magnitudes, noise, grad_audio, window_size=257, scale_fn=exp_sigmoid,
                            initial_bias=-5.0):
  """dL/d magnitudes [B,F,M] of filtered_noise() given dL/d audio [B,N] (fp64 truth).

  out is linear in the impulse responses (fft_convolve = time-varying FIR) and the
  impulse response of a frame is linear in its magnitudes; only exp_sigmoid is not.
  """
  mags_raw = as_float(magnitudes, np.float64)
  x = as_float(noise, np.float64)
  g = as_float(grad_audio, np.float64)
  b, f, m = mags_raw.shape
  n = x.shape[1]
  ctl = scale_fn(mags_raw + initial_bias, dtype=np.float64) if scale_fn is not None else mags_raw
  # A [L, M]: the (windowed, causal) impulse response of each unit magnitude
  a = frequency_impulse_response(np.eye(m)[None], window_size=window_size, dtype=np.float64)[0].T
  l = a.shape[0]
  frame_size = int(np.ceil(n / f))
  start = (l - 1) // 2 - 1
  gz = np.zeros((b, f * frame_size + l + start))
  gz[:, start:start + n] = g                              # out[n] = z[n + start]
  xp = np.zeros((b, f * frame_size))
  xp[:, :n] = x
  d_ir = np.zeros((b, f, l))
  for fr in range(f):
    for i in range(frame_size):
      p0 = fr * frame_size + i
      d_ir[:, fr, :] += xp[:, p0:p0 + 1] * gz[:, p0:p0 + l]
  d_ctl = d_ir @ a                                        # [B,F,L] x [L,M]
  if scale_fn is not None:
    return d_ctl * exp_sigmoid_grad(mags_raw + initial_bias)
  return d_ctl


def reverb_backward(audio, ir, grad_out, add_dry=True):
  """(dL/d audio [B,N], dL/d ir [Bir,L]) of reverb() given dL/d out [B,N] (fp64 truth).

  out[n] = sum_{k>=1} ir[k] audio[n-k] (+ audio[n]):  dL/d audio[m] = sum_n g[n] ir[n-m] (+ g[m]),
  dL/d ir[k] = sum_n g[n] audio[n-k] for k >= 1 and 0 for the masked tap; a single (tiled) IR
  collects the sum over the batch.
  """
  x = as_float(audio, np.float64)
  g = as_float(grad_out, np.float64)
  h = reverb_mask_dry_ir(ir, np.float64)
  b, n = x.shape
  bir, l = h.shape
  d_audio = np.zeros_like(x)
  d_ir = np.zeros((b, l))
  for r in range(b):
    hr = h[r % bir]
    # sum_n g[n] h[n-m] = correlate(g, h)[m]
    d_audio[r] = np.correlate(np.concatenate([g[r], np.zeros(l - 1)]), hr, mode='valid')[:n]
    full = np.convolve(g[r], x[r][::-1])                   # full[N-1+k] = sum_n g[n] x[n-k]
    d_ir[r, :min(l, n)] = full[n - 1:n - 1 + min(l, n)]
  d_ir[:, 0] = 0.0
  if add_dry:
    d_audio = d_audio + g
  if bir == 1:
    d_ir = d_ir.sum(axis=0, keepdims=True)
  return d_audio, d_ir


# ----------------------------------------------------------------------------
# losses.SpectralLoss  (ddsp/losses.py:100-243, ddsp/spectral_ops.py:34-70)
# ----------------------------------------------------------------------------
def stft(audio, frame_size=2048, overlap=0.75, pad_end=True, dtype=np.float32):
  """spectral_ops.stft (spectral_ops.py:34-47) = tf.signal.stft: frames of `frame_size` every
  frame_size*(1-overlap) samples (zero pad_end), periodic Hann, rfft of the enclosing power of 2."""
  audio = as_float(audio, dtype)
  if audio.ndim == 3:
    audio = audio[..., 0]
  frame_size = int(frame_size)
  hop = int(frame_size * (1.0 - overlap))
  fft_length = 1 << int(np.ceil(np.log2(frame_size)))
  frames = frame_pad_end(audio, frame_size, hop) if pad_end else None
  if frames is None:
    n = 1 + (audio.shape[1] - frame_size) // hop
    frames = np.stack([audio[:, i * hop:i * hop + frame_size] for i in range(n)], axis=1)
  return np.fft.rfft(frames * hann_window_periodic(frame_size, dtype), fft_length)


def compute_mag(audio, size=2048, overlap=0.75, pad_end=True, dtype=np.float32):
  """spectral_ops.compute_mag (spectral_ops.py:67-70): |stft|."""
  return np.abs(stft(audio, size, overlap, pad_end, dtype)).astype(dtype)


def safe_log(x, eps=1e-5):
  """core.safe_log (core.py:213-216): log of x with non-positive entries replaced by eps."""
  return np.log(np.where(x <= 0.0, np.asarray(eps, x.dtype), x))


def mean_difference(target, value, loss_type='L1', weights=None):
  """losses.mean_difference (losses.py:102-128).  'COSINE' is tf.compat.v1.losses.cosine_distance(target, value,
  weights, axis=-1): 1 - sum(target * value, -1, keepdims) per row, weighted, summed and divided by the number of
  non-zero weights (Reduction.SUM_BY_NONZERO_WEIGHTS; 0 if there are none)."""
  difference = target - value
  weights = 1.0 if weights is None else weights
  loss_type = loss_type.upper()
  if loss_type == 'L1':
    return np.mean(np.abs(difference * weights), dtype=difference.dtype)
  if loss_type == 'L2':
    return np.mean(difference**2 * weights, dtype=difference.dtype)
  if loss_type == 'COSINE':
    losses = 1.0 - np.sum(target * value, axis=-1, keepdims=True)
    w = np.broadcast_to(np.asarray(weights, losses.dtype), losses.shape)
    present = np.count_nonzero(w)
    return (np.sum(losses * w) / present).astype(losses.dtype) if present else losses.dtype.type(0.0)
  raise ValueError('Loss type ({}), must be "L1", "L2", or "COSINE"'.format(loss_type))


def diff(x, axis=-1):
  """core.diff (core.py:171-199): x[1:] - x[:-1] along `axis`."""
  x = np.moveaxis(x, axis, 0)
  return np.moveaxis(x[1:] - x[:-1], 0, axis)


def a_weighting_db(sample_rate=16000, n_fft=2048):
  """librosa.A_weighting(librosa.fft_frequencies(sr, n_fft)) as spectral_ops.compute_loudness calls them (spectral_ops.py:307-308).
  librosa is a third-party dependency of the reference (setup.py: 'librosa', unpinned; not under /root/reference): its published
  formula (librosa/core/convert.py, unchanged 0.8 .. 0.10) - the IEC 61672 A-curve in dB, clipped at min_db = -80 (f = 0)."""
  f_sq = np.fft.rfftfreq(n=n_fft, d=1.0 / sample_rate) ** 2.0
  c = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
  with np.errstate(divide='ignore'):
    w = 2.0 + 20.0 * (np.log10(c[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + c[0]) - np.log10(f_sq + c[1])
                      - 0.5 * np.log10(f_sq + c[2]) - 0.5 * np.log10(f_sq + c[3]))
  return np.maximum(-80.0, w)


def _loudness_frames(audio, n_fft, hop):
  """pad(audio, 'center') (spectral_ops.py:171-218) + tf.signal.frame(pad_end=False): frames of n_fft every hop, the first
  centred on sample 0 -> [B, 1 + N // hop, n_fft]."""
  b, n = audio.shape
  padded = np.pad(audio, [(0, 0), (n_fft // 2, n_fft // 2)])
  n_frames = 1 + (padded.shape[1] - n_fft) // hop
  idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
  return padded[:, idx]


def compute_loudness(audio, sample_rate=16000, frame_rate=250, n_fft=2048, range_db=80.0, ref_db=0.0, dtype=np.float32):
  """spectral_ops.compute_loudness(use_tf=True, padding='center') (spectral_ops.py:253-324): A-weighted mean power per frame in
  dB, floored at -range_db -> [B, 1 + N // hop]."""
  audio = as_float(audio, dtype)
  if audio.ndim == 3:
    audio = audio[..., 0]
  hop = sample_rate // frame_rate
  frames = _loudness_frames(audio, n_fft, hop)
  s = np.fft.rfft(frames * hann_window_periodic(n_fft, dtype), n_fft)
  power = (np.abs(s).astype(dtype)) ** 2
  weighting = (10.0 ** (a_weighting_db(sample_rate, n_fft) / 10.0)).astype(dtype)
  avg_power = np.mean(power * weighting, axis=-1, dtype=dtype)
  pmin = dtype(10.0 ** -(range_db / 10.0))                        # core.power_to_db (core.py:253-267)
  db = dtype(10.0) * (np.log(np.maximum(pmin, avg_power)) / dtype(np.log(10.0))).astype(dtype)
  return np.maximum(db - dtype(ref_db), dtype(-range_db)).astype(dtype)


def compute_loudness_backward(audio, grad_loudness, sample_rate=16000, frame_rate=250, n_fft=2048, range_db=80.0):
  """dL/d audio of compute_loudness given dL/d loudness [B, frames] (fp64): through max(., -range) and max(pmin, .) (no
  gradient where either clips), the weighted mean power, |rfft|^2, the window, the frames' overlap-add and the centre padding."""
  a = as_float(audio, np.float64)
  b, n = a.shape
  hop = sample_rate // frame_rate
  frames = _loudness_frames(a, n_fft, hop)
  win = hann_window_periodic(n_fft, np.float64)
  s = np.fft.rfft(frames * win, n_fft)
  weighting = 10.0 ** (a_weighting_db(sample_rate, n_fft) / 10.0)
  bins = s.shape[-1]
  avg_power = np.mean(np.abs(s) ** 2 * weighting, axis=-1)
  pmin = 10.0 ** -(range_db / 10.0)
  db = 10.0 * np.log10(np.maximum(pmin, avg_power))
  live = (avg_power > pmin) & (db > -range_db)
  g_p = np.where(live, np.asarray(grad_loudness, np.float64) * 10.0 / (np.log(10.0) * np.where(live, avg_power, 1.0)), 0.0)
  # d|X_k|^2 / dx_i = 2 Re(X_k exp(+2 pi i k i / n)): the sum over the rfft's bins is the real part of an inverse transform of
  # the half spectrum (no doubling of the inner bins: each |X_k|^2 of the mean is one term)
  g_s = (g_p[..., None] * weighting / bins) * 2.0 * s
  full = np.zeros(s.shape[:-1] + (n_fft,), complex)
  full[..., :bins] = g_s
  g_frames = np.real(np.fft.ifft(full, axis=-1) * n_fft) * win
  padded = np.zeros((b, n + n_fft))
  for f in range(g_frames.shape[1]):
    padded[:, f * hop:f * hop + n_fft] += g_frames[:, f]
  return padded[:, n_fft // 2:n_fft // 2 + n]


def spectral_loss(target_audio, audio, fft_sizes=(2048, 1024, 512, 256, 128, 64), loss_type='L1',
                  mag_weight=1.0, logmag_weight=0.0, dtype=np.float32, delta_time_weight=0.0, delta_freq_weight=0.0,
                  cumsum_freq_weight=0.0, weights=None, loudness_weight=0.0):
  """losses.SpectralLoss.call (losses.py:189-243), every term (the loudness one: compute_loudness above, n_fft = 2048)."""
  loss = dtype(0.0)
  if weights is not None:
    weights = np.asarray(weights, dtype)
  for size in fft_sizes:
    target_mag = compute_mag(target_audio, size, dtype=dtype)
    value_mag = compute_mag(audio, size, dtype=dtype)
    if mag_weight > 0:
      loss += dtype(mag_weight) * mean_difference(target_mag, value_mag, loss_type, weights)
    if delta_time_weight > 0:
      loss += dtype(delta_time_weight) * mean_difference(diff(target_mag, 1), diff(value_mag, 1), loss_type, weights)
    if delta_freq_weight > 0:
      loss += dtype(delta_freq_weight) * mean_difference(diff(target_mag, 2), diff(value_mag, 2), loss_type, weights)
    if cumsum_freq_weight > 0:
      loss += dtype(cumsum_freq_weight) * mean_difference(np.cumsum(target_mag, axis=2, dtype=dtype),
                                                          np.cumsum(value_mag, axis=2, dtype=dtype), loss_type, weights)
    if logmag_weight > 0:
      loss += dtype(logmag_weight) * mean_difference(safe_log(target_mag), safe_log(value_mag),
                                                     loss_type, weights)
  if loudness_weight > 0:                                          # losses.py:238-242
    loss += dtype(loudness_weight) * mean_difference(compute_loudness(target_audio, n_fft=2048, dtype=dtype),
                                                     compute_loudness(audio, n_fft=2048, dtype=dtype), loss_type, weights)
  return loss


def spectral_loss_backward(target_audio, audio, fft_sizes=(2048, 1024, 512, 256, 128, 64),
                           mag_weight=1.0, logmag_weight=0.0, fp32_envelope=None):
  """dL/d audio [B,N] of spectral_loss(target_audio, audio, loss_type='L1') (fp64 truth).

  |z| has gradient z/|z| (0 at z = 0, as tf.abs), safe_log passes a gradient only where its
  argument is positive, sign(0) = 0; the zero-padded tail of the last frames gets no gradient.

  fp32_envelope = r (tests/test_gpu_parity.py::check_loss_case: 5e-6): returns (gradient, envelope) instead.  The L1 loss is not differentiable where
  two magnitudes are equal, and the logmag term's 1 / |X| (core.safe_log replaces only NON-POSITIVE arguments, core.py:213-216) is
  unbounded at a spectral null (and the mag term's z / |z| has no direction there).  fp32 arithmetic - TensorFlow's included - knows a magnitude to `floor` = r of its frame's
  spectrum + 1e-6 of the norm of the frame's UNWINDOWED samples (the window's own absolute accuracy: what is left of a frame that
  only touches the signal with the last points of its window).  A bin whose |X_t| - |X_a| is within r of their sum + floor has no
  sign fp32 can tell from its neighbours in the SUBDIFFERENTIAL; a bin whose |X_a| is within 30 floor of zero has a 1 / |X_a| that is
  anything, one within 3 floor a phasor X_a / |X_a| that is any.  Such bins are left out of `gradient`, and `envelope[b, n]` is the largest magnitude their terms can add at sample n
  (every admissible coefficient, every phase; inf under a frame with a bin at the noise floor): a correct fp32 gradient g satisfies
  |g - gradient| <= envelope + rounding, sample by sample.
  """
  t = as_float(target_audio, np.float64)
  a = as_float(audio, np.float64)
  b, n = a.shape
  grad = np.zeros_like(a)
  envelope = np.zeros_like(a)
  for size in fft_sizes:
    hop = int(size * 0.25)
    zt = stft(t, size, dtype=np.float64)
    za = stft(a, size, dtype=np.float64)
    mt, ma = np.abs(zt), np.abs(za)
    count = float(mt.size)
    coef = -mag_weight * np.sign(mt - ma)
    if logmag_weight > 0:
      coef = coef - logmag_weight * np.sign(safe_log(mt) - safe_log(ma)) * np.where(
          ma > 0.0, 1.0 / np.where(ma > 0.0, ma, 1.0), 0.0)
    if fp32_envelope is not None:
      r = float(fp32_envelope)
      frame_rms = np.sqrt(np.mean(ma * ma, axis=-1, keepdims=True)) + np.sqrt(np.mean(mt * mt, axis=-1, keepdims=True))
      frame_l2 = (np.sqrt((frame_pad_end(a, size, hop) ** 2).sum(axis=-1, keepdims=True)) +
                  np.sqrt((frame_pad_end(t, size, hop) ** 2).sum(axis=-1, keepdims=True)))
      floor = r * frame_rms + 1e-6 * frame_l2                      # what fp32 knows a magnitude of this frame to
      unsure = np.abs(mt - ma) <= r * (mt + ma) + floor            # the sign of the difference is rounding's to decide
      # |z| has gradient z / |z|: the unit phasor of a magnitude INSIDE the noise floor is rounding's to choose too (a DC bin
      # whose windowed samples cancel to 2.5e-8: fuzz seed loss:47050735, round 6 - mag term alone, one frame of 16 samples off by
      # its whole coefficient); the coefficient itself stays bounded by mag_weight
      unsure |= ma <= 3.0 * floor
      bound = np.full(ma.shape, float(mag_weight))
      if logmag_weight > 0:
        null = (ma > 0.0) & (ma <= 30.0 * floor)                   # 1 / |X_a| of a magnitude inside the noise floor: anything
        unsure |= null
        with np.errstate(divide='ignore'):
          bound = bound + np.where(null, np.inf, logmag_weight / np.maximum(ma - floor, 1e-300))
      # a bin's term at sample i of its frame: (coef / count) w[i] Re(unit phasor) (x 2 for the bins the rfft holds once)
      weight = np.where(unsure, 2.0 * bound / count, 0.0).sum(axis=-1)                       # [B, frames] (inf: the frame is out)
      if logmag_weight > 0:
        # ... and what the floor leaves uncertain of 1 / |X_a| in every OTHER bin: d(1 / m) = floor / m^2 - nothing for a bin of
        # ordinary size, a few per cent of a large coefficient for one 40 floors above zero
        with np.errstate(divide='ignore', invalid='ignore'):
          slack = np.where(unsure | (ma <= 0.0), 0.0, logmag_weight * floor / (ma * ma))
        weight = weight + (2.0 * slack / count).sum(axis=-1)
      coef = np.where(unsure, 0.0, coef)
      with np.errstate(invalid='ignore'):
        env_frames = np.where(np.isinf(weight)[..., None], np.inf, weight[..., None] * hann_window_periodic(size, np.float64)[None, None, :])
      n_fr = env_frames.shape[1]
      env_padded = np.zeros((b, (n_fr - 1) * hop + size))
      for f in range(n_fr):
        env_padded[:, f * hop:f * hop + size] += env_frames[:, f]
      envelope += env_padded[:, :n]
    g_bins = (coef / count) * np.where(ma > 0.0, za / np.where(ma > 0.0, ma, 1.0), 0.0)
    # (the transform has the enclosing power of two as its length - stft() above -: the frame's gradient is the first `size`
    #  samples of the adjoint, the zero padding has none)
    fft_length = 2 * (za.shape[-1] - 1)
    full = np.zeros(za.shape[:-1] + (fft_length,), complex)
    full[..., :fft_length // 2 + 1] = g_bins
    g_frames = np.real(np.fft.ifft(full, axis=-1) * fft_length)[..., :size] * hann_window_periodic(size, np.float64)
    n_frames = g_frames.shape[1]
    padded = np.zeros((b, (n_frames - 1) * hop + size))
    for f in range(n_frames):
      padded[:, f * hop:f * hop + size] += g_frames[:, f]
    grad += padded[:, :n]
  return grad if fp32_envelope is None else (grad, envelope)


def spectral_loss_value_bounds(target_audio, audio, fft_sizes=(2048, 1024, 512, 256, 128, 64), mag_weight=1.0, logmag_weight=0.0,
                               fp32_floor=5e-6):
  """(lo, hi, n_null): the interval a correct fp32 evaluation of spectral_loss(.., loss_type='L1') may land in when some bin is a
  spectral NULL, and how many such bins there are (tests/test_gpu_parity.py::check_loss_case uses it only when n_null > 0).

  core.safe_log replaces only NON-POSITIVE arguments (core.py:213-216), so the logmag term of a bin whose magnitude cancels to
  2e-7 - the Nyquist bin of one frame in ten thousand (tools/fuzz_parity.py loss:107037044, round 6: exact 2.46e-7, the reference's
  own fp32 op order 4.75e-7, the MI355X 6e-9; 0.66 and 3.7 nats of a 10 320-term mean) - is log of whatever rounding left of it:
  any fp32 implementation, TensorFlow's included, returns an arbitrary number there.  `floor` is what fp32 knows a magnitude of a
  frame to (as in spectral_loss_backward: fp32_floor of the frame's spectrum + 1e-6 of the norm of its unwindowed samples).  A bin
  with |X_t| or |X_a| within 3 floor of zero is a null: its logmag term is only known to be >= 0 and <= its exact value + 110
  (a magnitude may come out as small as fp32's smallest positive number, or as 0 and take log(eps)); every other bin's term is known
  to -log(1 - floor / |X|) for each of its two magnitudes, and every mag term to 2 floor."""
  t = as_float(target_audio, np.float64)
  a = as_float(audio, np.float64)
  lo = hi = 0.0
  n_null = 0
  for size in fft_sizes:
    hop = int(size * 0.25)
    mt, ma = np.abs(stft(t, size, dtype=np.float64)), np.abs(stft(a, size, dtype=np.float64))
    count = float(mt.size)
    frame_rms = np.sqrt(np.mean(ma * ma, axis=-1, keepdims=True)) + np.sqrt(np.mean(mt * mt, axis=-1, keepdims=True))
    frame_l2 = (np.sqrt((frame_pad_end(a, size, hop) ** 2).sum(axis=-1, keepdims=True)) +
                np.sqrt((frame_pad_end(t, size, hop) ** 2).sum(axis=-1, keepdims=True)))
    floor = np.broadcast_to(float(fp32_floor) * frame_rms + 1e-6 * frame_l2, mt.shape)
    if mag_weight > 0:
      d = np.abs(mt - ma)
      lo += mag_weight * np.maximum(d - 2.0 * floor, 0.0).sum() / count
      hi += mag_weight * (d + 2.0 * floor).sum() / count
    if logmag_weight > 0:
      null = ((mt > 0.0) & (mt <= 3.0 * floor)) | ((ma > 0.0) & (ma <= 3.0 * floor))
      n_null += int(null.sum())
      terms = np.abs(safe_log(mt) - safe_log(ma))
      def wobble(m):        # how far log m moves when m moves by floor (m > 3 floor: at most 0.41)
        safe = np.where(m > 3.0 * floor, m, 4.0 * floor)
        return np.where(m > 3.0 * floor, -np.log1p(-floor / safe), 0.0)
      slack = np.where(null, 0.0, wobble(mt) + wobble(ma))
      lo += logmag_weight * np.where(null, 0.0, np.maximum(terms - slack, 0.0)).sum() / count
      hi += logmag_weight * np.where(null, terms + 110.0, terms + slack).sum() / count
  return lo, hi, n_null


def add(signal_one, signal_two):
  """processors.Add.get_signal (processors.py:174-176)."""
  return signal_one + signal_two


# ----------------------------------------------------------------------------
# Device noise generator restated (NOT part of the reference: the reference uses
# TF's stateful RNG, whose stream cannot be matched from outside TF - SURVEY.md H6).
# Philox4x32-R (Salmon et al., SC'11) with R = NOISE_ROUNDS = 10, the paper's default
# (round 3 measured R = 7 on the MI355X: no gain, profiles/r03k_*, so the contract stayed),
# counter = (sample_quad_index, batch_row, 0, 0), key = (seed_lo, seed_hi); word w of the
# output block is sample 4*quad + w.  u = bits>>9 as a 23-bit mantissa in [1,2),
# noise = (u - 1) * 2 - 1, i.e. the construction tf.random.uniform uses for fp32.
# ----------------------------------------------------------------------------
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


NOISE_ROUNDS = 10


def philox4x32_10(c0, c1, c2, c3, k0, k1):
  """The paper's default, kept for its published known-answer vector (tests/test_oracle.py)."""
  return philox4x32(c0, c1, c2, c3, k0, k1, 10)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=NOISE_ROUNDS):
  c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)]
  k0, k1 = np.uint64(k0), np.uint64(k1)
  mask = np.uint64(0xFFFFFFFF)
  for _ in range(rounds):
    p0 = c0 * np.uint64(PHILOX_M0)
    p1 = c2 * np.uint64(PHILOX_M1)
    hi0, lo0 = p0 >> np.uint64(32), p0 & mask
    hi1, lo1 = p1 >> np.uint64(32), p1 & mask
    c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
    k0 = (k0 + np.uint64(PHILOX_W0)) & mask
    k1 = (k1 + np.uint64(PHILOX_W1)) & mask
  return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def device_uniform_noise(batch_size, n_samples, seed=0, batch_offset=0, noise_bits=23):
  """The noise `ddsp_filtered_noise_f32` generates on chip when noise==NULL (csrc/common.h, "the generated noise"; the contract
  text is in include/ddsp_amd.h).  noise_bits=11 (FilteredNoise(noise_bits=11); the C entry points' form without DDSP_NOISE_BITS_23): 2048 equally spaced levels u = (2 k - 2047) / 2048 - zero mean,
  variance 1/3, every value an fp16 number -, k an 11-bit field of a Philox4x32 word; eight samples per block: sample n of a
  row is field ((n >> 1) & 3, n & 1) of block (n >> 3, row, 0, 0), bits [10:0] of the word for even n, [26:16] for odd n.
  noise_bits=23 (the default, as FilteredNoise's since round 6; DDSP_NOISE_BITS_23): the 2^23 levels of tf.random.uniform's fp32 samples
  (ddsp/synths.py:192-193) - sample n is word n & 3 of block (n >> 3, row, 1 + ((n >> 2) & 1), 0), its top 23 bits the
  mantissa of u in [1, 2), value 2 u - 3 (as rounds 1-3 made them, four per block)."""
  n_oct = -(-n_samples // 8)
  octet = np.arange(n_oct, dtype=np.uint64)[None, :].repeat(batch_size, 0)
  row = (np.arange(batch_size, dtype=np.uint64) + np.uint64(batch_offset))[:, None]
  row = np.broadcast_to(row, octet.shape)
  if noise_bits == 23:
    halves = []
    for half in (1, 2):                                                            # samples 8 q .. + 3, then 8 q + 4 .. + 7
      words = philox4x32(octet, row, np.full_like(octet, half), np.zeros_like(octet),
                         seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, NOISE_ROUNDS)
      halves.append(np.stack(words, axis=-1).astype(np.uint32))                    # [B, n_oct, 4]
    bits = np.stack(halves, axis=2).reshape(batch_size, n_oct * 8)[:, :n_samples]
    u = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).astype(np.uint32).view(np.float32)
    return (u * np.float32(2.0) - np.float32(3.0)).astype(np.float32)              # exact: u 2 in [2, 4), spacing 2^-22
  if noise_bits != 11:
    raise ValueError('noise_bits must be 11 or 23')
  words = philox4x32(octet, row, np.zeros_like(octet), np.zeros_like(octet),
                     seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, NOISE_ROUNDS)
  w = np.stack(words, axis=-1).astype(np.uint32)                                   # [B, n_oct, 4]
  fields = np.stack([w & np.uint32(0x7FF), (w >> np.uint32(16)) & np.uint32(0x7FF)], axis=-1)     # [B, n_oct, 4, 2]: even, odd
  k = fields.reshape(batch_size, n_oct * 8)[:, :n_samples].astype(np.int64)
  return ((2 * k - 2047).astype(np.float32) * np.float32(1.0 / 2048.0)).astype(np.float32)
