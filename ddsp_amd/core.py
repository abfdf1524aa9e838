"""Host-side mirror of the `ddsp.core` functions on the Harmonic / FilteredNoise path.

Same names, argument meaning and ValueErrors as the reference (`ddsp/core.py`), but the
arithmetic runs in the hand-written gfx950 kernels behind the C ABI of
`include/ddsp_amd.h`.  Tensors are `torch.Tensor`s in HBM; torch is plumbing only
(allocation, streams) - there is no torch / CPU compute fallback.
"""
import math

import numpy as np
import torch

from ddsp_amd import _lib


# --------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------
def _device():
  if not torch.cuda.is_available():
    raise _lib.DdspLibraryError(
        'ddsp_amd needs an AMD GPU (torch.cuda.is_available() is False); '
        'there is no CPU fallback.')
  return torch.device('cuda', torch.cuda.current_device())


def tf_float32(x):
  """core.tf_float32 (ddsp/core.py:31-36): anything -> contiguous fp32 tensor in HBM."""
  if type(x) is torch.Tensor and x.is_cuda and x.dtype is torch.float32 and x.is_contiguous():
    _check_current_device(x)
    return x                                  # the common case: nothing to do (host overhead matters)
  if isinstance(x, torch.Tensor):
    if not x.is_cuda:
      x = x.to(_device())
    else:
      _check_current_device(x)
    return x.to(torch.float32).contiguous()
  return torch.as_tensor(np.asarray(x, dtype=np.float32), device=_device()).contiguous()


def _check_current_device(x):
  """Every kernel is enqueued on the CURRENT device's current stream (one process per GPU is the model): a tensor
  that lives on another GPU of the process would be handed to a stream of the wrong device.  Fail loudly."""
  if x.device.index != torch._C._cuda_getDevice():
    raise ValueError('tensor on {} but the current device is cuda:{}: ddsp_amd launches on the current device - wrap the '
                     'call in `with torch.cuda.device({})`'.format(x.device, torch._C._cuda_getDevice(), x.device.index))


def _stream():
  """Raw hipStream_t of torch's current stream (the C call: torch.cuda.current_stream() alone
  costs ~4 us of Python per launch, more than the launch itself)."""
  return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream_of(device):
  """Raw handle of torch's current stream ON `device` (which need not be the current device)."""
  index = device.index if device.index is not None else torch._C._cuda_getDevice()
  return torch._C._cuda_getCurrentRawStream(index)


def require_no_grad(op, *tensors):
  """The raw kernel wrappers return tensors without a grad_fn.  Where no torch.autograd node covers an op, an input
  that requires grad must fail loudly instead of silently cutting the graph (a branch summed through Add would just
  stop training): the differentiable entries are the Processors' __call__ (Harmonic, FilteredNoise,
  Reverb, ...) and FilteredNoise.get_signal."""
  if torch.is_grad_enabled():
    for t in tensors:
      if isinstance(t, torch.Tensor) and t.requires_grad:
        raise NotImplementedError(
            '{}: an input requires grad, but this entry point has no backward pass on the MI355X path; call it '
            'under torch.no_grad() / on detached tensors, or go through the Processor __call__'.format(op))


_ws_bytes_cache = {}


def cached_workspace_bytes(fn_name, *shape):
  """Workspace size queries are pure functions of the shape: one ctypes call per new shape."""
  key = (fn_name,) + shape
  n = _ws_bytes_cache.get(key)
  if n is None:
    n = getattr(_lib.load(), fn_name)(*shape)
    _ws_bytes_cache[key] = n
  return n


class Workspace:
  """Grow-only scratch buffers in HBM, one per (device, stream): kernels enqueued on different streams never share
  scratch, and a buffer is only ever replaced by work enqueued behind its last user on the same stream (the caching
  allocator hands a freed block back to the stream it was allocated on).  A Processor instance may therefore be
  called from several streams; each stream pays for its own scratch.  The stream is the current one of the tensor's
  device, which is the current device (tf_float32 refuses tensors of another GPU: kernels launch on the current one); at most `kMaxStreams` buffers are kept, least recently used first out (short-lived streams do not pile up
  scratch, and a recycled stream handle finds a buffer that was last used on that very handle)."""
  kMaxStreams = 8

  def __init__(self):
    self._bufs = {}            # insertion order = recency (a hit is re-inserted)

  def get(self, nbytes, device):
    nbytes = max(int(nbytes), 16)
    key = (device, _stream_of(device) if device.type == 'cuda' else None)
    buf = self._bufs.pop(key, None)
    if buf is None or buf.numel() < nbytes:
      buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    self._bufs[key] = buf
    while len(self._bufs) > self.kMaxStreams:
      self._bufs.pop(next(iter(self._bufs)))
    return buf


_default_ws = Workspace()


# --------------------------------------------------------------------------------------
# dict helpers used by the DAG plumbing  (ddsp/core.py:39-129)
# --------------------------------------------------------------------------------------
def make_iterable(x):
  """Wrap in a list if not iterable, return empty list if None (ddsp/core.py:39-47)."""
  if x is None:
    return []
  if isinstance(x, (np.ndarray, torch.Tensor, dict, str)):
    return [x]
  return x if hasattr(x, '__iter__') else [x]


def to_dict(x, keys):
  """Converts list to a dictionary with supplied keys (ddsp/core.py:50-61)."""
  if isinstance(x, dict):
    return x
  x = make_iterable(x)
  if len(keys) != len(x):
    raise ValueError(f'Keys: {keys} must be the same length as {x}')
  return dict(zip(keys, x))


def nested_keys(nested_dict, delimiter='/', prefix=''):
  """Returns a flattened list of nested key strings (ddsp/core.py:76-102)."""
  keys = []
  for k, v in nested_dict.items():
    key = k if prefix == '' else f'{prefix}{delimiter}{k}'
    if not isinstance(v, dict):
      keys.append(key)
    else:
      keys += nested_keys(v, prefix=key)
  return keys


def nested_lookup(nested_key, nested_dict, delimiter='/'):
  """Returns the value of a nested dict according to "key/key/key" (ddsp/core.py:105-129)."""
  value = nested_dict
  for key in nested_key.split(delimiter):
    try:
      value = value[key]
    except KeyError:
      raise KeyError(f'Key \'{key}\' as a part of nested key \'{nested_key}\' '
                     'not found during nested dictionary lookup, out of '
                     f'available keys: {nested_keys(nested_dict)}')
  return value


# --------------------------------------------------------------------------------------
# scaling  (ddsp/core.py:386-404)
# --------------------------------------------------------------------------------------
def exp_sigmoid(x, exponent=10.0, max_value=2.0, threshold=1e-7):
  """Exponentiated sigmoid: max_value * sigmoid(x)**log(exponent) + threshold."""
  x = tf_float32(x)
  require_no_grad('core.exp_sigmoid', x)
  out = torch.empty_like(x)
  rc = _lib.load().ddsp_exp_sigmoid_f32(x.data_ptr(), out.data_ptr(), x.numel(),
                                        float(exponent), float(max_value), float(threshold),
                                        _stream())
  _lib.check(rc, 'ddsp_exp_sigmoid_f32')
  return out


def safe_divide(numerator, denominator, eps=1e-7):
  """core.safe_divide (ddsp/core.py:207-210): numerator / where(denominator == 0, eps, denominator).

  The shapes are broadcast the way TF would; a denominator that only broadcasts along the last axis ([..., 1], what
  normalize_harmonics passes) is read in place."""
  numerator, denominator = tf_float32(numerator), tf_float32(denominator)
  require_no_grad('core.safe_divide', numerator, denominator)
  shape = torch.broadcast_shapes(numerator.shape, denominator.shape)
  num = numerator.expand(shape).contiguous()
  if len(shape) == 0:
    num, den_cols, den = num.reshape(1, 1), 1, denominator.reshape(1, 1).contiguous()
  elif denominator.dim() == len(shape) and tuple(denominator.shape[:-1]) == tuple(shape[:-1]) and denominator.shape[-1] == 1 \
      and shape[-1] != 1:
    den_cols, den = 1, denominator.contiguous()
  else:
    den_cols, den = int(shape[-1]), denominator.expand(shape).contiguous()
  cols = int(shape[-1]) if len(shape) else 1
  out = torch.empty(shape, dtype=torch.float32, device=num.device)
  rc = _lib.load().ddsp_safe_divide_f32(num.data_ptr(), den.data_ptr(), out.data_ptr(), num.numel() // max(cols, 1), cols,
                                        den_cols, float(eps), _stream())
  _lib.check(rc, 'ddsp_safe_divide_f32')
  return out


def safe_log(x, eps=1e-5):
  """core.safe_log (ddsp/core.py:213-216): log of x, non-positive entries replaced by eps first."""
  x = tf_float32(x).contiguous()
  require_no_grad('core.safe_log', x)
  out = torch.empty_like(x)
  rc = _lib.load().ddsp_safe_log_f32(x.data_ptr(), out.data_ptr(), x.numel(), float(eps), _stream())
  _lib.check(rc, 'ddsp_safe_log_f32')
  return out


def get_harmonic_frequencies(frequencies, n_harmonics):
  """core.get_harmonic_frequencies (ddsp/core.py:1028-1045): [batch, :, 1] fundamentals -> [batch, :, n_harmonics]."""
  frequencies = tf_float32(frequencies).contiguous()
  require_no_grad('core.get_harmonic_frequencies', frequencies)
  if frequencies.dim() < 1 or frequencies.shape[-1] != 1:
    raise ValueError('frequencies must be [batch_size, :, 1], got {}'.format(tuple(frequencies.shape)))
  k = int(n_harmonics)
  out = torch.empty(tuple(frequencies.shape[:-1]) + (k,), dtype=torch.float32, device=frequencies.device)
  rc = _lib.load().ddsp_harmonic_frequencies_f32(frequencies.data_ptr(), out.data_ptr(), frequencies.numel(), k, _stream())
  _lib.check(rc, 'ddsp_harmonic_frequencies_f32')
  return out


def remove_above_nyquist(frequency_envelopes, amplitude_envelopes, sample_rate=16000):
  """core.remove_above_nyquist (ddsp/core.py:869-891): amplitudes of oscillators at or above sample_rate / 2 set to 0."""
  frequency_envelopes, amplitude_envelopes = tf_float32(frequency_envelopes), tf_float32(amplitude_envelopes)
  require_no_grad('core.remove_above_nyquist', frequency_envelopes, amplitude_envelopes)
  shape = torch.broadcast_shapes(frequency_envelopes.shape, amplitude_envelopes.shape)
  f = frequency_envelopes.expand(shape).contiguous()
  a = amplitude_envelopes.expand(shape).contiguous()
  out = torch.empty(shape, dtype=torch.float32, device=a.device)
  rc = _lib.load().ddsp_remove_above_nyquist_f32(f.data_ptr(), a.data_ptr(), out.data_ptr(), a.numel(), float(sample_rate),
                                                 _stream())
  _lib.check(rc, 'ddsp_remove_above_nyquist_f32')
  return out


def angular_cumsum(angular_frequency, chunk_size=1000):
  """core.angular_cumsum (ddsp/core.py:800-866): the accumulated phase in [0, 2 pi), shape [batch, time, ...].

  The scan runs in fp64 revolutions on chip (sums per chunk of 256 samples, a wrapped prefix over the chunks, the
  running phase inside each): `chunk_size` - the reference's guard against fp32 accumulation error - is accepted and has
  nothing to control; the result is the exactly accumulated phase rounded to fp32, where the reference's fp32 chunks are
  up to ~1e-2 rad off after a 4 s clip (a 3 kHz oscillator at 16 kHz).  Axis 0 is the batch and axis 1 time, as the reference's code has it; a 1-D input is
  taken as one clip."""
  del chunk_size
  w = tf_float32(angular_frequency)
  require_no_grad('core.angular_cumsum', w)
  if w.dim() < 1:
    raise ValueError('angular_frequency must be [batch, time, ...], got a scalar')
  squeeze = w.dim() == 1
  if squeeze:
    w = w[None]
  shape = tuple(w.shape)
  b, t = shape[0], shape[1]
  c = 1
  for d in shape[2:]:
    c *= int(d)
  w = w.contiguous()
  out = torch.empty(shape, dtype=torch.float32, device=w.device)
  if w.numel() == 0:
    return out[0] if squeeze else out
  lib = _lib.load()
  ws = _default_ws.get(lib.ddsp_angular_cumsum_workspace_bytes(b, t, c), w.device)
  rc = lib.ddsp_angular_cumsum_f32(w.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), b, t, c, _stream())
  _lib.check(rc, 'ddsp_angular_cumsum_f32')
  return out[0] if squeeze else out


# --------------------------------------------------------------------------------------
# resampling  (ddsp/core.py:573-714) - stand-alone; the synths evaluate it on the fly
# --------------------------------------------------------------------------------------
def resample(inputs, n_timesteps, method='linear', add_endpoint=True):
  """core.resample (ddsp/core.py:573-642): [n_frames] / [B, n_frames] / [B, n_frames, C] /
  [B, n_frames, n_freq, C] -> n_timesteps along time; methods 'nearest', 'linear', 'cubic', 'window'."""
  inputs = tf_float32(inputs)
  is_1d, is_2d, is_4d = inputs.dim() == 1, inputs.dim() == 2, inputs.dim() == 4
  if is_1d:
    inputs = inputs[None, :, None]
  elif is_2d:
    inputs = inputs[:, :, None]
  if method not in RESAMPLE_METHODS:
    raise ValueError('Method ({}) is invalid. Must be one of {}.'.format(
        method, "['nearest', 'linear', 'cubic', 'window']"))
  if method == 'window':
    outputs = upsample_with_windows(inputs, n_timesteps, add_endpoint)     # ValueError unless 3-D
  elif is_4d:
    # tf.image.resize on [B, n_frames, n_freq, C] leaves the n_freq axis alone (core.py:613-621)
    b, f, q, c = inputs.shape
    outputs = _resample_call(inputs.reshape(b, f, q * c), int(n_timesteps), method, add_endpoint)
    outputs = outputs.reshape(b, int(n_timesteps), q, c)
  elif inputs.dim() == 3:
    outputs = _resample_call(inputs, int(n_timesteps), method, add_endpoint)
  else:
    raise ValueError('resample() takes 1-D to 4-D inputs, got shape {}'.format(tuple(inputs.shape)))
  if is_1d:
    outputs = outputs[0, :, 0]
  elif is_2d:
    outputs = outputs[:, :, 0]
  return outputs


def upsample_with_windows(inputs, n_timesteps, add_endpoint=True):
  """core.upsample_with_windows (ddsp/core.py:645-714): overlapping Hann windows,
  [B, n_frames, C] -> [B, n_timesteps, C]."""
  inputs = tf_float32(inputs)
  if inputs.dim() != 3:
    raise ValueError('Upsample_with_windows() only supports 3 dimensions, '
                     'not {}.'.format(tuple(inputs.shape)))
  n_frames = int(inputs.shape[1]) + (1 if add_endpoint else 0)   # the reference appends the endpoint frame
  n_intervals = n_frames - 1
  if n_frames >= n_timesteps:
    raise ValueError('Upsample with windows cannot be used for downsampling'
                     'More input frames ({}) than output timesteps ({})'.format(
                         n_frames, n_timesteps))
  if n_intervals <= 0 or n_timesteps % n_intervals != 0.0:
    raise ValueError(
        'For upsampling, the target the number of timesteps must be divisible '
        'by the number of input frames{}. (timesteps:{}, frames:{}, '
        'add_endpoint={}).'.format('' if add_endpoint else ' - 1', n_timesteps, n_frames, add_endpoint))
  return _resample_call(inputs, int(n_timesteps), 'window', add_endpoint)


def _resample_call(inputs, n_timesteps, method, add_endpoint):
  """[B,F,C] -> [B,N,C]: 'linear' / 'window' with the endpoint go to the stand-alone kernel of the hot
  path's own forms (ddsp_resample_f32), every other combination to the general one (ddsp_resample_ex_f32)."""
  inputs = inputs.contiguous()
  b, f, c = inputs.shape
  require_no_grad('core.resample / upsample_with_windows', inputs)
  out = torch.empty((b, n_timesteps, c), dtype=torch.float32, device=inputs.device)
  if b == 0 or c == 0:
    return out
  lib = _lib.load()
  if add_endpoint and method in ('linear', 'window'):
    rc = lib.ddsp_resample_f32(inputs.data_ptr(), out.data_ptr(), b, f, n_timesteps, c,
                               1 if method == 'window' else 0, _stream())
    _lib.check(rc, 'ddsp_resample_f32')
  else:
    rc = lib.ddsp_resample_ex_f32(inputs.data_ptr(), out.data_ptr(), b, f, n_timesteps, c,
                                  _lib.RESAMPLE_METHODS[method], 1 if add_endpoint else 0, _stream())
    _lib.check(rc, 'ddsp_resample_ex_f32')
  return out


def normalize_harmonics(harmonic_distribution, f0_hz=None, sample_rate=None):
  """core.normalize_harmonics (ddsp/core.py:894-907): optional Nyquist removal, then sum-normalise."""
  harmonic_distribution = tf_float32(harmonic_distribution)
  require_no_grad('core.normalize_harmonics', harmonic_distribution, f0_hz)
  b, f, k = harmonic_distribution.shape
  bandlimit = sample_rate is not None and f0_hz is not None
  f0 = tf_float32(f0_hz) if bandlimit else torch.zeros((b, f, 1), device=harmonic_distribution.device)
  ones = torch.ones((b, f, 1), dtype=torch.float32, device=harmonic_distribution.device)
  ctl_amp, ctl_hd = torch.empty_like(ones), torch.empty_like(harmonic_distribution)
  rc = _lib.load().ddsp_harmonic_controls_f32(
      ones.data_ptr(), harmonic_distribution.data_ptr(), f0.data_ptr(), ctl_amp.data_ptr(),
      ctl_hd.data_ptr(), b, f, k, int(sample_rate) if bandlimit else 2,
      _lib.HARM_NORMALIZE_NYQUIST if bandlimit else 0, _stream())
  _lib.check(rc, 'ddsp_harmonic_controls_f32')
  return ctl_hd


def get_fft_size(frame_size, ir_size, power_of_2=True):
  """core.get_fft_size (ddsp/core.py:1317-1335); host arithmetic, kept for API completeness."""
  convolved_frame_size = ir_size + frame_size - 1
  if power_of_2:
    return int(2**np.ceil(np.log2(convolved_frame_size)))
  from scipy import fftpack
  return int(fftpack.next_fast_len(convolved_frame_size))


def crop_and_compensate_delay(audio, audio_size, ir_size, padding, delay_compensation):
  """core.crop_and_compensate_delay (ddsp/core.py:1338-1379): a slice, no arithmetic."""
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


def oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=16000,
                    sum_sinusoids=True, use_angular_cumsum=False):
  """core.oscillator_bank (ddsp/core.py:912-962) on [batch, n_samples, n_sinusoids] envelopes.

  The phase scan runs in fp64 revolutions on chip, so both settings of `use_angular_cumsum`
  give the same (more exact) result.
  """
  del use_angular_cumsum
  frequency_envelopes = tf_float32(frequency_envelopes)
  amplitude_envelopes = tf_float32(amplitude_envelopes)
  require_no_grad('core.oscillator_bank', frequency_envelopes, amplitude_envelopes)
  if frequency_envelopes.dim() != 3 or frequency_envelopes.shape != amplitude_envelopes.shape:
    raise ValueError('frequency and amplitude envelopes must both be [batch, n_samples, n_sinusoids]'
                     ', got {} and {}'.format(tuple(frequency_envelopes.shape),
                                              tuple(amplitude_envelopes.shape)))
  b, n, k = frequency_envelopes.shape
  lib = _lib.load()
  out = torch.empty((b, n) if sum_sinusoids else (b, n, k), dtype=torch.float32,
                    device=frequency_envelopes.device)
  ws = _default_ws.get(lib.ddsp_oscillator_bank_workspace_bytes(b, n, k), frequency_envelopes.device)
  rc = lib.ddsp_oscillator_bank_f32(frequency_envelopes.data_ptr(), amplitude_envelopes.data_ptr(),
                                    out.data_ptr(), ws.data_ptr(), ws.numel(), b, n, k,
                                    int(sample_rate), 1 if sum_sinusoids else 0, _stream())
  _lib.check(rc, 'ddsp_oscillator_bank_f32')
  return out


# --------------------------------------------------------------------------------------
# harmonic synthesis  (ddsp/core.py:1048-1111)
# --------------------------------------------------------------------------------------
RESAMPLE_METHODS = ['nearest', 'linear', 'cubic', 'window']


def _check_amp_method(method, n_frames, n_samples):
  """The argument checks of core.resample / upsample_with_windows (core.py:633-634, 677-693)."""
  if method not in RESAMPLE_METHODS:
    raise ValueError('Method ({}) is invalid. Must be one of {}.'.format(
        method, "['nearest', 'linear', 'cubic', 'window']"))
  if method == 'window':
    if n_frames + 1 >= n_samples:
      raise ValueError('Upsample with windows cannot be used for downsampling'
                       'More input frames ({}) than output timesteps ({})'.format(
                           n_frames + 1, n_samples))
    if n_samples % n_frames != 0:
      raise ValueError(
          'For upsampling, the target the number of timesteps must be divisible '
          'by the number of input frames{}. (timesteps:{}, frames:{}, '
          'add_endpoint={}).'.format('', n_samples, n_frames + 1, True))


def _on_closed_form_kernels(method, n_frames, n_samples):
  """True where the synthesis kernels' closed forms apply ('window' / 'linear' envelopes, n_samples a
  multiple of n_frames); other argument combinations follow the reference's own chain of materialised
  envelopes (_harmonic_synthesis_materialised)."""
  return method in ('window', 'linear') and n_samples % n_frames == 0


def _harmonic_flags(scale, normalize, amp_resample_method, use_angular_cumsum):
  flags = 0
  if scale:
    flags |= _lib.HARM_SCALE_EXP_SIGMOID
  if normalize:
    flags |= _lib.HARM_NORMALIZE_NYQUIST
  if amp_resample_method == 'linear':
    flags |= _lib.HARM_AMP_LINEAR
  if use_angular_cumsum:
    flags |= _lib.HARM_ANGULAR_CUMSUM
  return flags


def _broadcast_batch(*tensors):
  """[1, frames, channels] inputs beside [batch, ..] ones: the reference's elementwise products broadcast them over the batch
  (its own test does: ddsp/core_test.py:549-590 passes shifts and a distribution of batch 1 with frequencies of batch 2).
  None passes through; anything else is left for the shape checks."""
  batches = [int(t.shape[0]) for t in tensors if t is not None and t.dim() == 3]
  b = max(batches) if batches else 1
  return tuple(t.expand(b, -1, -1).contiguous() if t is not None and t.dim() == 3 and t.shape[0] == 1 and b > 1 else t
               for t in tensors)


def _check_harmonic_shapes(amplitudes, harmonic_distribution, f0_hz):
  if harmonic_distribution.dim() != 3:
    raise ValueError('harmonic_distribution must be [batch, n_frames, n_harmonics], got {}'
                     .format(tuple(harmonic_distribution.shape)))
  b, f, _ = harmonic_distribution.shape
  for name, t in (('amplitudes', amplitudes), ('f0_hz', f0_hz)):
    if tuple(t.shape) != (b, f, 1):
      raise ValueError('{} must have shape [{}, {}, 1], got {}'.format(
          name, b, f, tuple(t.shape)))
  return b, f, harmonic_distribution.shape[2]


def harmonic_synthesis(frequencies, amplitudes, harmonic_shifts=None,
                       harmonic_distribution=None, n_samples=64000, sample_rate=16000,
                       amp_resample_method='window', use_angular_cumsum=False,
                       workspace=None, tf_op_order=False):
  """core.harmonic_synthesis (ddsp/core.py:1048-1111): frame-rate controls -> audio [batch, n_samples].

  'window' / 'linear' amplitude envelopes without harmonic_shifts (everything synths.Harmonic and the
  shipped gin configs ask for) run the closed-form synthesis kernels; harmonic_shifts, 'nearest' / 'cubic'
  envelopes and n_samples that is not a multiple of n_frames follow the reference's own chain
  (frame-rate tensors -> resample -> oscillator_bank) on materialised [B, N, K] envelopes.
  tf_op_order=True (extension, validation only) runs the slow kernel that follows the reference's
  fp32 op order exactly, sequential phase accumulation included.
  """
  frequencies, amplitudes = tf_float32(frequencies), tf_float32(amplitudes)
  if harmonic_shifts is not None:
    harmonic_shifts = tf_float32(harmonic_shifts)
  require_no_grad('core.harmonic_synthesis (Harmonic.__call__ is the differentiable entry)', frequencies, amplitudes,
                  harmonic_shifts, harmonic_distribution)
  if harmonic_distribution is not None:
    harmonic_distribution = tf_float32(harmonic_distribution)
  frequencies, amplitudes, harmonic_shifts, harmonic_distribution = _broadcast_batch(frequencies, amplitudes, harmonic_shifts,
                                                                                     harmonic_distribution)
  if harmonic_distribution is None:
    if harmonic_shifts is not None:                          # n_harmonics from the shifts (core.py:1082-1084)
      return _harmonic_synthesis_materialised(frequencies, amplitudes, harmonic_shifts, None, int(n_samples),
                                              int(sample_rate), amp_resample_method, use_angular_cumsum)
    harmonic_distribution = torch.ones_like(amplitudes)
  harmonic_distribution = tf_float32(harmonic_distribution)
  b, f, k = _check_harmonic_shapes(amplitudes, harmonic_distribution, frequencies)
  _check_amp_method(amp_resample_method, f, int(n_samples))
  if harmonic_shifts is not None or not _on_closed_form_kernels(amp_resample_method, f, int(n_samples)):
    if tf_op_order:
      raise NotImplementedError('tf_op_order covers the closed-form kernels\' argument space only')
    return _harmonic_synthesis_materialised(frequencies, amplitudes, harmonic_shifts, harmonic_distribution,
                                            int(n_samples), int(sample_rate), amp_resample_method,
                                            use_angular_cumsum)
  lib = _lib.load()
  audio = torch.empty((b, int(n_samples)), dtype=torch.float32, device=amplitudes.device)
  if tf_op_order:
    rc = lib.ddsp_harmonic_signal_tf_order_f32(
        amplitudes.data_ptr(), harmonic_distribution.data_ptr(), frequencies.data_ptr(),
        audio.data_ptr(), b, f, k, int(n_samples), int(sample_rate),
        _harmonic_flags(False, False, amp_resample_method, use_angular_cumsum), _stream())
    _lib.check(rc, 'ddsp_harmonic_signal_tf_order_f32')
    return audio
  nbytes = lib.ddsp_harmonic_workspace_bytes(b, f, k, int(n_samples))
  ws = (workspace or _default_ws).get(nbytes, amplitudes.device)
  rc = lib.ddsp_harmonic_signal_f32(
      amplitudes.data_ptr(), harmonic_distribution.data_ptr(), frequencies.data_ptr(),
      audio.data_ptr(), ws.data_ptr(), ws.numel(), b, f, k, int(n_samples), int(sample_rate),
      _harmonic_flags(False, False, amp_resample_method, use_angular_cumsum), _stream())
  _lib.check(rc, 'ddsp_harmonic_signal_f32')
  return audio


def _harmonic_synthesis_materialised(frequencies, amplitudes, harmonic_shifts, harmonic_distribution,
                                     n_samples, sample_rate, amp_resample_method, use_angular_cumsum):
  """The reference's chain op for op (ddsp/core.py:1080-1111), each op a kernel of this library:
  harmonic_frequencies = f0 [1..K] (1 + shifts), harmonic_amplitudes = amplitudes * distribution
  (ddsp_harmonic_envelopes_f32), resample both to [B, N, K], oscillator_bank."""
  if amplitudes.dim() != 3 or amplitudes.shape[2] != 1 or frequencies.shape != amplitudes.shape:
    raise ValueError('frequencies and amplitudes must both be [batch, n_frames, 1], got {} and {}'.format(
        tuple(frequencies.shape), tuple(amplitudes.shape)))
  b, f, _ = amplitudes.shape
  per_harmonic = harmonic_distribution if harmonic_distribution is not None else harmonic_shifts
  if per_harmonic.dim() != 3 or tuple(per_harmonic.shape[:2]) != (b, f):
    raise ValueError('per-harmonic controls must be [{}, {}, n_harmonics], got {}'.format(
        b, f, tuple(per_harmonic.shape)))
  k = int(per_harmonic.shape[2])
  if harmonic_shifts is not None and tuple(harmonic_shifts.shape) != (b, f, k):
    raise ValueError('harmonic_shifts must be [{}, {}, {}], got {}'.format(b, f, k, tuple(harmonic_shifts.shape)))
  if amp_resample_method not in RESAMPLE_METHODS:
    raise ValueError('Method ({}) is invalid. Must be one of {}.'.format(
        amp_resample_method, "['nearest', 'linear', 'cubic', 'window']"))
  dev = amplitudes.device
  harmonic_frequencies = torch.empty((b, f, k), dtype=torch.float32, device=dev)
  harmonic_amplitudes = torch.empty((b, f, k), dtype=torch.float32, device=dev)
  rc = _lib.load().ddsp_harmonic_envelopes_f32(
      amplitudes.data_ptr(), harmonic_distribution.data_ptr() if harmonic_distribution is not None else None,
      frequencies.data_ptr(), harmonic_shifts.data_ptr() if harmonic_shifts is not None else None,
      harmonic_frequencies.data_ptr(), harmonic_amplitudes.data_ptr(), b, f, k, _stream())
  _lib.check(rc, 'ddsp_harmonic_envelopes_f32')
  frequency_envelopes = resample(harmonic_frequencies, n_samples)
  amplitude_envelopes = resample(harmonic_amplitudes, n_samples, method=amp_resample_method)
  return oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate=sample_rate,
                         use_angular_cumsum=use_angular_cumsum)


def harmonic_oscillator_bank(frequency, amplitude_envelopes, initial_phase=None, sample_rate=16000,
                             use_angular_cumsum=True, workspace=None):
  """core.harmonic_oscillator_bank (ddsp/core.py:966-1025) -> (audio [batch, n_samples], final_phase [batch, 1, 1]).

  Audio-rate inputs: `frequency` [batch, n_samples, 1] (one fundamental per clip), `amplitude_envelopes`
  [batch, n_samples, n_sinusoids], `initial_phase` [batch, 1, 1] radians.  audio = sum_k A_k sin((k+1) phase),
  phase = cumsum(2 pi f / sr) + initial_phase, no Nyquist mask (as the reference); final_phase is the phase of the
  last sample, with the scan's part wrapped to [0, 2 pi) when use_angular_cumsum (the reference's default here).
  """
  frequency, amplitude_envelopes = tf_float32(frequency), tf_float32(amplitude_envelopes)
  require_no_grad('core.harmonic_oscillator_bank', frequency, amplitude_envelopes, initial_phase)
  if amplitude_envelopes.dim() != 3 or frequency.dim() != 3 or frequency.shape[2] != 1 or \
      tuple(frequency.shape[:2]) != tuple(amplitude_envelopes.shape[:2]):
    raise ValueError('frequency must be [batch, n_samples, 1] and amplitude_envelopes [batch, n_samples, n_sinusoids], '
                     'got {} and {}'.format(tuple(frequency.shape), tuple(amplitude_envelopes.shape)))
  b, n, k = amplitude_envelopes.shape
  dev = amplitude_envelopes.device
  if initial_phase is not None:
    initial_phase = tf_float32(initial_phase).reshape(-1)
    if initial_phase.numel() != b:
      raise ValueError('initial_phase must be [batch, 1, 1], got {} values for batch {}'.format(
          initial_phase.numel(), b))
    initial_phase = initial_phase.contiguous()
  audio = torch.empty((b, n), dtype=torch.float32, device=dev)
  final_phase = torch.empty((b, 1, 1), dtype=torch.float32, device=dev)
  if b == 0 or n == 0:
    return audio, final_phase
  lib = _lib.load()
  ws = (workspace or _default_ws).get(cached_workspace_bytes('ddsp_harmonic_oscillator_bank_workspace_bytes', b, n), dev)
  rc = lib.ddsp_harmonic_oscillator_bank_f32(
      frequency.data_ptr(), amplitude_envelopes.data_ptr(),
      initial_phase.data_ptr() if initial_phase is not None else None, audio.data_ptr(), final_phase.data_ptr(),
      ws.data_ptr(), ws.numel(), b, n, k, int(sample_rate), 1 if use_angular_cumsum else 0, _stream())
  _lib.check(rc, 'ddsp_harmonic_oscillator_bank_f32')
  return audio, final_phase


def streaming_harmonic_synthesis(frequencies, amplitudes, harmonic_distribution=None,
                                 initial_phase=None, n_samples=64000, sample_rate=16000,
                                 amp_resample_method='linear', workspace=None):
  """core.streaming_harmonic_synthesis (ddsp/core.py:1114-1164) -> (audio, final_phase).

  One chunk of audio [batch, n_samples] from frame-wise controls, with the fundamental's phase
  carried in (`initial_phase` [batch, 1, 1], radians) and out (`final_phase` [batch, 1, 1] =
  (sum of omega mod 2 pi) + initial_phase, as harmonic_oscillator_bank returns it with its
  default angular cumsum, core.py:1002-1012).  No audio-rate Nyquist mask, as in the reference.
  """
  frequencies, amplitudes = tf_float32(frequencies), tf_float32(amplitudes)
  flags = _lib.HARM_AMP_LINEAR if amp_resample_method == 'linear' else 0
  require_no_grad('core.streaming_harmonic_synthesis', frequencies, amplitudes, harmonic_distribution, initial_phase)
  n = int(n_samples)
  if harmonic_distribution is None and amplitudes.dim() == 3 and amplitudes.shape[-1] != 1:
    # amplitudes given PER HARMONIC [batch, n_frames, n_harmonics] and no distribution: the reference then takes them as the
    # harmonic amplitudes themselves (core.py:1150-1151, `harmonic_amplitudes = amplitudes` - its docstring says [.., 1], its code
    # takes any last axis; tools/fuzz_parity.py found this entry raising): the chain on materialised envelopes
    if frequencies.dim() != 3 or frequencies.shape[-1] != 1 or frequencies.shape[:2] != amplitudes.shape[:2]:
      raise ValueError('frequencies must have shape [{}, {}, 1], got {}'.format(amplitudes.shape[0], amplitudes.shape[1],
                                                                               tuple(frequencies.shape)))
    _check_amp_method(amp_resample_method, int(amplitudes.shape[1]), n)
    frequency_envelope = resample(frequencies, n)
    amplitude_envelopes = resample(amplitudes.contiguous(), n, method=amp_resample_method)
    return harmonic_oscillator_bank(frequency_envelope, amplitude_envelopes, initial_phase, sample_rate=sample_rate,
                                    workspace=workspace)
  if harmonic_distribution is None:
    harmonic_distribution = torch.ones_like(amplitudes)
    flags |= _lib.HARM_INPUTS_ARE_AMPLITUDES
  harmonic_distribution = tf_float32(harmonic_distribution)
  frequencies, amplitudes, harmonic_distribution = _broadcast_batch(frequencies, amplitudes, harmonic_distribution)
  b, f, k = _check_harmonic_shapes(amplitudes, harmonic_distribution, frequencies)
  _check_amp_method(amp_resample_method, f, n)
  if amp_resample_method not in ('linear', 'window') or n % f:
    # the closed-form kernel knows the two envelopes the shipped configs use on whole frames; everything else follows
    # the reference's own chain on materialised envelopes (core.py:1144-1164): normalize_harmonics, resample both,
    # harmonic_oscillator_bank - each a kernel of this library
    if flags & _lib.HARM_INPUTS_ARE_AMPLITUDES:
      harmonic_amplitudes = amplitudes
    else:
      hd_norm = normalize_harmonics(harmonic_distribution, frequencies, sample_rate)
      # amplitudes * distribution on ddsp_harmonic_envelopes_f32 (its harmonic-frequency output is not needed here)
      harmonic_amplitudes = torch.empty_like(hd_norm)
      unused = torch.empty_like(hd_norm)
      rc = _lib.load().ddsp_harmonic_envelopes_f32(
          amplitudes.data_ptr(), hd_norm.data_ptr(), frequencies.data_ptr(), None, unused.data_ptr(),
          harmonic_amplitudes.data_ptr(), b, f, k, _stream())
      _lib.check(rc, 'ddsp_harmonic_envelopes_f32')
    frequency_envelope = resample(frequencies, n)
    amplitude_envelopes = resample(harmonic_amplitudes, n, method=amp_resample_method)
    return harmonic_oscillator_bank(frequency_envelope, amplitude_envelopes, initial_phase, sample_rate=sample_rate,
                                    workspace=workspace)
  if n % f:
    raise ValueError('streaming_harmonic_synthesis needs n_samples ({}) to be a multiple of '
                     'n_frames ({}) on the MI355X path.'.format(n, f))
  dev = amplitudes.device
  if initial_phase is not None:
    initial_phase = tf_float32(initial_phase).reshape(-1)
    if initial_phase.numel() != b:
      raise ValueError('initial_phase must be [batch, 1, 1], got {} values for batch {}'.format(
          initial_phase.numel(), b))
    initial_phase = initial_phase.contiguous()
  lib = _lib.load()
  audio = torch.empty((b, n), dtype=torch.float32, device=dev)
  final_phase = torch.empty((b, 1, 1), dtype=torch.float32, device=dev)
  ws = (workspace or _default_ws).get(cached_workspace_bytes('ddsp_harmonic_workspace_bytes', b, f, k, n), dev)
  rc = lib.ddsp_harmonic_streaming_f32(
      amplitudes.data_ptr(), harmonic_distribution.data_ptr(), frequencies.data_ptr(),
      initial_phase.data_ptr() if initial_phase is not None else None, audio.data_ptr(),
      final_phase.data_ptr(), ws.data_ptr(), ws.numel(), b, f, k, n, int(sample_rate), flags,
      _stream())
  _lib.check(rc, 'ddsp_harmonic_streaming_f32')
  return audio, final_phase


# --------------------------------------------------------------------------------------
# time-varying FIR  (ddsp/core.py:1382-1565, 1628-1655)
# --------------------------------------------------------------------------------------
def frequency_impulse_response(magnitudes, window_size=0):
  """core.frequency_impulse_response: [B,F,M] (or [B,M]) magnitudes -> causal windowed IR."""
  magnitudes = tf_float32(magnitudes)
  squeeze = magnitudes.dim() == 2
  require_no_grad('core.frequency_impulse_response (FilteredNoise is the differentiable entry)', magnitudes)
  if squeeze:
    magnitudes = magnitudes[:, None, :].contiguous()
  b, f, m = magnitudes.shape
  lib = _lib.load()
  size = lib.ddsp_fir_size(m, int(window_size))
  _lib.check(min(size, 0), 'ddsp_fir_size')
  ir = torch.empty((b, f, size), dtype=torch.float32, device=magnitudes.device)
  rc = lib.ddsp_frequency_impulse_response_f32(magnitudes.data_ptr(), ir.data_ptr(), b, f, m,
                                               int(window_size), _stream())
  _lib.check(rc, 'ddsp_frequency_impulse_response_f32')
  return ir[:, 0, :] if squeeze else ir


def apply_window_to_impulse_response(impulse_response, window_size=0, causal=False):
  """core.apply_window_to_impulse_response (ddsp/core.py:1477-1531): Hann-window a series of zero-phase (`causal`: causal)
  impulse responses [..., ir_size] and return them causal, cropped to the window when it is shorter."""
  impulse_response = tf_float32(impulse_response)
  require_no_grad('core.apply_window_to_impulse_response', impulse_response)
  if impulse_response.dim() < 1 or impulse_response.shape[-1] < 1:
    raise ValueError('impulse_response must be [..., ir_size], got {}'.format(tuple(impulse_response.shape)))
  l0 = int(impulse_response.shape[-1])
  lib = _lib.load()
  l = lib.ddsp_window_impulse_response_size(l0, int(window_size))
  flat = impulse_response.reshape(-1, l0).contiguous()
  out = torch.empty((flat.shape[0], l), dtype=torch.float32, device=flat.device)
  if flat.shape[0]:
    rc = lib.ddsp_apply_window_to_impulse_response_f32(flat.data_ptr(), out.data_ptr(), flat.shape[0], l0, int(window_size),
                                                       1 if causal else 0, _stream())
    _lib.check(rc, 'ddsp_apply_window_to_impulse_response_f32')
  return out.reshape(tuple(impulse_response.shape[:-1]) + (l,))


def _crop_range(audio_size, n_ir_frames, ir_size, padding, delay_compensation):
  """(start as requested, first kept index, number of kept samples) of the slice
  crop_and_compensate_delay (ddsp/core.py:1338-1379) takes from the overlap-added FFT frames of
  fft_convolve - python slice semantics included: audio[:, start:-end] is empty when the FFT size leaves
  nothing to crop at the end (end <= 0) or start is negative (ir_size <= 2 with automatic compensation)."""
  if padding == 'valid':
    crop_size = ir_size + audio_size - 1
  elif padding == 'same':
    crop_size = audio_size
  else:
    raise ValueError('Padding must be \'valid\' or \'same\', instead '
                     'of {}.'.format(padding))
  frame_size = int(np.ceil(audio_size / n_ir_frames))
  total_size = (n_ir_frames - 1) * frame_size + get_fft_size(frame_size, ir_size, power_of_2=True)
  start = (ir_size - 1) // 2 - 1 if delay_compensation < 0 else int(delay_compensation)
  end = (total_size - crop_size) - start
  kept = range(total_size)[start:-end]
  return start, (kept[0] if len(kept) else 0), len(kept)


def fft_convolve(audio, impulse_response, padding='same', delay_compensation=-1):
  """core.fft_convolve (ddsp/core.py:1382-1473), evaluated as the equivalent direct time-varying FIR
  (one long impulse response: as a partitioned FFT convolution), cropped as
  crop_and_compensate_delay (:1338-1379) crops the overlap-added FFT frames."""
  audio, impulse_response = tf_float32(audio), tf_float32(impulse_response)
  if audio.dim() != 2:
    raise ValueError('audio must be [batch, audio_timesteps], got {}'.format(tuple(audio.shape)))
  require_no_grad('core.fft_convolve (effects.Reverb / FilteredNoise are the differentiable entries)', audio,
                  impulse_response)
  batch_size, audio_size = audio.shape
  if impulse_response.dim() == 2:
    impulse_response = impulse_response[:, None, :].contiguous()
  batch_size_ir, n_ir_frames, ir_size = impulse_response.shape
  if batch_size_ir == 1 and batch_size > 1:
    batch_size_ir_eff = batch_size        # broadcast (core.py:1433-1434), done in-kernel
  else:
    batch_size_ir_eff = batch_size_ir
  if batch_size != batch_size_ir_eff:
    raise ValueError('Batch size of audio ({}) and impulse response ({}) must '
                     'be the same.'.format(batch_size, batch_size_ir))
  _check_frames(audio_size, n_ir_frames)
  start_requested, start, n_out = _crop_range(audio_size, n_ir_frames, ir_size, padding, delay_compensation)
  if n_out == 0:
    return torch.empty((batch_size, 0), dtype=torch.float32, device=audio.device)
  if n_ir_frames == 1 and ir_size > LONG_IR_TAPS:
    # one long IR (a reverb): partitioned FFT convolution instead of the direct FIR
    return fft_convolve_long(audio, impulse_response[:, 0, :], delay=start, n_out=n_out)
  out = torch.empty((batch_size, n_out), dtype=torch.float32, device=audio.device)
  lib = _lib.load()
  rc = _lib.ERR_UNSUPPORTED
  if n_out == audio_size and start == start_requested:
    rc = lib.ddsp_fft_convolve_same_f32(
        audio.data_ptr(), impulse_response.data_ptr(), out.data_ptr(), batch_size, batch_size_ir,
        n_ir_frames, ir_size, audio_size, int(delay_compensation), _stream())
    if rc != _lib.ERR_UNSUPPORTED:     # a time-varying IR beyond the tiled kernel's LDS budget takes the general one
      _lib.check(rc, 'ddsp_fft_convolve_same_f32')
  if rc == _lib.ERR_UNSUPPORTED:
    rc = lib.ddsp_fft_convolve_f32(
        audio.data_ptr(), impulse_response.data_ptr(), out.data_ptr(), batch_size, batch_size_ir,
        n_ir_frames, ir_size, audio_size, n_out, start, _stream())
    _lib.check(rc, 'ddsp_fft_convolve_f32')
  return out


LONG_IR_TAPS = 1024       # single-frame IRs longer than this take the FFT path


def fft_convolve_long(audio, impulse_response, delay=0, add_dry=False, mask_tap0=False,
                      workspace=None, n_out=None, reverse_audio=False, reverse_ir=False,
                      reverse_out=False, zero_out0=False):
  """out[b, n] = sum_k ir[b, k] audio[b, n + delay - k] (+ audio[b, n]) for one IR per row.

  The single-frame case of core.fft_convolve (ddsp/core.py:1428-1430, padding='same',
  crop start = delay) as effects.Reverb uses it (ddsp/effects.py:113-117), evaluated by
  ddsp_fft_convolve_long_ex_f32.  impulse_response [batch or 1, ir_size].  n_out (default
  n_samples) outputs per row; the reverse_* switches read an input / write the output with its
  index reversed (the correlations of the Reverb backward pass).
  """
  audio, impulse_response = tf_float32(audio), tf_float32(impulse_response)
  if audio.dim() != 2 or impulse_response.dim() != 2:
    raise ValueError('audio must be [batch, n_samples] and impulse_response [batch, ir_size], got '
                     '{} and {}'.format(tuple(audio.shape), tuple(impulse_response.shape)))
  b, n = audio.shape
  b_ir, l = impulse_response.shape
  if b_ir != b and b_ir != 1:
    raise ValueError('Batch size of audio ({}) and impulse response ({}) must '
                     'be the same.'.format(b, b_ir))
  if delay < 0:
    raise ValueError('delay must be >= 0, got {}'.format(delay))
  n_out = n if n_out is None else int(n_out)
  lib = _lib.load()
  nbytes = cached_workspace_bytes('ddsp_fft_convolve_long_ex_workspace_bytes', b, b_ir, n, l, n_out,
                                  int(delay))
  ws = (workspace if workspace is not None else Workspace()).get(nbytes, audio.device)
  out = torch.empty((b, n_out), dtype=torch.float32, device=audio.device)
  flags = ((_lib.CONV_ADD_DRY if add_dry else 0) | (_lib.CONV_MASK_TAP0 if mask_tap0 else 0) |
           (_lib.CONV_REVERSE_AUDIO if reverse_audio else 0) | (_lib.CONV_REVERSE_IR if reverse_ir else 0) |
           (_lib.CONV_REVERSE_OUT if reverse_out else 0) | (_lib.CONV_ZERO_OUT0 if zero_out0 else 0))
  rc = lib.ddsp_fft_convolve_long_ex_f32(audio.data_ptr(), impulse_response.data_ptr(), out.data_ptr(),
                                         ws.data_ptr(), ws.numel(), b, b_ir, n, l, n_out, int(delay),
                                         flags, _stream())
  if rc == -3:
    raise NotImplementedError('the FFT convolution holds fewer than 2**28 taps per row and at most 65535 rows, got {} taps, '
                              '{} rows'.format(l, b))
  _lib.check(rc, 'ddsp_fft_convolve_long_ex_f32')
  return out


def _check_frames(audio_size, n_ir_frames):
  """The frame-count check of core.fft_convolve (ddsp/core.py:1446-1457)."""
  frame_size = int(np.ceil(audio_size / n_ir_frames))
  n_audio_frames = int(math.ceil(audio_size / frame_size))
  if n_audio_frames != n_ir_frames:
    raise ValueError(
        'Number of Audio frames ({}) and impulse response frames ({}) do not '
        'match. For small hop size = ceil(audio_size / n_ir_frames), '
        'number of impulse response frames must be a multiple of the audio '
        'size.'.format(n_audio_frames, n_ir_frames))


def frequency_filter(audio, magnitudes, window_size=0, padding='same'):
  """core.frequency_filter = fft_convolve(audio, frequency_impulse_response(magnitudes))."""
  impulse_response = frequency_impulse_response(magnitudes, window_size=window_size)
  return fft_convolve(audio, impulse_response, padding=padding)


def prepare(n_harmonics=0, n_noise_bands=0, window_size=0):
  """Make the constant operand tables of the matrix-core kernels for these shapes on the current device now (C ABI
  ddsp_prepare): the one thing in the library that allocates and copies synchronously, once per device and shape, on first
  use otherwise - which must not happen inside a HIP-graph capture."""
  _lib.check(_lib.load().ddsp_prepare(int(n_harmonics), int(n_noise_bands), int(window_size)), 'ddsp_prepare')


def uniform_noise(batch_size, n_samples, seed=0, batch_offset=0, noise_bits=23):
  """The on-chip stand-in for tf.random.uniform([B, N], -1, 1) (ddsp/synths.py:192-193): what FilteredNoise generates
  for itself: the 2^23 levels of an fp32 uniform (noise_bits=23, the default) or 2048 levels (noise_bits=11)."""
  if noise_bits not in (11, 23):
    raise ValueError('noise_bits must be 11 or 23, got {!r}'.format(noise_bits))
  out = torch.empty((int(batch_size), int(n_samples)), dtype=torch.float32, device=_device())
  rc = _lib.load().ddsp_uniform_noise_ex_f32(out.data_ptr(), int(batch_size), int(n_samples),
                                             int(seed), int(batch_offset), int(noise_bits), _stream())
  _lib.check(rc, 'ddsp_uniform_noise_ex_f32')
  return out
