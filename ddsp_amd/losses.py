"""Losses: the multi-scale spectrogram loss (mirror of ddsp/losses.py:41-48, 102-128, 131-243).

SURVEY.md section 8(f) rank 2.  What `gin/models/ae.gin:36-41` uses - loss_type 'L1' with the magnitude and
log-magnitude terms - runs fused kernels whose spectra never leave LDS (csrc/spectral_loss.hip).  The rest of the
reference's argument space - delta_time / delta_freq / cumsum_freq terms, 'L2' and 'COSINE', the `weights` mask -
runs on spectrograms materialised in HBM, one FFT size at a time (csrc/spectral_terms.hip); the loudness term
(spectral_ops.compute_loudness, spectral_ops.py:253-324) likewise, under its own frame geometry.  `mean_difference` is the
reference's public function on the same kernels.
The call is a torch.autograd node: the gradient reaches `audio` (not `target_audio`, as a training step needs it).
"""
import ctypes

import numpy as np
import torch

from ddsp_amd import _lib
from ddsp_amd import core
from ddsp_amd import dags


def a_weighting_linear(sample_rate, n_fft):
  """10 ** (A_weighting / 10) at the bins of an n_fft-point transform: the A-curve librosa publishes (IEC 61672; clipped at -80 dB,
  f = 0) - what spectral_ops.compute_loudness multiplies the power by (spectral_ops.py:307-312).  A constant table made on the
  host in double precision, as every constant table of this library (oracle/ddsp_oracle.py::a_weighting_db restates the same
  formula for the tests)."""
  f_sq = (np.arange(n_fft // 2 + 1, dtype=np.float64) * (sample_rate / n_fft)) ** 2
  c = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
  with np.errstate(divide='ignore'):
    db = 2.0 + 20.0 * (np.log10(c[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + c[0]) - np.log10(f_sq + c[1])
                       - 0.5 * np.log10(f_sq + c[2]) - 0.5 * np.log10(f_sq + c[3]))
  return (10.0 ** (np.maximum(-80.0, db) / 10.0)).astype(np.float32)


class Loss:
  """Base class. Duck typing: losses just must implement get_losses_dict() (losses.py:41-48)."""

  def __init__(self, name):
    self.name = name

  def __call__(self, *args, **kwargs):
    return self.call(*args, **kwargs)

  def get_losses_dict(self, *args, **kwargs):
    """Returns a dictionary of losses for the model."""
    loss = self(*args, **kwargs)
    return {self.name: loss}


class LossGroup(dags.DAGLayer):
  """Compute a group of loss layers on an outputs dictionary (ddsp/losses.py:51-97): a DAG of `(loss, [input key, ...])` nodes
  -> one flat dictionary {loss name: scalar}."""

  def __init__(self, dag, **kwarg_losses):
    super().__init__(dag, **kwarg_losses)
    self.loss_names = self.module_names

  @property
  def losses(self):
    return [getattr(self, name) for name in self.loss_names]

  def call(self, outputs, **kwargs):
    dag_outputs = super().call(outputs, **kwargs)
    loss_outputs = {}
    for k in self.loss_names:
      loss_outputs.update(dag_outputs[k])
    return loss_outputs

  def get_losses_dict(self, outputs, **kwargs):
    return self(outputs, **kwargs)


_MD_MAX_LAST = 4097                    # kStMaxBins of csrc/spectral_terms.hip: the longest row a block of the term kernels holds
_md_ws = core.Workspace()


def _md_view(shape):
  """Any shape -> the [batch, rows, last axis] view the term kernels take (the last axis is what 'COSINE' reduces)."""
  shape = tuple(int(v) for v in shape)
  if len(shape) == 0:
    return 1, 1, 1
  if len(shape) == 1:
    return 1, 1, shape[0]
  rows = 1
  for v in shape[1:-1]:
    rows *= v
  return shape[0], rows, shape[-1]


def _md_raw(target, value, loss_type, weights, want_grad):
  """One call of ddsp_spectral_terms_f32 with the magnitude term alone: (loss, d loss / d value or None)."""
  dev = value.device
  loss = torch.empty((), dtype=torch.float32, device=dev)
  if value.numel() == 0:
    # tf.reduce_mean over no elements is NaN; cosine_distance's weighted mean divides safely: 0 (losses.py:118-124)
    loss.fill_(0.0 if loss_type == 'COSINE' else float('nan'))
    return loss, (torch.zeros_like(value) if want_grad else None)
  b, f, k = _md_view(value.shape)
  w = None
  if weights is not None:
    w = core.tf_float32(weights if isinstance(weights, torch.Tensor) else torch.as_tensor(weights, dtype=torch.float32))
    core.require_no_grad('mean_difference weights', w)
    if w.numel() == 1:
      w = w.reshape(1, 1, 1).contiguous()
    else:
      # `difference * weights` ('L1' / 'L2') or weights against [..., 1] ('COSINE'): materialised at the shape it multiplies
      full = tuple(value.shape[:-1]) + (1,) if loss_type == 'COSINE' else tuple(value.shape)
      if w.dim() > len(full):
        raise ValueError('weights of shape {} do not broadcast against {}'.format(tuple(w.shape), full))
      try:
        w = w.expand(full).contiguous()
      except RuntimeError:
        raise ValueError('weights of shape {} do not broadcast against {}'.format(tuple(w.shape), full))
      w = w.reshape(b, f, 1 if loss_type == 'COSINE' else k)
  if k > _MD_MAX_LAST:
    if loss_type == 'COSINE':
      raise NotImplementedError('mean_difference(COSINE) along an axis of more than {} elements is not built on the MI355X '
                                'path (got {})'.format(_MD_MAX_LAST, k))
    # 'L1' / 'L2' are means over every element: any factorisation of the element count serves (the mask, materialised, with it)
    total = b * f * k
    cols = next(c for c in range(4096, 0, -1) if total % c == 0)
    b, f, k = 1, total // cols, cols
    if w is not None and w.numel() != 1:
      w = w.reshape(b, f, k)
  acc = torch.empty((), dtype=torch.float64, device=dev)
  grad = torch.empty_like(value) if want_grad else None
  ws = _md_ws.get(core.cached_workspace_bytes('ddsp_spectral_terms_workspace_bytes', b, f), dev)
  wb, wf, wk = (int(v) for v in w.shape) if w is not None else (0, 0, 0)
  rc = _lib.load().ddsp_spectral_terms_f32(
      target.data_ptr(), value.data_ptr(), w.data_ptr() if w is not None else None, wb, wf, wk,
      grad.data_ptr() if want_grad else None, acc.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(), b, f, k,
      _lib.LOSS_TYPES[loss_type], 1.0, 0.0, 0.0, 0.0, 0.0, 1, core._stream())
  _lib.check(rc, 'ddsp_spectral_terms_f32')
  return loss, grad


class _MeanDifferenceFunction(torch.autograd.Function):
  """torch.autograd node of mean_difference.  Every loss type is symmetric in its two arguments, so d / d target is d / d value
  of the call with the arguments exchanged."""

  @staticmethod
  def forward(ctx, target, value, loss_type, weights):
    loss, grad_value = _md_raw(target, value, loss_type, weights, ctx.needs_input_grad[1])
    grad_target = _md_raw(value, target, loss_type, weights, True)[1] if ctx.needs_input_grad[0] else None
    ctx.has = (grad_target is not None, grad_value is not None)
    ctx.save_for_backward(*[g for g in (grad_target, grad_value) if g is not None])
    return loss

  @staticmethod
  def backward(ctx, grad_loss):
    saved = list(ctx.saved_tensors)
    grad_target = _scale(saved.pop(0), grad_loss) if ctx.has[0] else None
    grad_value = _scale(saved.pop(0), grad_loss) if ctx.has[1] else None
    return grad_target, grad_value, None, None


def mean_difference(target, value, loss_type='L1', weights=None):
  """Common loss functions (ddsp/losses.py:102-128): the mean of |difference * weights| ('L1') or difference**2 * weights ('L2'),
  or tf.losses.cosine_distance along the last axis ('COSINE') - on the kernels SpectralLoss's general form runs its terms on
  (ddsp_spectral_terms_f32: fixed-order fp64 sums), differentiable in `target` and `value`.

  Raises:
    ValueError: If loss_type is not an allowed value.
  """
  loss_type = loss_type.upper()
  if loss_type not in _lib.LOSS_TYPES:
    raise ValueError('Loss type ({}), must be '
                     '"L1", "L2", or "COSINE"'.format(loss_type))
  target, value = core.tf_float32(target), core.tf_float32(value)
  if target.shape != value.shape:
    core.require_no_grad('mean_difference of tensors that broadcast against each other', target, value)
    target, value = torch.broadcast_tensors(target, value)
  target, value = target.contiguous(), value.contiguous()
  if torch.is_grad_enabled() and (target.requires_grad or value.requires_grad):
    return _MeanDifferenceFunction.apply(target, value, loss_type, weights)
  return _md_raw(target, value, loss_type, weights, False)[0]


class SpectralLoss(Loss):
  """Multi-scale spectrogram loss (ddsp/losses.py:131-243)."""

  def __init__(self,
               fft_sizes=(2048, 1024, 512, 256, 128, 64),
               loss_type='L1',
               mag_weight=1.0,
               delta_time_weight=0.0,
               delta_freq_weight=0.0,
               cumsum_freq_weight=0.0,
               logmag_weight=0.0,
               loudness_weight=0.0,
               name='spectral_loss'):
    super().__init__(name=name)
    self.fft_sizes = fft_sizes
    self.loss_type = loss_type
    self.mag_weight = mag_weight
    self.delta_time_weight = delta_time_weight
    self.delta_freq_weight = delta_freq_weight
    self.cumsum_freq_weight = cumsum_freq_weight
    self.logmag_weight = logmag_weight
    self.loudness_weight = loudness_weight
    self._ws = core.Workspace()

  def call(self, target_audio, audio, weights=None):
    """Scalar loss (0-dim tensor in HBM) between two batches of audio [batch, n_samples(, 1)]."""
    if self.loss_type.upper() not in ('L1', 'L2', 'COSINE'):
      # losses.mean_difference (losses.py:102-128) raises when it is CALLED: a loss whose every weight is zero never calls it
      if max(self.mag_weight, self.delta_time_weight, self.delta_freq_weight, self.cumsum_freq_weight, self.logmag_weight,
             self.loudness_weight) > 0:
        raise ValueError('Loss type ({}), must be '
                         '"L1", "L2", or "COSINE"'.format(self.loss_type.upper()))
      return torch.zeros((), dtype=torch.float32, device=core._device())
    target_audio, audio = core.tf_float32(target_audio), core.tf_float32(audio)
    if target_audio.dim() == 3:
      target_audio = target_audio[..., 0].contiguous()
    if audio.dim() == 3:
      audio = audio[..., 0].contiguous()
    if target_audio.dim() != 2 or target_audio.shape != audio.shape:
      raise ValueError('target_audio and audio must both be [batch, n_samples], got {} and {}'.format(
          tuple(target_audio.shape), tuple(audio.shape)))
    plain_terms = (self.loss_type.upper() != 'L1' or weights is not None or self.delta_time_weight > 0 or
                   self.delta_freq_weight > 0 or self.cumsum_freq_weight > 0 or self.loudness_weight > 0)
    # frame sizes the fused 'L1' kernels take: 2^k in [16, 8192] and, since round 6, 3 * 2^k in [48, 6144] (vst_48k.gin:56 asks for
    # 6144, 3072 .. 192; 8192-point transforms one frame and one signal at a time: stft_l1_big_kernel); any other even size on the
    # plain kernels
    others = [v for v in self.fft_sizes if not self._fused_size(v)]
    if not plain_terms and others and len(others) < len(self.fft_sizes):
      # the loss is a sum over its scales (losses.py:199-236): the fused kernels for the scales they take, the plain ones for the rest
      fused_part, plain_part = self._split_by_kernel(others)
      return fused_part.call(target_audio, audio) + plain_part.call(target_audio, audio)
    general = plain_terms or bool(others)
    if general:
      weights = self._weights_tensor(weights, audio.device)
      if torch.is_grad_enabled() and audio.requires_grad:
        return _SpectralLossGeneralFunction.apply(target_audio.detach(), audio, self, weights)
      return self._general(target_audio, audio, weights, want_grad=False)[0]
    if torch.is_grad_enabled() and audio.requires_grad:
      return _SpectralLossFunction.apply(target_audio.detach(), audio, self)
    return self._forward(target_audio, audio)

  @staticmethod
  def _fused_size(v):
    v = int(v)
    return (16 <= v <= 8192 and not v & (v - 1)) or (48 <= v <= 6144 and v % 3 == 0 and not (v // 3) & (v // 3 - 1))

  def _split_by_kernel(self, others):
    key = tuple(int(v) for v in self.fft_sizes)
    if getattr(self, '_split_key', None) != key:
      kw = dict(loss_type=self.loss_type, mag_weight=self.mag_weight, logmag_weight=self.logmag_weight)
      self._split_parts = (SpectralLoss(fft_sizes=tuple(v for v in self.fft_sizes if v not in others), **kw),
                           SpectralLoss(fft_sizes=tuple(others), **kw))
      self._split_key = key
    for part in self._split_parts:           # (the weights are plain attributes a caller may change between calls)
      part.mag_weight, part.logmag_weight = self.mag_weight, self.logmag_weight
    return self._split_parts

  @staticmethod
  def _weights_tensor(weights, device):
    """`weights` of losses.mean_difference: None, a number, or a mask of rank <= 3 -> None or a [b, f, k] tensor."""
    if weights is None:
      return None
    w = core.tf_float32(weights if isinstance(weights, torch.Tensor) else torch.as_tensor(weights, dtype=torch.float32))
    core.require_no_grad('SpectralLoss weights', w)
    if w.dim() > 3:
      raise ValueError('weights must broadcast against [batch, frames, bins], got shape {}'.format(tuple(w.shape)))
    return w.reshape((1,) * (3 - w.dim()) + tuple(w.shape)).contiguous()

  def _general(self, target_audio, audio, weights, want_grad):
    """The reference's loop over FFT sizes (losses.py:199-236) on materialised spectrograms -> (loss, grad_audio)."""
    b, n = audio.shape
    lib = _lib.load()
    dev = audio.device
    loss_type = _lib.LOSS_TYPES[self.loss_type.upper()]
    term_w = [float(self.mag_weight), float(self.delta_time_weight), float(self.delta_freq_weight),
              float(self.cumsum_freq_weight), float(self.logmag_weight)]
    loss = torch.empty((), dtype=torch.float32, device=dev)
    acc = torch.empty((), dtype=torch.float64, device=dev)
    grad_audio = torch.zeros_like(audio) if want_grad else None
    if not self.fft_sizes and not self.loudness_weight > 0:
      loss.zero_()
      return loss, grad_audio
    for z, size in enumerate(self.fft_sizes):
      size = int(size)
      pow2 = 16 <= size <= 4096 and not size & (size - 1)
      other = (34 <= size <= 8190 and size % 2 == 0 and size & (size - 1)) or size == 8192       # (vst_48k.gin: 6144, 3072 .. 192; any since round 6)
      if not (pow2 or other):
        raise ValueError('fft_sizes must be powers of two in [16, 8192] or even sizes in [34, 8190] on the MI355X path (odd frames, '
                         'and frames of fewer than 34 samples that are not powers of two, are not built), got {}'.format(
                             tuple(self.fft_sizes)))
      # spectral_ops.stft (spectral_ops.py:40-45): tf.signal.stft with fft_length=None transforms the ENCLOSING power of two
      frames, bins = -(-n // (size // 4)), (1 << (size - 1).bit_length()) // 2 + 1
      wb = wf = wk = 0
      if weights is not None:
        wb, wf, wk = (int(v) for v in weights.shape)
        # the mask must broadcast against every term it meets, as `difference * weights` does in the reference
        dims = [(frames, bins)] * 5
        dims[1], dims[2] = (frames - 1, bins), (frames, bins - 1)
        for on, (tf_, tk) in zip(term_w, dims):
          if on > 0 and (wb not in (1, b) or wf not in (1, tf_) or wk not in (1, tk)):
            raise ValueError('weights of shape {} do not broadcast against a term of shape {}'.format(
                tuple(weights.shape), (b, tf_, tk)))
        if loss_type == _lib.LOSS_TYPES['COSINE'] and wk != 1:
          raise ValueError('COSINE weights must broadcast against [batch, frames, 1], got {}'.format(tuple(weights.shape)))
      target_mag = torch.empty((b, frames, bins), dtype=torch.float32, device=dev)
      mag = torch.empty_like(target_mag)
      rc = lib.ddsp_stft_mag_f32(target_audio.data_ptr(), audio.data_ptr(), target_mag.data_ptr(), mag.data_ptr(), b, n,
                                 size, core._stream())
      _lib.check(rc, 'ddsp_stft_mag_f32')
      cot = torch.empty_like(mag) if want_grad else None
      ws = self._ws.get(core.cached_workspace_bytes('ddsp_spectral_terms_workspace_bytes', b, frames), dev)
      rc = lib.ddsp_spectral_terms_f32(
          target_mag.data_ptr(), mag.data_ptr(), weights.data_ptr() if weights is not None else None, wb, wf, wk,
          cot.data_ptr() if want_grad else None, acc.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(), b, frames,
          bins, loss_type, *term_w, 1 if z == 0 else 0, core._stream())
      _lib.check(rc, 'ddsp_spectral_terms_f32')
      if want_grad:
        rc = lib.ddsp_stft_mag_backward_f32(audio.data_ptr(), cot.data_ptr(), grad_audio.data_ptr(), b, n, size,
                                            core._stream())
        _lib.check(rc, 'ddsp_stft_mag_backward_f32')
    if self.loudness_weight > 0:
      self._loudness_term(target_audio, audio, weights, loss, acc, grad_audio, first=not self.fft_sizes)
    if (self.delta_time_weight > 0 and self.loss_type.upper() in ('L1', 'L2') and
        any(-(-n // (int(size) // 4)) < 2 for size in self.fft_sizes)):
      # a clip of ONE frame at some size: the reference's delta-time term is the mean of an empty difference - NaN
      # (tf.reduce_mean over no elements, losses.py:102-128, 213-216), and so is the loss it returns.  ('COSINE' goes through
      # tf.compat.v1.losses.cosine_distance, whose weighted mean divides safely: 0 for no elements - what the kernel adds.)
      loss.fill_(float('nan'))
    return loss, grad_audio

  # spectral_ops.compute_loudness as SpectralLoss calls it (losses.py:238-242: n_fft = 2048, everything else the defaults of
  # spectral_ops.py:253-260: 16 kHz, 250 frames a second, 80 dB of range, reference 0 dB, centre padding)
  LOUDNESS_N_FFT, LOUDNESS_SAMPLE_RATE, LOUDNESS_FRAME_RATE, LOUDNESS_RANGE_DB, LOUDNESS_REF_DB = 2048, 16000, 250, 80.0, 0.0
  _a_weighting = {}

  @classmethod
  def _loudness_weighting(cls, device):
    key = str(device)
    if key not in cls._a_weighting:
      cls._a_weighting[key] = torch.as_tensor(a_weighting_linear(cls.LOUDNESS_SAMPLE_RATE, cls.LOUDNESS_N_FFT), device=device)
    return cls._a_weighting[key]

  def _loudness_term(self, target_audio, audio, weights, loss, acc, grad_audio, first):
    """loss += loudness_weight * mean_difference(compute_loudness(target), compute_loudness(audio)) (losses.py:238-242) and, if
    grad_audio is given, its gradient: |STFT| under compute_loudness's frames (ddsp_stft_frames_mag_f32), the A-weighted mean power
    in dB per frame (ddsp_loudness_from_mag_f32), the difference term on [batch, 1, frames] (ddsp_spectral_terms_f32), and back."""
    b, n = audio.shape
    lib = _lib.load()
    dev = audio.device
    n_fft = self.LOUDNESS_N_FFT
    hop = self.LOUDNESS_SAMPLE_RATE // self.LOUDNESS_FRAME_RATE
    frames, bins = 1 + n // hop, n_fft // 2 + 1
    wt = self._loudness_weighting(dev)
    loud, mags = [], []
    for x in (target_audio, audio):
      mag = torch.empty((b, frames, bins), dtype=torch.float32, device=dev)
      _lib.check(lib.ddsp_stft_frames_mag_f32(x.data_ptr(), mag.data_ptr(), b, n, n_fft, hop, n_fft // 2, frames, core._stream()),
                 'ddsp_stft_frames_mag_f32')
      ld = torch.empty((b, 1, frames), dtype=torch.float32, device=dev)
      _lib.check(lib.ddsp_loudness_from_mag_f32(mag.data_ptr(), wt.data_ptr(), ld.data_ptr(), b, frames, bins,
                                                self.LOUDNESS_RANGE_DB, self.LOUDNESS_REF_DB, core._stream()),
                 'ddsp_loudness_from_mag_f32')
      loud.append(ld); mags.append(mag)
    wb = wf = wk = 0
    if weights is not None:
      # (`weights` multiplies the [batch, frames] difference in the reference: a [batch, frames, 1]-shaped mask reads as
      #  [batch, 1, frames] here only if it is broadcast along what it does not have - kept to masks of one value per clip)
      wb, wf, wk = (int(v) for v in weights.shape)
      if wf != 1 or wk != 1:
        raise ValueError('with loudness_weight > 0 the weights mask must be one value per clip ([batch, 1, 1]), got {}'.format(
            tuple(weights.shape)))
    want_grad = grad_audio is not None
    cot = torch.empty_like(loud[1]) if want_grad else None
    ws = self._ws.get(core.cached_workspace_bytes('ddsp_spectral_terms_workspace_bytes', b, 1), dev)
    rc = lib.ddsp_spectral_terms_f32(
        loud[0].data_ptr(), loud[1].data_ptr(), weights.data_ptr() if weights is not None else None, wb, wf, wk,
        cot.data_ptr() if want_grad else None, acc.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(), b, 1, frames,
        _lib.LOSS_TYPES[self.loss_type.upper()], float(self.loudness_weight), 0.0, 0.0, 0.0, 0.0, 1 if first else 0,
        core._stream())
    _lib.check(rc, 'ddsp_spectral_terms_f32')
    if want_grad:
      grad_mag = mags[0]                                         # (the target's magnitudes are done with: their buffer)
      _lib.check(lib.ddsp_loudness_from_mag_backward_f32(mags[1].data_ptr(), wt.data_ptr(), cot.data_ptr(), grad_mag.data_ptr(),
                                                         b, frames, bins, self.LOUDNESS_RANGE_DB, self.LOUDNESS_REF_DB,
                                                         core._stream()), 'ddsp_loudness_from_mag_backward_f32')
      _lib.check(lib.ddsp_stft_frames_mag_backward_f32(audio.data_ptr(), grad_mag.data_ptr(), grad_audio.data_ptr(), b, n, n_fft,
                                                       hop, n_fft // 2, frames, core._stream()),
                 'ddsp_stft_frames_mag_backward_f32')

  def _sizes(self):
    return (ctypes.c_int * len(self.fft_sizes))(*[int(v) for v in self.fft_sizes])

  def _forward(self, target_audio, audio):
    b, n = audio.shape
    sizes = self._sizes()
    lib = _lib.load()
    nbytes = lib.ddsp_spectral_loss_workspace_bytes(b, n, sizes, len(self.fft_sizes))
    if nbytes == 0:
      raise ValueError('fft_sizes must be at most 16 sizes the fused kernels take (2**k in [16, 8192], 3 * 2**k in [48, 6144]), got {}'.format(
          tuple(self.fft_sizes)))
    ws = self._ws.get(nbytes, audio.device)
    loss = torch.empty((), dtype=torch.float32, device=audio.device)
    rc = lib.ddsp_spectral_loss_f32(target_audio.data_ptr(), audio.data_ptr(), loss.data_ptr(),
                                    ws.data_ptr(), ws.numel(), b, n, sizes, len(self.fft_sizes),
                                    float(self.mag_weight), float(self.logmag_weight), core._stream())
    _lib.check(rc, 'ddsp_spectral_loss_f32')
    return loss

  def _value_and_grad(self, target_audio, audio):
    b, n = audio.shape
    sizes = self._sizes()
    lib = _lib.load()
    nbytes = lib.ddsp_spectral_loss_workspace_bytes(b, n, sizes, len(self.fft_sizes))
    if nbytes == 0:
      raise ValueError('fft_sizes must be at most 16 sizes the fused kernels take (2**k in [16, 8192], 3 * 2**k in [48, 6144]), got {}'.format(
          tuple(self.fft_sizes)))
    ws = self._ws.get(nbytes, audio.device)
    loss = torch.empty((), dtype=torch.float32, device=audio.device)
    grad_audio = torch.empty_like(audio)
    rc = lib.ddsp_spectral_loss_value_and_grad_f32(
        target_audio.data_ptr(), audio.data_ptr(), loss.data_ptr(), grad_audio.data_ptr(), ws.data_ptr(),
        ws.numel(), b, n, sizes, len(self.fft_sizes), float(self.mag_weight), float(self.logmag_weight),
        core._stream())
    _lib.check(rc, 'ddsp_spectral_loss_value_and_grad_f32')
    return loss, grad_audio

  def _backward(self, target_audio, audio, grad_loss):
    b, n = audio.shape
    grad_loss = core.tf_float32(grad_loss).reshape(1).contiguous()
    grad_audio = torch.empty_like(audio)
    rc = _lib.load().ddsp_spectral_loss_backward_f32(
        target_audio.data_ptr(), audio.data_ptr(), grad_loss.data_ptr(), grad_audio.data_ptr(), b, n,
        self._sizes(), len(self.fft_sizes), float(self.mag_weight), float(self.logmag_weight),
        core._stream())
    _lib.check(rc, 'ddsp_spectral_loss_backward_f32')
    return grad_audio


def _scale(grad_audio, grad_loss):
  """grad_audio * (the upstream scalar dL/dloss), on ddsp_scale_f32."""
  grad_loss = core.tf_float32(grad_loss).reshape(1).contiguous()
  out = torch.empty_like(grad_audio)
  rc = _lib.load().ddsp_scale_f32(grad_audio.data_ptr(), grad_loss.data_ptr(), out.data_ptr(), grad_audio.numel(),
                                  core._stream())
  _lib.check(rc, 'ddsp_scale_f32')
  return out


class _SpectralLossFunction(torch.autograd.Function):
  """torch.autograd node of SpectralLoss.call: the gradient flows to `audio` only."""

  @staticmethod
  def forward(ctx, target_audio, audio, loss_obj):
    # value and gradient in one pass (the frame spectra are computed once for both)
    loss, grad_audio = loss_obj._value_and_grad(target_audio, audio.detach())
    ctx.save_for_backward(grad_audio)
    return loss

  @staticmethod
  def backward(ctx, grad_loss):
    (grad_audio,) = ctx.saved_tensors
    return None, _scale(grad_audio, grad_loss), None


class _SpectralLossGeneralFunction(torch.autograd.Function):
  """torch.autograd node of the general SpectralLoss: value and dL/d audio come out of the same pass."""

  @staticmethod
  def forward(ctx, target_audio, audio, loss_obj, weights):
    loss, grad_audio = loss_obj._general(target_audio, audio.detach(), weights, want_grad=True)
    ctx.save_for_backward(grad_audio)
    return loss

  @staticmethod
  def backward(ctx, grad_loss):
    (grad_audio,) = ctx.saved_tensors
    return None, _scale(grad_audio, grad_loss), None, None
