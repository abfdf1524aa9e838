"""Losses: the multi-scale spectrogram loss, forward pass (mirror of ddsp/losses.py:41-48, 131-243).

SURVEY.md section 8(f) rank 2.  Only what `gin/models/ae.gin:36-41` uses is built: loss_type 'L1'
with the magnitude and log-magnitude terms; the other weights raise NotImplementedError.
The call is a torch.autograd node: the gradient reaches `audio` (not `target_audio`).
"""
import ctypes

import torch

from ddsp_amd import _lib
from ddsp_amd import core


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
      raise ValueError('Loss type ({}), must be '
                       '"L1", "L2", or "COSINE"'.format(self.loss_type.upper()))
    unsupported = [k for k in ('delta_time_weight', 'delta_freq_weight', 'cumsum_freq_weight',
                               'loudness_weight') if getattr(self, k) > 0]
    if unsupported or self.loss_type.upper() != 'L1' or weights is not None:
      raise NotImplementedError(
          'the MI355X SpectralLoss implements loss_type="L1" with mag_weight / logmag_weight only '
          '(asked for: {})'.format(unsupported or [self.loss_type, 'weights']))
    target_audio, audio = core.tf_float32(target_audio), core.tf_float32(audio)
    if target_audio.dim() == 3:
      target_audio = target_audio[..., 0].contiguous()
    if audio.dim() == 3:
      audio = audio[..., 0].contiguous()
    if target_audio.dim() != 2 or target_audio.shape != audio.shape:
      raise ValueError('target_audio and audio must both be [batch, n_samples], got {} and {}'.format(
          tuple(target_audio.shape), tuple(audio.shape)))
    if torch.is_grad_enabled() and audio.requires_grad:
      return _SpectralLossFunction.apply(target_audio.detach(), audio, self)
    return self._forward(target_audio, audio)

  def _sizes(self):
    return (ctypes.c_int * len(self.fft_sizes))(*[int(v) for v in self.fft_sizes])

  def _forward(self, target_audio, audio):
    b, n = audio.shape
    sizes = self._sizes()
    lib = _lib.load()
    nbytes = lib.ddsp_spectral_loss_workspace_bytes(b, n, sizes, len(self.fft_sizes))
    if nbytes == 0:
      raise ValueError('fft_sizes must be at most 16 powers of two in [16, 4096], got {}'.format(
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
      raise ValueError('fft_sizes must be at most 16 powers of two in [16, 4096], got {}'.format(
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
    return None, grad_audio * grad_loss, None       # scaling by the upstream scalar: plumbing
