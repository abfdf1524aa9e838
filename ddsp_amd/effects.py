"""Effects: the convolutional reverb (mirror of ddsp/effects.py:27-117).

SURVEY.md section 8(f) rank 1.  The reference convolves with one 131 072-point FFT per clip; here
`core.fft_convolve_long` runs a partitioned overlap-save convolution with LDS-resident FFTs
(csrc/reverb.hip).  FilteredNoiseReverb and FIRFilter are compositions of kernels that exist;
ExpDecayReverb's impulse response has its own small kernels (csrc/general.hip); ModDelay is not built.
"""
import torch

from ddsp_amd import _lib
from ddsp_amd import core
from ddsp_amd import processors

tf_float32 = core.tf_float32


class Reverb(processors.Processor):
  """Convolutional (FIR) reverb (ddsp/effects.py:27-117)."""

  _variable_names = ('_ir',)

  def __init__(self, trainable=False, reverb_length=48000, add_dry=True, name='reverb'):
    """Takes neural network outputs directly as the impulse response.

    Args:
      trainable: hold the impulse response as a single tensor for the entire dataset
        (the reference's tf.Variable, effects.py:71-80); set `rev._ir.requires_grad_(True)` to
        train it - get_signal is a torch.autograd node.
      reverb_length: length of the impulse response; only used if trainable=True.
      add_dry: add the dry signal to the reverberated signal on output.
      name: name of the processor module.
    """
    super().__init__(name=name, trainable=trainable)
    self._reverb_length = reverb_length
    self._add_dry = add_dry
    self._ir = None
    self.built = False
    self._ws = core.Workspace()
    self._ws_bwd = core.Workspace()

  def _ir_2d(self, ir):
    """[L] / [B, L] / [B, L, 1] -> contiguous [B or 1, L] (effects.py:50-57, 62-66)."""
    if ir.dim() == 1:
      ir = ir[None, :]
    if ir.dim() == 3:
      ir = ir[:, :, 0]
    if ir.dim() != 2:
      raise ValueError('ir must be [ir_size], [batch, ir_size] or [batch, ir_size, 1], got {}'.format(
          tuple(ir.shape)))
    return ir.contiguous()

  def build(self, unused_input_shape=None, device=None, seed=0):
    """Initialise the impulse response: N(0, 1e-6) as tf.random_normal_initializer (effects.py:71-80)."""
    if self.trainable and self._ir is None:
      gen = torch.Generator(device='cpu').manual_seed(seed)
      ir = torch.randn(self._reverb_length, generator=gen, dtype=torch.float32) * 1e-6
      self._ir = ir.to(device if device is not None else core._device())
    self.built = True

  def get_controls(self, audio, ir=None):
    """Dry audio [batch, n_samples] and the impulse response (effects.py:82-98).

    Raises:
      ValueError: if trainable=False and ir is not provided.
    """
    if self.trainable:
      if not self.built:
        self.build(device=tf_float32(audio).device)
      ir = self._ir                 # one IR; _match_dimensions' tile happens inside the kernel
    elif ir is None:
      raise ValueError('Must provide "ir" tensor if Reverb trainable=False.')
    return {'audio': audio, 'ir': ir}

  def get_signal(self, audio, ir):
    """Apply the impulse response -> [batch, n_samples] (effects.py:100-117)."""
    audio = tf_float32(audio)
    ir = tf_float32(ir)
    if torch.is_grad_enabled() and (audio.requires_grad or ir.requires_grad):
      return _ReverbFunction.apply(audio, ir, self)
    return self._forward(audio, self._ir_2d(ir))

  def _forward(self, audio, ir2d):
    # _mask_dry_ir (tap 0 -> 0), fft_convolve(padding='same', delay_compensation=0) and the
    # optional dry sum are one C-ABI call
    return core.fft_convolve_long(audio, ir2d, delay=0, add_dry=self._add_dry, mask_tap0=True,
                                  workspace=self._ws)

  def _backward(self, audio, ir2d, grad_out, need_audio, need_ir):
    """The two correlations of the backward pass, as FFT convolutions with reversed indices."""
    g = tf_float32(grad_out)
    n, l = audio.shape[1], ir2d.shape[1]
    grad_audio = grad_ir = None
    if need_audio:       # dL/d audio = reverse(conv(reverse(g), masked ir)[0:N]) (+ g)
      grad_audio = core.fft_convolve_long(g, ir2d, delay=0, add_dry=self._add_dry, mask_tap0=True,
                                          workspace=self._ws, reverse_audio=True, reverse_out=True)
    if need_ir:          # dL/d ir[k] = conv(g, reverse(audio))[N-1+k]; the masked tap gets none
      # (the masked dry tap gets no gradient: logical output 0 written as zero)
      grad_ir = core.fft_convolve_long(g, audio, delay=n - 1, n_out=l, reverse_ir=True, zero_out0=True,
                                       workspace=self._ws_bwd)
      if ir2d.shape[0] == 1 and audio.shape[0] > 1:
        # one IR for the whole batch collects the rows' gradients, in a fixed order (ddsp_sum_rows_f32)
        summed = torch.empty((1, l), dtype=torch.float32, device=grad_ir.device)
        _lib.check(_lib.load().ddsp_sum_rows_f32(grad_ir.data_ptr(), summed.data_ptr(), grad_ir.shape[0], l, 0, core._stream()),
                   'ddsp_sum_rows_f32')
        grad_ir = summed
    return grad_audio, grad_ir


class _ReverbFunction(torch.autograd.Function):
  """torch.autograd node of Reverb.get_signal (plumbing: both directions are C-ABI calls)."""

  @staticmethod
  def forward(ctx, audio, ir, rev):
    ir2d = rev._ir_2d(ir.detach())
    ctx.save_for_backward(audio, ir2d)
    ctx.rev, ctx.ir_shape = rev, ir.shape
    return rev._forward(audio.detach(), ir2d)

  @staticmethod
  def backward(ctx, grad_out):
    audio, ir2d = ctx.saved_tensors
    ga, gi = ctx.rev._backward(audio.detach(), ir2d, grad_out, ctx.needs_input_grad[0],
                               ctx.needs_input_grad[1])
    if gi is not None:
      gi = gi.reshape(ctx.ir_shape)
    return ga, gi, None


class ExpDecayReverb(Reverb):
  """Parameterize impulse response as a simple exponential decay (ddsp/effects.py:120-199).

  `seed` is this package's extension: the reference draws a fresh tf.random.uniform([1, reverb_length])
  burst from TensorFlow's global generator on every call; here the burst is Philox4x32-10 keyed by
  (seed, call counter).  `get_controls(..., noise=...)` is the parity entry with a supplied burst.
  """

  _variable_names = ('_gain', '_decay')

  def __init__(self, trainable=False, reverb_length=48000, scale_fn=core.exp_sigmoid, add_dry=True,
               name='exp_decay_reverb', seed=0):
    super().__init__(name=name, add_dry=add_dry, trainable=trainable)
    self._reverb_length = reverb_length
    self._scale_fn = scale_fn
    self._gain = None
    self._decay = None
    self.seed = int(seed)
    self._calls = 0

  def build(self, unused_input_shape=None, device=None, seed=0):
    """gain = 2.0, decay = 4.0, one value each, when trainable (effects.py:153-168)."""
    del seed
    if self.trainable and self._gain is None:
      device = device if device is not None else core._device()
      self._gain = torch.full((1,), 2.0, dtype=torch.float32, device=device)
      self._decay = torch.full((1,), 4.0, dtype=torch.float32, device=device)
    self.built = True

  def _get_ir(self, gain, decay, noise=None):
    """Simple exponential decay of white noise (effects.py:144-151): gain, decay [batch, 1] -> [batch, L]."""
    gain, decay = tf_float32(gain), tf_float32(decay)
    fused = self._scale_fn is core.exp_sigmoid
    if not fused and self._scale_fn is not None:
      gain = tf_float32(self._scale_fn(gain))       # any other callable runs as given, on device tensors
    if gain.numel() != decay.numel() or gain.dim() > 2 or decay.dim() > 2:
      raise ValueError('gain and decay must both be [batch, 1], got {} and {}'.format(
          tuple(gain.shape), tuple(decay.shape)))
    if noise is None:
      seed = (self.seed & 0xFFFFFFFF) | ((self._calls & 0xFFFFFFFF) << 32)
      self._calls += 1
      noise = core.uniform_noise(1, self._reverb_length, seed=seed)
    else:
      noise = tf_float32(noise).reshape(1, -1)
      if noise.shape[1] != self._reverb_length:
        raise ValueError('noise must hold reverb_length = {} samples, got {}'.format(
            self._reverb_length, noise.shape[1]))
    if torch.is_grad_enabled() and (gain.requires_grad or decay.requires_grad):
      return _ExpDecayIrFunction.apply(gain, decay, noise.contiguous(), fused)
    return _exp_decay_ir(gain, decay, noise.contiguous(), fused)

  def get_controls(self, audio, gain=None, decay=None, noise=None):
    """Convert network outputs into the impulse response (effects.py:170-199).

    Raises:
      ValueError: if trainable=False and gain and decay are not provided.
    """
    if self.trainable:
      if not self.built:
        self.build(device=tf_float32(audio).device)
      gain, decay = self._gain[None, :], self._decay[None, :]
    elif gain is None or decay is None:
      raise ValueError('Must provide "gain" and "decay" tensors if '
                       'ExpDecayReverb trainable=False.')
    ir = self._get_ir(gain, decay, noise)   # trainable: [1, L]; _match_dimensions' tile happens in the kernel
    return {'audio': audio, 'ir': ir}


def _exp_decay_ir(gain, decay, noise, fused_scale):
  b, l = gain.numel(), noise.shape[1]
  gain, decay = gain.reshape(b).contiguous(), decay.reshape(b).contiguous()
  ir = torch.empty((b, l), dtype=torch.float32, device=gain.device)
  rc = _lib.load().ddsp_exp_decay_ir_f32(gain.data_ptr(), decay.data_ptr(), noise.data_ptr(), ir.data_ptr(), b, l,
                                         _lib.DECAY_SCALE_EXP_SIGMOID if fused_scale else 0, core._stream())
  _lib.check(rc, 'ddsp_exp_decay_ir_f32')
  return ir


class _ExpDecayIrFunction(torch.autograd.Function):
  """torch.autograd node of ExpDecayReverb._get_ir (plumbing: both directions are C-ABI calls)."""

  @staticmethod
  def forward(ctx, gain, decay, noise, fused_scale):
    ctx.save_for_backward(gain, decay, noise)
    ctx.fused_scale = fused_scale
    return _exp_decay_ir(gain.detach(), decay.detach(), noise, fused_scale)

  @staticmethod
  def backward(ctx, grad_ir):
    gain, decay, noise = ctx.saved_tensors
    b, l = gain.numel(), noise.shape[1]
    lib = _lib.load()
    g = tf_float32(grad_ir)
    gain_flat, decay_flat = gain.detach().reshape(b).contiguous(), decay.detach().reshape(b).contiguous()
    grad_gain = torch.empty(b, dtype=torch.float32, device=g.device)
    grad_decay = torch.empty(b, dtype=torch.float32, device=g.device)
    ws = core.Workspace().get(core.cached_workspace_bytes('ddsp_exp_decay_ir_backward_workspace_bytes', b, l),
                              g.device)
    rc = lib.ddsp_exp_decay_ir_backward_f32(
        gain_flat.data_ptr(), decay_flat.data_ptr(), noise.data_ptr(), g.data_ptr(), grad_gain.data_ptr(),
        grad_decay.data_ptr(), ws.data_ptr(), ws.numel(), b, l,
        _lib.DECAY_SCALE_EXP_SIGMOID if ctx.fused_scale else 0, core._stream())
    _lib.check(rc, 'ddsp_exp_decay_ir_backward_f32')
    return grad_gain.reshape(gain.shape), grad_decay.reshape(decay.shape), None, None


class FilteredNoiseReverb(Reverb):
  """Impulse response = the output of a filtered-noise synth (ddsp/effects.py:200-278).

  `seed` is this package's extension (FilteredNoise draws Philox noise keyed by it).
  """

  _variable_names = ('_magnitudes',)

  def __init__(self, trainable=False, reverb_length=48000, window_size=257, n_frames=1000,
               n_filter_banks=16, scale_fn=core.exp_sigmoid, initial_bias=-3.0, add_dry=True,
               name='filtered_noise_reverb', seed=0):
    from ddsp_amd import synths          # effects <- synths, as in the reference
    super().__init__(name=name, add_dry=add_dry, trainable=trainable)
    self._n_frames = n_frames
    self._n_filter_banks = n_filter_banks
    self._magnitudes = None
    self._synth = synths.FilteredNoise(n_samples=reverb_length, window_size=window_size,
                                       scale_fn=scale_fn, initial_bias=initial_bias, seed=seed)

  def build(self, unused_input_shape=None, device=None, seed=0):
    """N(0, 1e-2) magnitudes [n_frames, n_filter_banks] when trainable (effects.py:238-247)."""
    if self.trainable and self._magnitudes is None:
      gen = torch.Generator(device='cpu').manual_seed(seed)
      m = torch.randn(self._n_frames, self._n_filter_banks, generator=gen, dtype=torch.float32) * 1e-2
      self._magnitudes = m.to(device if device is not None else core._device())
    self.built = True

  def get_controls(self, audio, magnitudes=None):
    """Dry audio and the impulse response synthesised from `magnitudes` (effects.py:249-278)."""
    if self.trainable:
      if not self.built:
        self.build(device=tf_float32(audio).device)
      magnitudes = self._magnitudes[None, :]
    elif magnitudes is None:
      raise ValueError('Must provide "magnitudes" tensor if '
                       'FilteredNoiseReverb trainable=False.')
    ir = self._synth(magnitudes)          # [batch or 1, reverb_length]; a batch-1 IR is tiled in the kernel
    return {'audio': audio, 'ir': ir}


class FIRFilter(processors.Processor):
  """Linear time-varying finite impulse response (LTV-FIR) filter (ddsp/effects.py:283-322)."""

  def __init__(self, window_size=257, scale_fn=core.exp_sigmoid, name='fir_filter'):
    super().__init__(name=name)
    self.window_size = window_size
    self.scale_fn = scale_fn

  def get_controls(self, audio, magnitudes):
    """Scaled magnitudes [batch, time, n_filter_banks] next to the dry audio."""
    if self.scale_fn is not None:
      magnitudes = self.scale_fn(magnitudes)
    return {'audio': audio, 'magnitudes': magnitudes}

  def get_signal(self, audio, magnitudes):
    """Filter audio [batch, n_samples] with the time-varying FIR designed from `magnitudes`."""
    return core.frequency_filter(audio, magnitudes, window_size=self.window_size)
