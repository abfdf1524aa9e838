"""Harmonic and FilteredNoise synthesisers (mirror of ddsp/synths.py:55-196).

Constructor kwargs, defaults, `get_controls` / `get_signal` / `__call__` signatures and
the controls-dict keys are the reference's.  Arithmetic: HIP kernels only.
"""
import torch

from ddsp_amd import _lib
from ddsp_amd import core
from ddsp_amd import processors


class TensorToAudio(processors.Processor):
  """Identity "synth" returning input samples with channel dimension removed (synths.py:23-52)."""

  def __init__(self, name='tensor_to_audio'):
    super().__init__(name=name)

  def get_controls(self, samples):
    return {'samples': samples}

  def get_signal(self, samples):
    samples = core.tf_float32(samples)
    if samples.dim() != 3 or samples.shape[2] != 1:
      raise ValueError('samples must be [batch, time, 1], got {}'.format(tuple(samples.shape)))
    return samples[:, :, 0]


class Harmonic(processors.Processor):
  """Synthesize audio with a bank of harmonic sinusoidal oscillators (synths.py:55-146).

  `kernel` (an attribute, not a constructor argument - the constructor is the reference's) selects how
  __call__ sums the harmonics where both kernels apply: 'auto' tabulates each frame's waveform on the
  matrix cores and interpolates it per sample (harm_table_kernel), 'direct' evaluates every harmonic at
  every sample (harm_fused_kernel).  Same result within the parity tolerance.
  """
  kernel = 'auto'

  def __init__(self,
               n_samples=64000,
               sample_rate=16000,
               scale_fn=core.exp_sigmoid,
               normalize_below_nyquist=True,
               amp_resample_method='window',
               use_angular_cumsum=False,
               name='harmonic'):
    super().__init__(name=name)
    self.n_samples = n_samples
    self.sample_rate = sample_rate
    self.scale_fn = scale_fn
    self.normalize_below_nyquist = normalize_below_nyquist
    self.amp_resample_method = amp_resample_method
    self.use_angular_cumsum = use_angular_cumsum
    self._ws = core.Workspace()
    self._ws_bwd = core.Workspace()

  # -- helpers -------------------------------------------------------------------------
  def _prescale(self, amplitudes, harmonic_distribution):
    """Returns (amplitudes, harmonic_distribution, fuse_exp_sigmoid)."""
    amplitudes = core.tf_float32(amplitudes)
    harmonic_distribution = core.tf_float32(harmonic_distribution)
    if self.scale_fn is None:
      return amplitudes, harmonic_distribution, False
    if self.scale_fn is core.exp_sigmoid:
      return amplitudes, harmonic_distribution, True          # fused into the kernel
    # any other user callable runs as given, on device tensors
    return (core.tf_float32(self.scale_fn(amplitudes)),
            core.tf_float32(self.scale_fn(harmonic_distribution)), False)

  def get_controls(self, amplitudes, harmonic_distribution, f0_hz):
    """Network outputs -> {'amplitudes', 'harmonic_distribution', 'f0_hz'} (synths.py:94-121)."""
    amplitudes, harmonic_distribution, fuse = self._prescale(amplitudes, harmonic_distribution)
    f0_hz = core.tf_float32(f0_hz)
    amplitudes, harmonic_distribution, f0_hz = core._broadcast_batch(amplitudes, harmonic_distribution, f0_hz)
    b, f, k = core._check_harmonic_shapes(amplitudes, harmonic_distribution, f0_hz)
    core.require_no_grad('Harmonic.get_controls (use __call__, which is differentiable)', amplitudes,
                         harmonic_distribution, f0_hz)
    ctl_amp = torch.empty_like(amplitudes)
    ctl_hd = torch.empty_like(harmonic_distribution)
    flags = core._harmonic_flags(fuse, self.normalize_below_nyquist, 'window', False)
    # normalisation (safe_divide by the sum) always runs; only the Nyquist mask is optional
    rc = _lib.load().ddsp_harmonic_controls_f32(
        amplitudes.data_ptr(), harmonic_distribution.data_ptr(), f0_hz.data_ptr(),
        ctl_amp.data_ptr(), ctl_hd.data_ptr(), b, f, k, int(self.sample_rate), flags,
        core._stream())
    _lib.check(rc, 'ddsp_harmonic_controls_f32')
    return {'amplitudes': ctl_amp, 'harmonic_distribution': ctl_hd, 'f0_hz': f0_hz}

  def get_signal(self, amplitudes, harmonic_distribution, f0_hz):
    """Controls -> audio [batch, n_samples] (synths.py:123-146)."""
    return core.harmonic_synthesis(
        frequencies=f0_hz, amplitudes=amplitudes, harmonic_distribution=harmonic_distribution,
        n_samples=self.n_samples, sample_rate=self.sample_rate,
        amp_resample_method=self.amp_resample_method,
        use_angular_cumsum=self.use_angular_cumsum, workspace=self._ws)

  def call(self, amplitudes, harmonic_distribution, f0_hz, return_outputs_dict=False, **kwargs):
    """get_signal(**get_controls(...)) (processors.py:53-68) as ONE fused C-ABI call.

    When an input tensor requires grad the call is recorded for torch.autograd: backward()
    produces dL/d amplitudes and dL/d harmonic_distribution (ddsp_harmonic_backward_f32) and, when
    f0_hz itself requires grad (the reference's models take f0 from CREPE, a constant; a learned f0
    decoder does not), dL/d f0_hz (ddsp_harmonic_f0_grad_f32).
    """
    for k in ['training', 'mask']:
      kwargs.pop(k, None)
    if kwargs:
      raise TypeError('unexpected keyword arguments: {}'.format(sorted(kwargs)))
    raw_amplitudes, raw_harmonic_distribution = amplitudes, harmonic_distribution
    amplitudes, harmonic_distribution, fuse = self._prescale(amplitudes, harmonic_distribution)
    f0_hz = core.tf_float32(f0_hz)
    amplitudes, harmonic_distribution, f0_hz = core._broadcast_batch(amplitudes, harmonic_distribution, f0_hz)
    b, f, k = core._check_harmonic_shapes(amplitudes, harmonic_distribution, f0_hz)
    core._check_amp_method(self.amp_resample_method, f, int(self.n_samples))
    needs_grad = torch.is_grad_enabled() and (amplitudes.requires_grad or harmonic_distribution.requires_grad or
                                              f0_hz.requires_grad)
    if k > 2048:
      # (harm_controls_kernel holds a row as ceil(K / 64) values per lane, 32 at most; the reference has no cap - 2048 harmonics
      #  are all below Nyquist only under f0 = 11.7 Hz at 48 kHz)
      raise NotImplementedError('the MI355X path holds at most 2048 harmonics per frame, got {}'.format(k))
    closed_form = core._on_closed_form_kernels(self.amp_resample_method, f, int(self.n_samples))
    if needs_grad and closed_form and (k > 256 or int(self.n_samples) // f > 2048):
      # the closed-form BACKWARD kernels take up to 256 harmonics and frames of up to 2048 samples (csrc/harmonic.hip); beyond
      # (300 harmonics at 48 kHz: found raising by a probe after tools/fuzz_parity.py) the chain of materialised envelopes
      # and its adjoint, which take any shape the forward takes
      closed_form = False
    if not closed_form:
      # 'nearest' / 'cubic' envelopes, or n_samples not a multiple of n_frames: the reference's own two
      # steps, get_signal following its chain of materialised envelopes (core.harmonic_synthesis)
      if needs_grad:
        if self.scale_fn is not None and self.scale_fn is not core.exp_sigmoid:
          raise NotImplementedError('the backward pass of Harmonic covers scale_fn=core.exp_sigmoid and scale_fn=None')
        # the same two steps as one torch.autograd node whose backward pass is the chain's adjoint, op for op
        # (round 5: oscillator_bank's gradient w.r.t. its amplitude envelopes, the adjoint of core.resample, get_controls)
        audio, ctl_amp, ctl_hd = _HarmonicMaterialisedFunction.apply(amplitudes, harmonic_distribution, f0_hz, self, fuse)
        if not return_outputs_dict:
          return audio
        return dict(signal=audio, controls={'amplitudes': ctl_amp, 'harmonic_distribution': ctl_hd, 'f0_hz': f0_hz})
      controls = self.get_controls(raw_amplitudes, raw_harmonic_distribution, f0_hz)
      signal = self.get_signal(**controls)
      return dict(signal=signal, controls=controls) if return_outputs_dict else signal
    if needs_grad:
      # one launch gives the audio and, if asked for, the controls dict (not differentiable here)
      audio, ctl_amp, ctl_hd = _HarmonicFunction.apply(amplitudes, harmonic_distribution, f0_hz, self, fuse,
                                                       bool(return_outputs_dict))
      if not return_outputs_dict:
        return audio
      return dict(signal=audio, controls={'amplitudes': ctl_amp, 'harmonic_distribution': ctl_hd, 'f0_hz': f0_hz})
    return self._forward(amplitudes, harmonic_distribution, f0_hz, fuse, return_outputs_dict)

  def call_add(self, amplitudes, harmonic_distribution, f0_hz, add_signal):
    """`Add()(add_signal, self(amplitudes, harmonic_distribution, f0_hz))` (processors.py:162-176) as ONE launch where
    the wavetable kernel applies (ddsp_harmonic_add_f32: one [batch, n_samples] stream written instead of three more
    moved), the two calls otherwise.  Bit-identical to the two calls.  Differentiable (round 4): with a tensor that requires grad the
    same launch is recorded as one torch.autograd node (_HarmonicAddFunction).  ProcessorGroup uses it when nothing asks for
    the intermediate signals."""
    add_signal = core.tf_float32(add_signal)
    amps, hd, fuse = self._prescale(amplitudes, harmonic_distribution)
    f0 = core.tf_float32(f0_hz)
    amps, hd, f0 = core._broadcast_batch(amps, hd, f0)
    wants_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (amps, hd, f0, add_signal))
    b, f, k = core._check_harmonic_shapes(amps, hd, f0)
    n = int(self.n_samples)
    if (tuple(add_signal.shape) == (b, n) and self.kernel == 'auto' and
        core._on_closed_form_kernels(self.amp_resample_method, f, n)):
      if wants_grad:
        # the same launch as a torch.autograd node: the Add's gradient is the identity on both operands, the Harmonic's is
        # Harmonic's (round 4: training DAGs keep the fused launch instead of falling back to two calls and an add kernel)
        try:
          return _HarmonicAddFunction.apply(amps, hd, f0, add_signal, self, fuse)
        except _FusedAddUnsupported:
          pass
      else:
        audio = self._forward_add(amps, hd, f0, add_signal, fuse)
        if audio is not None:
          return audio
    if wants_grad:
      # (a shape or flag set the fused kernel does not take, with a gradient wanted: the two differentiable calls)
      return processors.Add()(add_signal, self.call(amplitudes, harmonic_distribution, f0_hz))
    return processors._add(add_signal, self.call(amplitudes, harmonic_distribution, f0_hz))

  def _forward_add(self, amps, hd, f0, add_signal, fuse):
    """ddsp_harmonic_add_f32, or None where the wavetable kernel does not take the shape / flags."""
    b, f, k = hd.shape
    n = int(self.n_samples)
    audio = torch.empty((b, n), dtype=torch.float32, device=amps.device)
    rc = _lib.load().ddsp_harmonic_add_f32(amps.data_ptr(), hd.data_ptr(), f0.data_ptr(), add_signal.data_ptr(),
                                           audio.data_ptr(), b, f, k, n, int(self.sample_rate), self._flags(fuse),
                                           core._stream())
    if rc == -3:                     # DDSP_ERR_UNSUPPORTED: this shape / these flags run on the other kernels
      return None
    _lib.check(rc, 'ddsp_harmonic_add_f32')
    return audio

  def _flags(self, fuse):
    flags = core._harmonic_flags(fuse, self.normalize_below_nyquist, self.amp_resample_method,
                                 self.use_angular_cumsum)
    if self.kernel == 'direct':
      flags |= _lib.HARM_DIRECT_SUM
    elif self.kernel != 'auto':
      raise ValueError("Harmonic.kernel must be 'auto' or 'direct', got {!r}".format(self.kernel))
    return flags

  def _forward(self, amplitudes, harmonic_distribution, f0_hz, fuse, return_outputs_dict=False):
    b, f, k = harmonic_distribution.shape
    n = int(self.n_samples)
    lib = _lib.load()
    dev = amplitudes.device
    audio = torch.empty((b, n), dtype=torch.float32, device=dev)
    ctl_amp = torch.empty_like(amplitudes) if return_outputs_dict else None
    ctl_hd = torch.empty_like(harmonic_distribution) if return_outputs_dict else None
    ws = self._ws.get(core.cached_workspace_bytes('ddsp_harmonic_workspace_bytes', b, f, k, n), dev)
    rc = lib.ddsp_harmonic_f32(
        amplitudes.data_ptr(), harmonic_distribution.data_ptr(), f0_hz.data_ptr(),
        audio.data_ptr(), ctl_amp.data_ptr() if return_outputs_dict else None,
        ctl_hd.data_ptr() if return_outputs_dict else None, ws.data_ptr(), ws.numel(), b, f, k,
        n, int(self.sample_rate), self._flags(fuse), core._stream())
    _lib.check(rc, 'ddsp_harmonic_f32')
    if return_outputs_dict:
      controls = {'amplitudes': ctl_amp, 'harmonic_distribution': ctl_hd, 'f0_hz': f0_hz}
      return dict(signal=audio, controls=controls)
    return audio

  def _backward(self, amplitudes, harmonic_distribution, f0_hz, fuse, grad_audio):
    b, f, k = harmonic_distribution.shape
    n = int(self.n_samples)
    lib = _lib.load()
    dev = amplitudes.device
    grad_audio = core.tf_float32(grad_audio)
    grad_amp = torch.empty_like(amplitudes)
    grad_hd = torch.empty_like(harmonic_distribution)
    ws = self._ws_bwd.get(core.cached_workspace_bytes('ddsp_harmonic_backward_workspace_bytes',
                                                      b, f, k, n), dev)
    rc = lib.ddsp_harmonic_backward_f32(
        amplitudes.data_ptr(), harmonic_distribution.data_ptr(), f0_hz.data_ptr(),
        grad_audio.data_ptr(), grad_amp.data_ptr(), grad_hd.data_ptr(), ws.data_ptr(), ws.numel(),
        b, f, k, n, int(self.sample_rate), self._flags(fuse), 0, core._stream())
    _lib.check(rc, 'ddsp_harmonic_backward_f32')
    return grad_amp, grad_hd


  def _controls(self, amplitudes, harmonic_distribution, f0_hz, fuse):
    """get_controls on tensors that went through _prescale already (one launch)."""
    b, f, k = harmonic_distribution.shape
    ctl_amp = torch.empty_like(amplitudes)
    ctl_hd = torch.empty_like(harmonic_distribution)
    rc = _lib.load().ddsp_harmonic_controls_f32(
        amplitudes.data_ptr(), harmonic_distribution.data_ptr(), f0_hz.data_ptr(), ctl_amp.data_ptr(),
        ctl_hd.data_ptr(), b, f, k, int(self.sample_rate),
        core._harmonic_flags(fuse, self.normalize_below_nyquist, 'window', False), core._stream())
    _lib.check(rc, 'ddsp_harmonic_controls_f32')
    return {'amplitudes': ctl_amp, 'harmonic_distribution': ctl_hd, 'f0_hz': f0_hz}

  def _backward_materialised(self, amplitudes, harmonic_distribution, f0_hz, fuse, grad_audio, want_controls=True,
                             want_f0=False):
    """(dL/d amplitudes, dL/d harmonic_distribution, dL/d f0_hz) through the chain of materialised envelopes
    (core.py:1080-1111), the chain's adjoint op for op; the ones not asked for are None."""
    b, f, k = harmonic_distribution.shape
    n = int(self.n_samples)
    lib = _lib.load()
    dev = amplitudes.device
    # This chain holds up to four [batch, n_samples, n_harmonics] fp32 envelopes at once (the reference materialises the same
    # tensors): 300 harmonics x 10 s at 48 kHz x batch 32 is 18 GB each.  Refuse with the arithmetic spelled out instead of
    # dying in the allocator half way through a backward pass.
    envelope_bytes = 4 * b * n * k
    free_bytes = torch.cuda.mem_get_info(dev)[0] if dev.type == 'cuda' else None
    if free_bytes is not None and 4 * envelope_bytes > free_bytes:
      raise MemoryError('Harmonic backward on the materialised-envelope chain (amp_resample_method={!r}, {} harmonics, frames of {} '
                        'samples) needs four [batch={}, n_samples={}, n_harmonics={}] fp32 envelopes = {:.1f} GB; {:.1f} GB are free on '
                        '{}: split the batch'.format(self.amp_resample_method, k, n // max(f, 1), b, n, k, 4 * envelope_bytes / 1e9,
                                                     free_bytes / 1e9, dev))
    grad_audio = core.tf_float32(grad_audio)
    # the envelopes the forward pass ran on: f0 [1 .. K] resampled 'linear', amplitudes * distribution resampled by the method
    harmonic_frequencies = torch.empty((b, f, k), dtype=torch.float32, device=dev)
    harmonic_amplitudes = torch.empty((b, f, k), dtype=torch.float32, device=dev)
    ctl = self._controls(amplitudes, harmonic_distribution, f0_hz, fuse)
    rc = lib.ddsp_harmonic_envelopes_f32(ctl['amplitudes'].data_ptr(), ctl['harmonic_distribution'].data_ptr(),
                                         f0_hz.data_ptr(), None, harmonic_frequencies.data_ptr(),
                                         harmonic_amplitudes.data_ptr(), b, f, k, core._stream())
    _lib.check(rc, 'ddsp_harmonic_envelopes_f32')
    frequency_envelopes = core.resample(harmonic_frequencies, n)
    half = lib.ddsp_oscillator_bank_workspace_bytes(b, n, k)
    ws = self._ws_bwd.get(2 * half, dev)
    grad_amp = grad_hd = grad_f0 = None
    if want_controls:
      grad_env = torch.empty((b, n, k), dtype=torch.float32, device=dev)
      rc = lib.ddsp_oscillator_bank_grad_amplitudes_f32(frequency_envelopes.data_ptr(), grad_audio.data_ptr(),
                                                        grad_env.data_ptr(), ws.data_ptr(), ws.numel(), b, n, k,
                                                        int(self.sample_rate), core._stream())
      _lib.check(rc, 'ddsp_oscillator_bank_grad_amplitudes_f32')
      grad_ha = torch.empty((b, f, k), dtype=torch.float32, device=dev)
      rc = lib.ddsp_resample_ex_backward_f32(grad_env.data_ptr(), grad_ha.data_ptr(), b, f, n, k,
                                             _lib.RESAMPLE_METHODS[self.amp_resample_method], 1, core._stream())
      _lib.check(rc, 'ddsp_resample_ex_backward_f32')
      del grad_env
      grad_amp = torch.empty_like(amplitudes)
      grad_hd = torch.empty_like(harmonic_distribution)
      rc = lib.ddsp_harmonic_controls_backward_f32(
          amplitudes.data_ptr(), harmonic_distribution.data_ptr(), f0_hz.data_ptr(), grad_ha.data_ptr(),
          grad_amp.data_ptr(), grad_hd.data_ptr(), b, f, k, int(self.sample_rate),
          core._harmonic_flags(fuse, self.normalize_below_nyquist, 'window', False), 0, core._stream())
      _lib.check(rc, 'ddsp_harmonic_controls_backward_f32')
    if want_f0:
      # dL/d frequency envelopes (a suffix sum over time of dL/d audio A mask cos(phase)), the adjoint of the 'linear'
      # resample, the sum over harmonics weighted [1 .. K]; the frame-rate mask of get_controls (tf.where) passes none
      amplitude_envelopes = core.resample(harmonic_amplitudes, n, method=self.amp_resample_method)
      grad_fenv = torch.empty((b, n, k), dtype=torch.float32, device=dev)
      rc = lib.ddsp_oscillator_bank_grad_frequencies_f32(
          frequency_envelopes.data_ptr(), amplitude_envelopes.data_ptr(), grad_audio.data_ptr(), grad_fenv.data_ptr(),
          ws.data_ptr(), ws.numel(), b, n, k, int(self.sample_rate), core._stream())
      _lib.check(rc, 'ddsp_oscillator_bank_grad_frequencies_f32')
      del amplitude_envelopes
      grad_hf = torch.empty((b, f, k), dtype=torch.float32, device=dev)
      rc = lib.ddsp_resample_ex_backward_f32(grad_fenv.data_ptr(), grad_hf.data_ptr(), b, f, n, k,
                                             _lib.RESAMPLE_METHODS['linear'], 1, core._stream())
      _lib.check(rc, 'ddsp_resample_ex_backward_f32')
      del grad_fenv
      grad_f0 = torch.empty_like(f0_hz)
      rc = lib.ddsp_harmonic_frequencies_backward_f32(grad_hf.data_ptr(), None, grad_f0.data_ptr(), b, f, k, core._stream())
      _lib.check(rc, 'ddsp_harmonic_frequencies_backward_f32')
    return grad_amp, grad_hd, grad_f0

  def _backward_f0(self, amplitudes, harmonic_distribution, f0_hz, fuse, grad_audio):
    """dL/d f0_hz [B,F,1]: the controls once more (one small launch), then ddsp_harmonic_f0_grad_f32."""
    b, f, k = harmonic_distribution.shape
    n = int(self.n_samples)
    lib = _lib.load()
    dev = amplitudes.device
    grad_audio = core.tf_float32(grad_audio)
    ctl_amp = torch.empty_like(amplitudes)
    ctl_hd = torch.empty_like(harmonic_distribution)
    rc = lib.ddsp_harmonic_controls_f32(
        amplitudes.data_ptr(), harmonic_distribution.data_ptr(), f0_hz.data_ptr(), ctl_amp.data_ptr(),
        ctl_hd.data_ptr(), b, f, k, int(self.sample_rate),
        core._harmonic_flags(fuse, self.normalize_below_nyquist, 'window', False), core._stream())
    _lib.check(rc, 'ddsp_harmonic_controls_f32')
    grad_f0 = torch.empty_like(f0_hz)
    ws = self._ws_bwd.get(core.cached_workspace_bytes('ddsp_harmonic_f0_grad_workspace_bytes', b, f, k, n), dev)
    rc = lib.ddsp_harmonic_f0_grad_f32(
        ctl_amp.data_ptr(), ctl_hd.data_ptr(), f0_hz.data_ptr(), grad_audio.data_ptr(), grad_f0.data_ptr(),
        ws.data_ptr(), ws.numel(), b, f, k, n, int(self.sample_rate),
        _lib.HARM_AMP_LINEAR if self.amp_resample_method == 'linear' else 0, core._stream())
    _lib.check(rc, 'ddsp_harmonic_f0_grad_f32')
    return grad_f0


class _HarmonicMaterialisedFunction(torch.autograd.Function):
  """torch.autograd node of Harmonic.__call__ on the reference's own chain of materialised envelopes ('nearest' / 'cubic'
  amplitude envelopes, n_samples that is not a multiple of n_frames; ddsp/core.py:1080-1111).  Forward: get_controls, then
  core.harmonic_synthesis.  Backward, the chain's adjoint op for op, every op a C-ABI call:
      dL/d amplitude_envelopes [B,N,K] = dL/d audio mask sin(phase)          ddsp_oscillator_bank_grad_amplitudes_f32
      dL/d (amplitudes distribution) [B,F,K] = resample^T of it              ddsp_resample_ex_backward_f32
      dL/d amplitudes, dL/d harmonic_distribution through get_controls      ddsp_harmonic_controls_backward_f32
      dL/d frequency_envelopes [B,N,K] (a suffix sum over time)              ddsp_oscillator_bank_grad_frequencies_f32
      dL/d f0_hz = sum_k k resample^T of it                                  ddsp_harmonic_frequencies_backward_f32"""

  @staticmethod
  def forward(ctx, amplitudes, harmonic_distribution, f0_hz, synth, fuse):
    ctx.save_for_backward(amplitudes, harmonic_distribution, f0_hz)
    ctx.synth, ctx.fuse = synth, fuse
    with torch.no_grad():
      controls = synth._controls(amplitudes.detach(), harmonic_distribution.detach(), f0_hz.detach(), fuse)
      signal = synth.get_signal(**controls)
    ctx.mark_non_differentiable(controls['amplitudes'], controls['harmonic_distribution'])
    return signal, controls['amplitudes'], controls['harmonic_distribution']

  @staticmethod
  def backward(ctx, grad_audio, _grad_ctl_amp, _grad_ctl_hd):
    amplitudes, harmonic_distribution, f0_hz = (t.detach() for t in ctx.saved_tensors)
    grad_amp, grad_hd, grad_f0 = ctx.synth._backward_materialised(
        amplitudes, harmonic_distribution, f0_hz, ctx.fuse, grad_audio,
        ctx.needs_input_grad[0] or ctx.needs_input_grad[1], ctx.needs_input_grad[2])
    return grad_amp, grad_hd, grad_f0, None, None


class _FusedAddUnsupported(Exception):
  """ddsp_harmonic_add_f32 does not take this shape / these flags (Harmonic.call_add falls back to the two calls)."""


class _HarmonicAddFunction(torch.autograd.Function):
  """torch.autograd node of Harmonic.call_add: `add_signal + Harmonic(amplitudes, harmonic_distribution, f0_hz)` in one launch."""

  @staticmethod
  def forward(ctx, amplitudes, harmonic_distribution, f0_hz, add_signal, synth, fuse):
    audio = synth._forward_add(amplitudes.detach(), harmonic_distribution.detach(), f0_hz.detach(), add_signal.detach(), fuse)
    if audio is None:
      raise _FusedAddUnsupported()
    ctx.save_for_backward(amplitudes, harmonic_distribution, f0_hz)
    ctx.synth, ctx.fuse = synth, fuse
    return audio

  @staticmethod
  def backward(ctx, grad_audio):
    amplitudes, harmonic_distribution, f0_hz = (t.detach() for t in ctx.saved_tensors)
    grad_amp = grad_hd = grad_f0 = None
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
      grad_amp, grad_hd = ctx.synth._backward(amplitudes, harmonic_distribution, f0_hz, ctx.fuse, grad_audio)
    if ctx.needs_input_grad[2]:
      grad_f0 = ctx.synth._backward_f0(amplitudes, harmonic_distribution, f0_hz, ctx.fuse, grad_audio)
    return grad_amp, grad_hd, grad_f0, (grad_audio if ctx.needs_input_grad[3] else None), None, None


class _HarmonicFunction(torch.autograd.Function):
  """torch.autograd node of Harmonic.__call__ (plumbing: both directions are C-ABI calls)."""

  @staticmethod
  def forward(ctx, amplitudes, harmonic_distribution, f0_hz, synth, fuse, want_controls):
    ctx.save_for_backward(amplitudes, harmonic_distribution, f0_hz)
    ctx.synth, ctx.fuse = synth, fuse
    out = synth._forward(amplitudes.detach(), harmonic_distribution.detach(), f0_hz.detach(), fuse, want_controls)
    if not want_controls:
      empty = amplitudes.new_empty(0)
      ctx.mark_non_differentiable(empty)
      return out, empty, empty
    ctl_amp, ctl_hd = out['controls']['amplitudes'], out['controls']['harmonic_distribution']
    ctx.mark_non_differentiable(ctl_amp, ctl_hd)
    return out['signal'], ctl_amp, ctl_hd

  @staticmethod
  def backward(ctx, grad_audio, _grad_ctl_amp, _grad_ctl_hd):
    amplitudes, harmonic_distribution, f0_hz = ctx.saved_tensors
    amplitudes, harmonic_distribution, f0_hz = amplitudes.detach(), harmonic_distribution.detach(), f0_hz.detach()
    grad_amp = grad_hd = grad_f0 = None
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
      grad_amp, grad_hd = ctx.synth._backward(amplitudes, harmonic_distribution, f0_hz, ctx.fuse, grad_audio)
    if ctx.needs_input_grad[2]:
      grad_f0 = ctx.synth._backward_f0(amplitudes, harmonic_distribution, f0_hz, ctx.fuse, grad_audio)
    return grad_amp, grad_hd, grad_f0, None, None, None


class FilteredNoise(processors.Processor):
  """Synthesize audio by filtering white noise (synths.py:149-196).

  `seed` is an extension: the reference draws from TensorFlow's stateful global generator;
  here noise is Philox4x32-10 keyed by (seed, call counter), generated inside the FIR
  kernel.  `get_signal(magnitudes, noise=...)` is the parity entry with supplied noise.

  `noise_bits` is the other extension: 23 (the default since round 6) draws the 2^23 levels tf.random.uniform's fp32
  samples have (synths.py:192-193), carried through the matrix-core FIR as fp16 hi / lo pairs; 11 draws every sample
  from 2048 equally spaced levels in (-1, 1) - zero mean, variance 1/3, white, each value exactly an fp16 number, which
  lets the FIR carry its noise operand in one fp16 plane (a third fewer matrix products and half the LDS traffic of
  that operand: bench.py's `fnoise_11_bit_levels` block prices it).  Both are Philox4x32-10 streams documented in
  include/ddsp_amd.h.

  `kernel` (an attribute, like Harmonic.kernel; the constructor is the reference's): 'auto' runs the canonical filter
  (65 bands, full window, frames of 64 c samples) on noise_mfma65_kernel - IR design and the time-varying FIR on the fp16
  matrix cores (hi/lo-split operands, fp32 accumulation); 'vector' keeps the FIR on the vector ALUs
  (noise_fused65_kernel).  Same result within the parity tolerance.
  """
  kernel = 'auto'

  def __init__(self,
               n_samples=64000,
               window_size=257,
               scale_fn=core.exp_sigmoid,
               initial_bias=-5.0,
               name='filtered_noise',
               seed=0,
               noise_bits=23):
    super().__init__(name=name)
    if noise_bits not in (11, 23):
      raise ValueError('noise_bits must be 23 (the 2^23 levels of tf.random.uniform, the default) or 11 (2048 levels), '
                       'got {!r}'.format(noise_bits))
    self.n_samples = n_samples
    self.window_size = window_size
    self.scale_fn = scale_fn
    self.initial_bias = initial_bias
    self.seed = int(seed)
    self.noise_bits = int(noise_bits)
    self._calls = 0
    self._ws = core.Workspace()
    self._ws_bwd = core.Workspace()

  def get_controls(self, magnitudes):
    """Network outputs -> {'magnitudes'} (synths.py:165-179)."""
    magnitudes = core.tf_float32(magnitudes)
    if self.scale_fn is None:
      return {'magnitudes': magnitudes}
    if self.scale_fn is not core.exp_sigmoid:
      return {'magnitudes': core.tf_float32(self.scale_fn(magnitudes + self.initial_bias))}
    if magnitudes.dim() != 3:
      raise ValueError('magnitudes must be [batch, n_frames, n_filter_banks], got {}'.format(
          tuple(magnitudes.shape)))
    b, f, m = magnitudes.shape
    core.require_no_grad('FilteredNoise.get_controls (use __call__, which is differentiable)', magnitudes)
    ctl = torch.empty_like(magnitudes)
    rc = _lib.load().ddsp_filtered_noise_controls_f32(
        magnitudes.data_ptr(), ctl.data_ptr(), b, f, m, float(self.initial_bias),
        _lib.NOISE_SCALE_EXP_SIGMOID, core._stream())
    _lib.check(rc, 'ddsp_filtered_noise_controls_f32')
    return {'magnitudes': ctl}

  def _ir_flag(self):
    bits = _lib.NOISE_BITS_23 if self.noise_bits == 23 else 0
    if self.kernel == 'auto':
      return bits
    if self.kernel != 'vector':
      raise ValueError("FilteredNoise.kernel must be 'auto' or 'vector', got {!r}".format(self.kernel))
    return _lib.NOISE_FIR_VECTOR_ALU | bits

  def _next_seed(self):
    s = (self.seed & 0xFFFFFFFF) | ((self._calls & 0xFFFFFFFF) << 32)
    self._calls += 1
    return s

  def _run(self, magnitudes, noise, fuse_scale, want_controls):
    magnitudes = core.tf_float32(magnitudes)
    if magnitudes.dim() != 3:
      raise ValueError('magnitudes must be [batch, n_frames, n_filter_banks], got {}'.format(
          tuple(magnitudes.shape)))
    b, f, m = magnitudes.shape
    n = int(self.n_samples)
    core._check_frames(n, f)
    lib = _lib.load()
    dev = magnitudes.device
    if noise is not None:
      noise = core.tf_float32(noise)
      if tuple(noise.shape) != (b, n):
        raise ValueError('noise must be [{}, {}], got {}'.format(b, n, tuple(noise.shape)))
    if m == 2:
      # an impulse response of 2 taps: crop_and_compensate_delay (core.py:1338-1379) starts its slice at
      # (ir_size - 1) // 2 - 1 = -1 and the reference returns what python's audio[:, -1:-end] leaves - nothing.
      # Same answer as core.fft_convolve gives here (core._crop_range), not [batch, n_samples] of something else.
      size = lib.ddsp_fir_size(m, int(self.window_size))
      _lib.check(min(size, 0), 'ddsp_fir_size')
      _, _, n_out = core._crop_range(n, f, size, 'same', -1)
      if n_out != n:
        self._next_seed()             # the call counts like any other
        ctl = None
        if want_controls:
          ctl = core.exp_sigmoid(magnitudes + float(self.initial_bias)) if fuse_scale else magnitudes
        return torch.empty((b, n_out), dtype=torch.float32, device=dev), ctl
    audio = torch.empty((b, n), dtype=torch.float32, device=dev)
    ctl = torch.empty_like(magnitudes) if want_controls else None
    ws = self._ws.get(core.cached_workspace_bytes('ddsp_filtered_noise_workspace_bytes', b, f, m, n,
                                                  int(self.window_size)), dev)
    rc = lib.ddsp_filtered_noise_f32(
        magnitudes.data_ptr(), noise.data_ptr() if noise is not None else None,
        audio.data_ptr(), ctl.data_ptr() if want_controls else None, ws.data_ptr(), ws.numel(),
        b, f, m, n, int(self.window_size), float(self.initial_bias),
        (_lib.NOISE_SCALE_EXP_SIGMOID if fuse_scale else 0) | self._ir_flag(), self._next_seed(), 0,
        core._stream())
    _lib.check(rc, 'ddsp_filtered_noise_f32')
    return audio, ctl

  def get_signal(self, magnitudes, noise=None):
    """Controls -> filtered noise [batch, n_samples] (synths.py:181-196).  Differentiable with respect to the
    controls (the same backward kernels as __call__, scale function off)."""
    mt = core.tf_float32(magnitudes)
    if torch.is_grad_enabled() and mt.requires_grad:
      return _FilteredNoiseFunction.apply(mt, self, noise, False)
    audio, _ = self._run(mt, noise, fuse_scale=False, want_controls=False)
    return audio

  def call(self, magnitudes, return_outputs_dict=False, noise=None, **kwargs):
    """get_signal(**get_controls(magnitudes)) as one fused C-ABI call.

    When `magnitudes` requires grad the call is recorded for torch.autograd; backward()
    regenerates the same noise.
    """
    for k in ['training', 'mask']:
      kwargs.pop(k, None)
    if kwargs:
      raise TypeError('unexpected keyword arguments: {}'.format(sorted(kwargs)))
    if self.scale_fn is None or self.scale_fn is core.exp_sigmoid:
      mt = core.tf_float32(magnitudes)
      if torch.is_grad_enabled() and mt.requires_grad:
        audio = _FilteredNoiseFunction.apply(mt, self, noise, self.scale_fn is not None)
        if not return_outputs_dict:
          return audio
        with torch.no_grad():
          ctl = self.get_controls(mt.detach())['magnitudes']
        return dict(signal=audio, controls={'magnitudes': ctl})
      audio, ctl = self._run(magnitudes, noise, fuse_scale=self.scale_fn is not None,
                             want_controls=return_outputs_dict)
      if return_outputs_dict and self.scale_fn is None:
        ctl = core.tf_float32(magnitudes)
    else:
      # a user callable runs as given on device tensors (torch autograd follows it); the synthesis behind it is
      # the differentiable get_signal
      ctl = self.get_controls(magnitudes)['magnitudes']
      audio = self.get_signal(ctl, noise=noise)
    if return_outputs_dict:
      return dict(signal=audio, controls={'magnitudes': ctl})
    return audio

  def _backward(self, magnitudes, noise, seed, grad_audio, fuse_scale):
    b, f, m = magnitudes.shape
    n = int(self.n_samples)
    lib = _lib.load()
    nbytes = core.cached_workspace_bytes('ddsp_filtered_noise_backward_workspace_bytes', b, f, m, n)
    ws = self._ws_bwd.get(nbytes, magnitudes.device)
    grad_audio = core.tf_float32(grad_audio)
    grad_mag = torch.empty_like(magnitudes)
    rc = lib.ddsp_filtered_noise_backward_f32(
        magnitudes.data_ptr(), noise.data_ptr() if noise is not None else None,
        grad_audio.data_ptr(), grad_mag.data_ptr(), ws.data_ptr(), ws.numel(), b, f, m, n,
        int(self.window_size), float(self.initial_bias),
        (_lib.NOISE_SCALE_EXP_SIGMOID if fuse_scale else 0) | (_lib.NOISE_BITS_23 if self.noise_bits == 23 else 0),
        seed, 0, core._stream())
    if rc == -3:
      raise NotImplementedError('FilteredNoise backward needs n_samples / n_frames <= 8192, at most 4097 '
                                'bands and an impulse response of at least 3 taps')
    _lib.check(rc, 'ddsp_filtered_noise_backward_f32')
    return grad_mag


class _FilteredNoiseFunction(torch.autograd.Function):
  """torch.autograd node of FilteredNoise.__call__ (plumbing: both directions are C-ABI calls)."""

  @staticmethod
  def forward(ctx, magnitudes, synth, noise, fuse_scale):
    if noise is not None:
      noise = core.tf_float32(noise)
    ctx.seed = (synth.seed & 0xFFFFFFFF) | ((synth._calls & 0xFFFFFFFF) << 32)   # what _run is about to use
    audio, _ = synth._run(magnitudes.detach(), noise, fuse_scale=fuse_scale, want_controls=False)
    ctx.save_for_backward(magnitudes)
    ctx.synth, ctx.noise, ctx.fuse_scale = synth, noise, fuse_scale
    return audio

  @staticmethod
  def backward(ctx, grad_audio):
    (magnitudes,) = ctx.saved_tensors
    return ctx.synth._backward(magnitudes.detach(), ctx.noise, ctx.seed, grad_audio, ctx.fuse_scale), None, None, None
