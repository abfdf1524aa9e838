"""Effects: the convolutional reverb (mirror of ddsp/effects.py:27-117).

SURVEY.md section 8(f) rank 1.  The reference convolves with one 131 072-point FFT per clip; here
`core.fft_convolve_long` runs a partitioned overlap-save convolution with LDS-resident FFTs
(csrc/reverb.hip).  ExpDecayReverb / FilteredNoiseReverb / FIRFilter / ModDelay are not built.
"""
import torch

from ddsp_amd import core
from ddsp_amd import processors

tf_float32 = core.tf_float32


class Reverb(processors.Processor):
  """Convolutional (FIR) reverb (ddsp/effects.py:27-117)."""

  def __init__(self, trainable=False, reverb_length=48000, add_dry=True, name='reverb'):
    """Takes neural network outputs directly as the impulse response.

    Args:
      trainable: hold the impulse response as a single tensor for the entire dataset
        (the reference's tf.Variable, effects.py:71-80; there is no autograd here yet).
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
    ir = self._ir_2d(tf_float32(ir))
    # _mask_dry_ir (tap 0 -> 0), fft_convolve(padding='same', delay_compensation=0) and the
    # optional dry sum are one C-ABI call
    return core.fft_convolve_long(audio, ir, delay=0, add_dry=self._add_dry, mask_tap0=True,
                                  workspace=self._ws)
