"""A numpy stand-in for the handful of TensorFlow ops `ddsp/core.py` calls.

Purpose: TensorFlow, gin and absl are not installed in the build container, so the
reference (`/root/reference/ddsp`) cannot be imported as is.  `install()` puts
minimal fakes of `tensorflow`, `tensorflow.compat.v2`, `gin` and `absl` into
`sys.modules` and registers `ddsp` as a namespace package that points at the
reference's source directory WITHOUT running its `__init__.py` (which would pull in
crepe / tfp / librosa).  After that `import ddsp.core, ddsp.synths, ddsp.processors`
executes the reference's own, unmodified source files on numpy arrays.

Only `tests/golden/make_golden.py` uses this, and only in the build container
(the GPU box has no /root/reference).  Op semantics follow TF <= 2.11 as listed in
SURVEY.md Appendix A: legacy bilinear / nearest / bicubic resize, inclusive sequential cumsum,
periodic Hann, zero-padded framing, plain overlap-add, fp32 FFTs.
"""

import sys
import types

import numpy as np


class TensorShape(tuple):
  def as_list(self):
    return list(self)


class Tensor(np.ndarray):
  """ndarray whose .shape has .as_list(), as tf.TensorShape does."""

  @property
  def shape(self):
    return TensorShape(np.ndarray.shape.__get__(self))

  def numpy(self):
    return np.asarray(self)


def _t(x):
  return np.asarray(x).view(Tensor)


def _f32(a):
  """Python scalars become fp32 tensors in TF (e.g. tf.math.log(10.0))."""
  return np.float32(a) if isinstance(a, (float, int)) else a


def _wrap(fn):
  def inner(*args, **kwargs):
    kwargs.pop('name', None)
    return _t(fn(*[_f32(a) for a in args], **kwargs))
  return inner


def _np(x):
  return np.asarray(x)


def convert_to_tensor(x, dtype=None):
  return _t(np.asarray(x, dtype=dtype))


def cast(x, dtype):
  return _t(np.asarray(x).astype(dtype))


def cumsum(x, axis=0, exclusive=False, reverse=False):
  assert not exclusive and not reverse
  x = _np(x)
  return _t(np.cumsum(x, axis=axis, dtype=x.dtype))  # sequential along axis


def reduce_sum(x, axis=None, keepdims=False):
  x = _np(x)
  return _t(np.sum(x, axis=axis, keepdims=keepdims, dtype=x.dtype))


def where(cond, x, y):
  x, y = _np(x), _np(y)
  dt = np.result_type(x, y) if (x.ndim and y.ndim) else (x.dtype if x.ndim else y.dtype)
  return _t(np.where(_np(cond), x, y).astype(dt))


def pad(x, paddings, mode='CONSTANT', constant_values=0):
  return _t(np.pad(_np(x), [tuple(int(q) for q in p) for p in paddings],
                   constant_values=constant_values))


def linspace(start, stop, num):
  return _t(np.linspace(start, stop, int(num)).astype(np.float32))


def complex_(real, imag):
  return _t(_np(real).astype(np.float32) + 1j * _np(imag).astype(np.float32)).astype(
      np.complex64).view(Tensor)


def sigmoid(x):
  x = _np(x)
  e = np.exp(-np.abs(x))
  one = x.dtype.type(1)
  return _t(np.where(x >= 0, one / (one + e), e / (one + e)))


# --- tf.signal ----------------------------------------------------------------
def hann_window(n, periodic=True, dtype=np.float32):
  """tf.signal.hann_window (tensorflow/python/ops/signal/window_ops.py, _raised_cosine_window): ones([1]) for a window of one
  sample; else a - b cos(2 pi count / n') with n' = window_length + periodic * even - 1, even = 1 - window_length % 2 - `periodic`
  changes EVEN lengths only, an odd length is the symmetric window either way (round 6; before, this stand-in divided by n)."""
  n = int(n)
  if n == 1:
    return _t(np.ones(1, dtype))
  d = n + (1 if periodic else 0) * (1 - n % 2) - 1
  if np.dtype(dtype) == np.float32:         # TF's op order, every step in `dtype`: constant(2 pi) * count / n, a - b * cos(.)
    count = np.arange(n, dtype=np.float32)
    cos_arg = (np.float32(2.0 * np.pi) * count) / np.float32(d)
    return _t((np.float32(0.5) - np.float32(0.5) * np.cos(cos_arg, dtype=np.float32)).astype(np.float32))
  i = np.arange(n, dtype=np.float64)
  return _t((0.5 - 0.5 * np.cos(2.0 * np.pi * i / d)).astype(dtype))


def overlap_and_add(signal, frame_step):
  s = _np(signal)
  n_frames, length = s.shape[-2], s.shape[-1]
  out = np.zeros(s.shape[:-2] + ((n_frames - 1) * frame_step + length,), s.dtype)
  for f in range(n_frames):
    out[..., f * frame_step:f * frame_step + length] += s[..., f, :]
  return _t(out)


def frame(signal, frame_length, frame_step, pad_end=False, pad_value=0, axis=-1):
  s = _np(signal)
  assert axis in (-1, s.ndim - 1)
  n = s.shape[-1]
  if pad_end:
    n_frames = -(-n // frame_step)
    padded_len = (n_frames - 1) * frame_step + frame_length
    p = np.full(s.shape[:-1] + (max(padded_len, n),), pad_value, s.dtype)
    p[..., :n] = s
    s = p
  else:
    n_frames = 1 + (n - frame_length) // frame_step
  idx = np.arange(n_frames)[:, None] * frame_step + np.arange(frame_length)[None, :]
  return _t(s[..., idx])


def rfft(x, fft_length=None):
  x = _np(x).astype(np.float32)
  n = int(fft_length[0]) if fft_length is not None else x.shape[-1]
  return _t(np.fft.rfft(x, n).astype(np.complex64))


def irfft(x, fft_length=None):
  x = _np(x).astype(np.complex64)
  n = int(fft_length[0]) if fft_length is not None else 2 * (x.shape[-1] - 1)
  return _t(np.fft.irfft(x, n).astype(np.float32))


def stft(signals, frame_length, frame_step, fft_length=None, window_fn=None, pad_end=False):
  """tf.signal.stft: frame (zero pad_end), periodic Hann, rfft of the enclosing power of two."""
  x = _np(signals)
  if fft_length is None:
    fft_length = 1 << int(np.ceil(np.log2(frame_length)))
  frames = _np(frame(x, frame_length, frame_step, pad_end=pad_end))
  win = _np(hann_window(frame_length, periodic=True)).astype(frames.dtype)
  return _t(np.fft.rfft(frames * win, int(fft_length)).astype(np.complex64))


def fftshift(x, axes=None):
  return _t(np.fft.fftshift(_np(x), axes=axes))


# --- tf.compat.v1.image.resize (legacy bilinear) ---------------------------------
class ResizeMethod:
  BILINEAR = 'bilinear'
  NEAREST_NEIGHBOR = 'nearest'
  BICUBIC = 'bicubic'


def _interp_weights(out_size, in_size, align_corners):
  if align_corners and out_size > 1:
    scale = np.float32(in_size - 1) / np.float32(out_size - 1)
  else:
    scale = np.float32(in_size) / np.float32(out_size)
  pos = np.arange(out_size, dtype=np.float32) * scale
  lo = np.floor(pos)
  hi = np.minimum(np.ceil(pos), np.float32(in_size - 1))
  return lo.astype(np.int64), hi.astype(np.int64), (pos - lo).astype(np.float32)


def _resize_scale(in_size, out_size, align_corners):
  """CalculateResizeScale (image_resizer_state.h), fp32."""
  if align_corners and out_size > 1:
    return np.float32(in_size - 1) / np.float32(out_size - 1)
  return np.float32(in_size) / np.float32(out_size)


def _nearest_rows(x, out_size, align_corners):
  """ResizeNearestNeighbor, legacy scaler (half_pixel_centers=False), along axis 1:
  in = min(align_corners ? roundf(out*scale) : floorf(out*scale), in_size - 1)."""
  in_size = x.shape[1]
  scale = _resize_scale(in_size, out_size, align_corners)
  src = np.empty(out_size, np.int64)
  for o in range(out_size):
    pos = np.float32(o) * scale
    r = np.floor(pos.astype(np.float64) + 0.5) if align_corners else np.floor(pos)   # roundf: half away from zero, pos >= 0 (the sum in fp64: exact)
    src[o] = min(int(r), in_size - 1)
  return x[:, src]


_BICUBIC_TABLE = None


def _bicubic_table():
  """InitCoeffsTable(A = -0.75) of resize_bicubic_op.cc: 1025 pairs, float storage, double arithmetic."""
  global _BICUBIC_TABLE
  if _BICUBIC_TABLE is None:
    a = -0.75
    tab = np.zeros((1024 + 1) * 2, np.float32)
    for i in range(1024 + 1):
      x = float(np.float32(i * 1.0 / 1024))
      tab[i * 2] = ((a + 2) * x - (a + 3)) * x * x + 1
      x = float(np.float32(x + 1.0))
      tab[i * 2 + 1] = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    _BICUBIC_TABLE = tab
  return _BICUBIC_TABLE


def _bicubic_rows(x, out_size, align_corners):
  """ResizeBicubic, legacy scaler, along axis 1 (GetWeightsAndIndices<LegacyScaler, false> +
  Interpolate1D: v0 w0 + v1 w1 + v2 w2 + v3 w3 in float, indices clamped)."""
  tab = _bicubic_table()
  in_size = x.shape[1]
  scale = _resize_scale(in_size, out_size, align_corners)
  out = np.empty((x.shape[0], out_size) + x.shape[2:], np.float32)
  bound = lambda i: min(max(i, 0), in_size - 1)
  for o in range(out_size):
    in_loc_f = np.float32(o) * scale
    in_loc = int(np.floor(in_loc_f))
    delta = np.float32(in_loc_f - np.float32(in_loc))
    offset = int(np.rint(delta * np.float32(1024)))
    w = (tab[offset * 2 + 1], tab[offset * 2], tab[(1024 - offset) * 2], tab[(1024 - offset) * 2 + 1])
    idx = (bound(in_loc - 1), bound(in_loc), bound(in_loc + 1), bound(in_loc + 2))
    acc = x[:, idx[0]] * w[0]
    for q in (1, 2, 3):
      acc = acc + x[:, idx[q]] * w[q]
    out[:, o] = acc
  return out


def image_resize_v1(images, size, method=ResizeMethod.BILINEAR, align_corners=False):
  """[B, H, W, C] legacy resize (tf.compat.v1.image.resize: no half-pixel centres).  The reference only
  ever resizes the H axis (W stays as it is, core.py:613-621); nearest / bicubic assert that."""
  x = _np(images).astype(np.float32)
  out_h, out_w = int(size[0]), int(size[1])
  if method in (ResizeMethod.NEAREST_NEIGHBOR, ResizeMethod.BICUBIC):
    # along W the scale is 1: nearest picks the same column, bicubic has weights (0, 1, 0, 0)
    assert out_w == x.shape[2], 'the reference never resizes the width axis'
    rows = _nearest_rows if method == ResizeMethod.NEAREST_NEIGHBOR else _bicubic_rows
    return _t(rows(x, out_h, align_corners))
  if method != ResizeMethod.BILINEAR:
    raise NotImplementedError(method)
  ylo, yhi, ylerp = _interp_weights(out_h, x.shape[1], align_corners)
  xlo, xhi, xlerp = _interp_weights(out_w, x.shape[2], align_corners)
  top_rows, bot_rows = x[:, ylo], x[:, yhi]

  def xinterp(rows):
    left, right = rows[:, :, xlo], rows[:, :, xhi]
    return left + (right - left) * xlerp[None, None, :, None]
  top, bottom = xinterp(top_rows), xinterp(bot_rows)
  return _t(top + (bottom - top) * ylerp[None, :, None, None])


# --- keras Layer ------------------------------------------------------------------
class Module:
  pass


class Layer(Module):                                          # (a keras Layer IS a tf.Module: dags.is_module, ddsp/dags.py:40)
  def __init__(self, name=None, trainable=False, **kwargs):
    self._name = name
    self.trainable = trainable
    self.built = False

  @property
  def name(self):
    return self._name

  def build(self, input_shape):
    pass

  def __call__(self, *args, **kwargs):
    return self.call(*args, **kwargs)


def librosa_fft_frequencies(sr=22050, n_fft=2048):
  return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def librosa_a_weighting(frequencies, min_db=-80.0):
  f_sq = np.asanyarray(frequencies) ** 2.0
  const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
  with np.errstate(divide='ignore'):
    weights = 2.0 + 20.0 * (np.log10(const[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + const[0]) - np.log10(f_sq + const[1])
                            - 0.5 * np.log10(f_sq + const[2]) - 0.5 * np.log10(f_sq + const[3]))
  return weights if min_db is None else np.maximum(min_db, weights)


def _identity_decorator(*dargs, **dkwargs):
  """gin.register / gin.configurable: bare or parameterised, both become no-ops."""
  if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
    return dargs[0]
  return lambda f: f


def build_tf_module():
  tf = types.ModuleType('tensorflow')
  tf.Tensor = Tensor
  tf.Module = Module
  tf.newaxis = None
  tf.float32 = np.float32
  tf.int32 = np.int32
  tf.convert_to_tensor = convert_to_tensor
  tf.cast = cast
  tf.cumsum = cumsum
  tf.reduce_sum = reduce_sum
  tf.reduce_mean = lambda x, axis=None, keepdims=False: _t(np.mean(_np(x), axis=axis, keepdims=keepdims, dtype=np.float32))
  tf.where = where
  tf.pad = pad
  tf.linspace = linspace
  tf.complex = complex_
  tf.sin = _wrap(np.sin)
  tf.abs = _wrap(np.abs)
  tf.sqrt = _wrap(np.sqrt)                                   # (processors.Mix: processors.py:225-226)
  tf.exp = _wrap(np.exp)
  tf.concat = lambda values, axis: _t(np.concatenate([_np(v) for v in values], axis=axis))
  tf.zeros_like = _wrap(np.zeros_like)
  tf.ones_like = _wrap(np.ones_like)
  tf.zeros = lambda shape, dtype=np.float32: _t(np.zeros(shape, dtype))
  tf.reshape = lambda x, shape: _t(np.reshape(_np(x), shape))
  tf.transpose = lambda x, perm=None: _t(np.transpose(_np(x), perm))
  tf.tile = lambda x, multiples: _t(np.tile(_np(x), multiples))
  tf.broadcast_to = lambda x, shape: _t(np.broadcast_to(_np(x), tuple(shape)))
  tf.multiply = lambda a, b: _t(_np(a) * _np(b))
  tf.greater_equal = lambda a, b: _t(_np(a) >= b)
  tf.less_equal = lambda a, b: _t(_np(a) <= b)
  tf.equal = lambda a, b: _t(_np(a) == b)
  tf.maximum = _wrap(np.maximum)
  tf.squeeze = lambda x, axis=None: _t(np.squeeze(_np(x), axis))
  tf.function = _identity_decorator
  tf.executing_eagerly = lambda: True
  tf.random = types.SimpleNamespace(
      uniform=lambda shape, minval=0.0, maxval=1.0, dtype=np.float32: _t(
          np.random.default_rng(1234).uniform(minval, maxval, shape).astype(np.float32)))

  # core.diff (core.py:171-199): tf.slice(x, begin, size)
  tf.slice = lambda x, begin, size: _t(_np(x)[tuple(slice(b, b + n) for b, n in zip(begin, size))])

  def cosine_distance(labels, predictions, axis=None, weights=1.0, dim=None):
    """tf.compat.v1.losses.cosine_distance: 1 - sum(labels * predictions, axis, keepdims), then compute_weighted_loss
    with Reduction.SUM_BY_NONZERO_WEIGHTS (sum of the weighted losses over the number of non-zero weights)."""
    axis = axis if axis is not None else dim
    labels, predictions = _np(labels).astype(np.float32), _np(predictions).astype(np.float32)
    losses = np.float32(1.0) - np.sum(labels * predictions, axis=axis, keepdims=True, dtype=np.float32)
    w = np.broadcast_to(np.asarray(weights, np.float32), losses.shape)
    present = np.count_nonzero(w)
    total = np.sum(losses * w, dtype=np.float32)
    return _t(np.float32(total / np.float32(present)) if present else np.float32(0.0))
  tf.losses = types.SimpleNamespace(cosine_distance=cosine_distance)

  tf.math = types.SimpleNamespace(
      log=_wrap(np.log), exp=_wrap(np.exp), real=_wrap(np.real), cumsum=cumsum,
      is_nan=_wrap(np.isnan))
  tf.nn = types.SimpleNamespace(sigmoid=sigmoid)
  tf.signal = types.SimpleNamespace(
      hann_window=hann_window, overlap_and_add=overlap_and_add, frame=frame,
      rfft=rfft, irfft=irfft, fftshift=fftshift, stft=stft)
  tf.keras = types.SimpleNamespace(layers=types.SimpleNamespace(Layer=Layer), Model=Layer)

  v1 = types.SimpleNamespace(
      image=types.SimpleNamespace(resize=image_resize_v1, ResizeMethod=ResizeMethod))
  compat = types.ModuleType('tensorflow.compat')
  v2 = types.ModuleType('tensorflow.compat.v2')
  v2.__dict__.update({k: v for k, v in tf.__dict__.items() if not k.startswith('__')})
  compat.v1, compat.v2 = v1, v2
  tf.compat = compat
  v2.compat = compat
  return tf, compat, v2


def install(reference_root='/root/reference'):
  """Make `import ddsp.core / ddsp.synths / ddsp.processors` run the reference source."""
  tf, compat, v2 = build_tf_module()
  sys.modules['tensorflow'] = tf
  sys.modules['tensorflow.compat'] = compat
  sys.modules['tensorflow.compat.v2'] = v2

  gin = types.ModuleType('gin')
  gin.register = _identity_decorator
  gin.configurable = _identity_decorator
  sys.modules['gin'] = gin

  absl = types.ModuleType('absl')
  import logging as pylogging
  absl.logging = pylogging
  sys.modules['absl'] = absl
  sys.modules['absl.logging'] = pylogging

  # losses.py / spectral_ops.py import these at module level and never touch them on the
  # SpectralLoss(mag, logmag) path
  for name in ('crepe', 'librosa', 'tensorflow_probability'):
    stub = types.ModuleType(name)
    stub.distributions = types.SimpleNamespace(HiddenMarkovModel=object)
    sys.modules[name] = stub
  # librosa (setup.py: 'librosa', unpinned; not under /root/reference): the two functions spectral_ops.compute_loudness calls
  # (spectral_ops.py:307-308), restated from librosa's published source (librosa/core/convert.py, 0.8 .. 0.10: unchanged):
  #   fft_frequencies(sr, n_fft) = linspace(0, sr / 2, 1 + n_fft // 2)                       (np.fft.rfftfreq)
  #   A_weighting(f, min_db=-80) = max(min_db, 2.0 + 20 (log10(c0) + 2 log10(f^2) - log10(f^2 + c0) - log10(f^2 + c1)
  #                                              - 0.5 log10(f^2 + c2) - 0.5 log10(f^2 + c3))),
  #   c = [12194.217, 20.598997, 107.65265, 737.86223]^2          (IEC 61672 A-curve; f = 0 -> -inf -> min_db)
  sys.modules['librosa'].fft_frequencies = librosa_fft_frequencies
  sys.modules['librosa'].A_weighting = librosa_a_weighting

  pkg = types.ModuleType('ddsp')
  pkg.__path__ = [reference_root + '/ddsp']      # namespace only: __init__.py not run
  sys.modules['ddsp'] = pkg
  return tf
