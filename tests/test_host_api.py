"""CPU tests of the host layer: the C-ABI library loads and exports every declared symbol,
argument checks raise the reference's errors, and the product path fails loudly (no CPU
fallback) when there is no GPU."""
import os
import re

import numpy as np
import pytest
import torch

from ddsp_amd import _lib, core, processors, synths
from ddsp_amd import build as build_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
  build_mod.build()
  return _lib.load()


def test_library_exports_every_symbol_declared_in_the_header(lib):
  header = open(os.path.join(ROOT, 'include', 'ddsp_amd.h')).read()
  header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
  declared = set(re.findall(r'\b(ddsp_[a-z0-9_]+)\s*\(', header))
  assert declared, 'no declarations parsed'
  assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
  for name in declared:
    assert hasattr(lib, name), name
  assert b'gfx950' in lib.ddsp_version()


def test_fir_size_matches_reference_rule(lib):          # core_test.py:825-855
  assert lib.ddsp_fir_size(1025, 257) == 257
  assert lib.ddsp_fir_size(1025, 256) == 255
  assert lib.ddsp_fir_size(1025, 0) == 2048
  assert lib.ddsp_fir_size(513, 1025 + 100) == 1024
  assert lib.ddsp_fir_size(65, 257) == 128               # ae.gin shape: window > ir size


def test_workspace_queries(lib):
  assert lib.ddsp_harmonic_workspace_bytes(128, 1000, 100, 64000) == max(
      128 * 1000 * 8 + 128 * 1001 * 112 * 4, 2048 * 9 * 112 * 4)
  assert lib.ddsp_filtered_noise_workspace_bytes(128, 1000, 65, 64000, 0) == 128 * 1000 * 128 * 4
  assert lib.ddsp_harmonic_workspace_bytes(0, 10, 10, 10) == 0


def test_null_pointers_and_bad_shapes_return_codes(lib):
  assert lib.ddsp_add_f32(None, None, None, 16, None) == -1
  assert lib.ddsp_harmonic_controls_f32(None, None, None, None, None, 1, 1, 1, 16000, 0, None) == -1
  assert lib.ddsp_uniform_noise_f32(None, 1, 1, 0, 0, None) == -1
  assert lib.ddsp_fir_size(1, 0) == -2


def test_amp_method_errors_match_reference():           # core_test.py:295-381
  with pytest.raises(ValueError, match='is invalid'):
    core._check_amp_method('bogus', 10, 100)
  with pytest.raises(ValueError, match='divisible'):
    core._check_amp_method('window', 10, 105)
  with pytest.raises(ValueError, match='downsampling'):
    core._check_amp_method('window', 10, 5)
  for method in ('window', 'linear', 'nearest', 'cubic'):
    core._check_amp_method(method, 10, 100)
  core._check_amp_method('cubic', 10, 105)               # only 'window' needs divisibility (core.py:687)
  # which argument combinations the closed-form synthesis kernels take; the rest follows the
  # reference's chain of materialised envelopes
  assert core._on_closed_form_kernels('window', 1000, 64000)
  assert core._on_closed_form_kernels('linear', 1000, 64000)
  assert not core._on_closed_form_kernels('cubic', 1000, 64000)
  assert not core._on_closed_form_kernels('nearest', 1000, 64000)
  assert not core._on_closed_form_kernels('linear', 1000, 64001)


@pytest.mark.parametrize('padding', ['same', 'valid'])
@pytest.mark.parametrize('delay', [-1, 0, 3, 40])
@pytest.mark.parametrize('n,f,l', [(64000, 1000, 128), (1000, 1, 10), (10, 1, 100), (250, 25, 7), (777, 7, 16),
                                   (500, 5, 31), (100, 7, 1), (100, 7, 2), (3000, 1, 2000), (96, 1, 200)])
def test_crop_range_is_the_reference_slice(padding, delay, n, f, l):          # core.py:1338-1379
  """The host layer sizes the output by the same python slice the reference takes; compared with the
  oracle's literal restatement (framed FFTs, overlap-add, audio[:, start:-end]) on an index ramp."""
  from oracle import ddsp_oracle as oracle
  audio = np.ones((1, n), np.float64)
  ir = np.ones((1, f, l), np.float64)
  ref = oracle.fft_convolve(audio, ir, padding=padding, delay_compensation=delay, dtype=np.float64)
  start_requested, start, n_out = core._crop_range(n, f, l, padding, delay)
  assert n_out == ref.shape[1]
  assert start_requested == ((l - 1) // 2 - 1 if delay < 0 else delay)
  if n_out:
    # z[m] of all-ones inputs counts the (sample, tap) pairs that land on m: check the first kept sample
    z = np.convolve(np.ones(n), np.ones(l)) if f == 1 else None
    if z is not None and start < z.size:
      assert abs(ref[0, 0] - z[start]) < 1e-6
  with pytest.raises(ValueError, match='Padding'):
    core._crop_range(n, f, l, 'bogus', delay)


def test_frame_count_error_matches_reference():         # core_test.py:868-886
  with pytest.raises(ValueError, match='do not match'):
    core._check_frames(100, 30)
  core._check_frames(100, 7)                              # ragged but consistent (pad_end)
  core._check_frames(64000, 1000)


def test_constructor_defaults_match_reference():        # synths.py:59-66, 153-163
  h = synths.Harmonic()
  assert (h.n_samples, h.sample_rate, h.normalize_below_nyquist, h.amp_resample_method,
          h.use_angular_cumsum, h.name) == (64000, 16000, True, 'window', False, 'harmonic')
  assert h.scale_fn is core.exp_sigmoid
  n = synths.FilteredNoise()
  assert (n.n_samples, n.window_size, n.initial_bias, n.name) == (64000, 257, -5.0,
                                                                  'filtered_noise')
  assert processors.Add().name == 'add'
  for cls in (synths.Harmonic, synths.FilteredNoise, processors.Add):
    assert hasattr(cls, 'get_controls') and hasattr(cls, 'get_signal')   # dags.py:44 duck typing


def test_constructor_defaults_of_the_neighbouring_processors():   # effects.py:123-129, 204-214; processors.py:183, 240; synths.py:27
  from ddsp_amd import effects
  e = effects.ExpDecayReverb()
  assert (e.name, e.trainable, e._reverb_length, e._add_dry) == ('exp_decay_reverb', False, 48000, True)
  assert e._scale_fn is core.exp_sigmoid
  r = effects.Reverb()
  assert (r.name, r.trainable, r._reverb_length, r._add_dry) == ('reverb', False, 48000, True)
  f = effects.FilteredNoiseReverb()
  assert (f.name, f._n_frames, f._n_filter_banks) == ('filtered_noise_reverb', 1000, 16)
  assert (f._synth.n_samples, f._synth.window_size, f._synth.initial_bias) == (48000, 257, -3.0)
  assert effects.FIRFilter().window_size == 257 and effects.FIRFilter().name == 'fir_filter'
  assert processors.Mix().name == 'mix' and synths.TensorToAudio().name == 'tensor_to_audio'
  assert processors.Crop(frame_size=64).crop_location == 'back'
  assert synths.Harmonic.kernel == 'auto' and synths.FilteredNoise.kernel == 'auto'           # the measured defaults
  with pytest.raises(ValueError, match='gain'):                       # effects_test.py:49-52 (raised before any launch)
    effects.ExpDecayReverb(trainable=False).get_controls(np.zeros((1, 8), np.float32))
  with pytest.raises(ValueError, match='ir'):
    effects.Reverb(trainable=False).get_controls(np.zeros((1, 8), np.float32))


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_gpu_fails_loudly_not_silently():
  h = synths.Harmonic(n_samples=640)
  with pytest.raises(_lib.DdspLibraryError, match='no CPU fallback'):
    h(np.zeros((1, 10, 1)), np.zeros((1, 10, 4)), np.full((1, 10, 1), 100.0))


def test_product_package_never_imports_the_oracle():
  pkg = os.path.join(ROOT, 'ddsp_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.hip', '.h')):
        text = open(os.path.join(dirpath, f)).read()
        assert not re.search(r'^\s*(from|import)\s+oracle|#include\s+[<"].*oracle', text, re.M), \
            os.path.join(dirpath, f)


# ---- ProcessorGroup / DAGLayer plumbing (ddsp/processors_test.py:28-101, ddsp/dags.py) ------------
class _Const(processors.Processor):
  """A processor that needs no GPU: returns a constant signal (host plumbing test double)."""

  def __init__(self, value, name):
    super().__init__(name=name)
    self.value = value

  def get_controls(self, x):
    return {'x': x}

  def get_signal(self, x):
    return x + self.value


class _Sum(processors.Processor):
  def __init__(self, name='add'):
    super().__init__(name=name)

  def get_controls(self, signal_one, signal_two):
    return {'signal_one': signal_one, 'signal_two': signal_two}

  def get_signal(self, signal_one, signal_two):
    return signal_one + signal_two


def test_processor_group_dag_construction_and_routing():          # processors_test.py:80-91
  a, b, add = _Const(1.0, 'harmonic'), _Const(10.0, 'filtered_noise'), _Sum()
  dag = [(a, ['amps']), (b, ['magnitudes']), (add, ['filtered_noise/signal', 'harmonic/signal'])]
  group = processors.ProcessorGroup(dag=dag)
  x = {'amps': np.ones(3), 'magnitudes': np.zeros(3)}
  out = group(x, return_outputs_dict=True)
  assert set(out) == {'signal', 'controls'}
  c = out['controls']
  for key in ['inputs', 'amps', 'magnitudes', 'harmonic', 'filtered_noise', 'add', 'out']:
    assert key in c
  assert set(c['harmonic']) == {'signal', 'controls'} and 'x' in c['harmonic']['controls']
  np.testing.assert_array_equal(out['signal'], np.ones(3) + 1.0 + 10.0)
  np.testing.assert_array_equal(group(x), out['signal'])
  assert group.processor_names == ['harmonic', 'filtered_noise', 'add']
  assert group.processors[0] is a and group.harmonic is a
  # string module names resolved from kwargs (gin style), output keys for non-dict returns
  from ddsp_amd import dags
  double = lambda v, **kw: 2 * v
  double.name = 'double'
  layer = dags.DAGLayer([['double', ['inputs/v'], ['twice']], [a, ['double/twice']]], double=double)
  res = layer({'v': np.full(2, 3.0)})
  np.testing.assert_array_equal(res['double']['twice'], np.full(2, 6.0))
  np.testing.assert_array_equal(res['out']['signal'], np.full(2, 7.0))
  with pytest.raises(KeyError, match='nested key'):
    core.nested_lookup('harmonic/nope', res)
  with pytest.raises(ValueError, match='same length'):
    core.to_dict([1, 2], ['only_one'])


def test_reverb_host_contract():                                   # effects.py:31-48, 82-98
  import inspect
  import ddsp_amd as ddsp
  sig = inspect.signature(ddsp.effects.Reverb.__init__)
  assert [(k, v.default) for k, v in list(sig.parameters.items())[1:]] == [
      ('trainable', False), ('reverb_length', 48000), ('add_dry', True), ('name', 'reverb')]
  rev = ddsp.effects.Reverb()
  assert rev.name == 'reverb' and rev.trainable is False
  with pytest.raises(ValueError, match='Must provide "ir" tensor if Reverb trainable=False.'):
    rev.get_controls(audio=None)
  lib = _lib.load()
  # 16 x-blocks + 12 IR partitions of 8192 complex bins each, per clip
  assert lib.ddsp_fft_convolve_long_workspace_bytes(32, 32, 64000, 48000, 0) == 32 * (16 + 12) * 8192 * 8
  assert lib.ddsp_fft_convolve_long_workspace_bytes(32, 1, 64000, 48000, 0) == (32 * 16 + 12) * 8192 * 8
  # programmer errors come back as codes, before any launch
  assert lib.ddsp_fft_convolve_long_f32(None, None, None, None, 0, 1, 1, 10, 10, 0, 0, None) == -1


def test_spectral_loss_host_contract():                            # losses.py:140-187
  import inspect
  import ddsp_amd as ddsp
  sig = inspect.signature(ddsp.losses.SpectralLoss.__init__)
  assert [(k, v.default) for k, v in list(sig.parameters.items())[1:]] == [
      ('fft_sizes', (2048, 1024, 512, 256, 128, 64)), ('loss_type', 'L1'), ('mag_weight', 1.0),
      ('delta_time_weight', 0.0), ('delta_freq_weight', 0.0), ('cumsum_freq_weight', 0.0),
      ('logmag_weight', 0.0), ('loudness_weight', 0.0), ('name', 'spectral_loss')]
  with pytest.raises(ValueError, match='must be "L1", "L2", or "COSINE"'):
    ddsp.losses.SpectralLoss(loss_type='L3')(None, None)
  import ctypes
  lib = _lib.load()
  sizes = (ctypes.c_int * 2)(2048, 64)
  # per-block fp64 pairs: ceil(frames / (4096/S)) blocks per clip
  assert lib.ddsp_spectral_loss_workspace_bytes(4, 64000, sizes, 2) == 4 * (63 + 63) * 16
  bad = (ctypes.c_int * 1)(1000)
  assert lib.ddsp_spectral_loss_workspace_bytes(4, 64000, bad, 1) == 0


def test_crop_processor():                                         # processors.py:237-263
  x = torch.arange(20, dtype=torch.float32).reshape(2, 10)
  class _Core:                                                     # Crop only needs tf_float32: keep the test off the GPU
    tf_float32 = staticmethod(lambda t: t)
  import ddsp_amd.processors as P
  old = P.core
  P.core = _Core
  try:
    assert P.Crop(frame_size=5)(x).tolist() == x[:, :-4].tolist()
    assert P.Crop(frame_size=4, crop_location='front')(x).tolist() == x[:, 4:].tolist()
    assert P.Crop(frame_size=4, crop_location='center')(x).tolist() == x[:, 2:-2].tolist()
    with pytest.raises(ValueError, match='must be "front", "center", or "back"'):
      P.Crop(frame_size=4, crop_location='middle')(x)
  finally:
    P.core = old


def test_fir_size_matches_the_reference_crop_for_every_small_window():
  """ddsp_fir_size (host arithmetic, no GPU): the taps apply_window_to_impulse_response leaves (ddsp/core.py:1477-1531) for every
  window size up to the filter's own - including windows of ONE or TWO samples, whose first slice `ir[L0 - half + 2:]` starts past
  the end and is empty in python: two taps, not one (tools/fuzz_api_vs_reference.py found the mirror's geometry one short)."""
  import numpy as np
  from oracle import ddsp_oracle as O
  lib = _lib.load()
  for m in (3, 5, 9, 17, 65):
    mags = np.ones((1, 1, m), np.float32)
    for ws in list(range(0, 2 * (m - 1) + 3)) + [257]:
      taps = O.frequency_impulse_response(mags, window_size=ws, dtype=np.float32).shape[-1]
      assert lib.ddsp_fir_size(m, ws) == taps, (m, ws, lib.ddsp_fir_size(m, ws), taps)
