"""ctypes binding of the C ABI in include/ddsp_amd.h.

The product path has NO fallback: if `libddsp_amd.so` is missing or a symbol is absent
this module raises, and every processor that needs a kernel fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libddsp_amd.so')

c_f32p = ctypes.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_int, c_uint, c_size_t = ctypes.c_int, ctypes.c_uint, ctypes.c_size_t
c_u64, c_float, c_voidp = ctypes.c_uint64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/ddsp_amd.h
SIGNATURES = {
    'ddsp_version': (ctypes.c_char_p, []),
    'ddsp_harmonic_controls_f32': (c_int, [c_f32p] * 5 + [c_int] * 4 + [c_uint, c_voidp]),
    'ddsp_harmonic_workspace_bytes': (c_size_t, [c_int] * 4),
    'ddsp_harmonic_signal_f32': (c_int, [c_f32p] * 4 + [c_voidp, c_size_t] + [c_int] * 5 +
                                 [c_uint, c_voidp]),
    'ddsp_harmonic_signal_tf_order_f32': (c_int, [c_f32p] * 4 + [c_int] * 5 + [c_uint, c_voidp]),
    'ddsp_harmonic_f32': (c_int, [c_f32p] * 6 + [c_voidp, c_size_t] + [c_int] * 5 +
                          [c_uint, c_voidp]),
    'ddsp_harmonic_add_f32': (c_int, [c_f32p] * 5 + [c_int] * 5 + [c_uint, c_voidp]),
    'ddsp_filtered_noise_controls_f32': (c_int, [c_f32p] * 2 + [c_int] * 3 +
                                         [c_float, c_uint, c_voidp]),
    'ddsp_fir_size': (c_int, [c_int, c_int]),
    'ddsp_frequency_impulse_response_f32': (c_int, [c_f32p] * 2 + [c_int] * 4 + [c_voidp]),
    'ddsp_filtered_noise_workspace_bytes': (c_size_t, [c_int] * 5),
    'ddsp_filtered_noise_f32': (c_int, [c_f32p] * 4 + [c_voidp, c_size_t] + [c_int] * 5 +
                                [c_float, c_uint, c_u64, c_u64, c_voidp]),
    'ddsp_filtered_noise_backward_workspace_bytes': (c_size_t, [c_int] * 4),
    'ddsp_filtered_noise_backward_f32': (c_int, [c_f32p] * 4 + [c_voidp, c_size_t] + [c_int] * 5 +
                                         [c_float, c_uint, c_u64, c_u64, c_voidp]),
    'ddsp_fft_convolve_same_f32': (c_int, [c_f32p] * 3 + [c_int] * 6 + [c_voidp]),
    'ddsp_harmonic_backward_workspace_bytes': (c_size_t, [c_int] * 4),
    'ddsp_harmonic_backward_f32': (c_int, [c_f32p] * 6 + [c_voidp, c_size_t] + [c_int] * 5 +
                                   [c_uint, c_int, c_voidp]),
    'ddsp_harmonic_streaming_f32': (c_int, [c_f32p] * 6 + [c_voidp, c_size_t] + [c_int] * 5 +
                                    [c_uint, c_voidp]),
    'ddsp_fft_convolve_long_workspace_bytes': (c_size_t, [c_int] * 5),
    'ddsp_fft_convolve_long_ex_workspace_bytes': (c_size_t, [c_int] * 6),
    'ddsp_fft_convolve_long_ex_f32': (c_int, [c_f32p] * 3 + [c_voidp, c_size_t] + [c_int] * 6 +
                                      [c_uint, c_voidp]),
    'ddsp_fft_convolve_long_f32': (c_int, [c_f32p] * 3 + [c_voidp, c_size_t] + [c_int] * 5 +
                                   [c_uint, c_voidp]),
    'ddsp_spectral_loss_workspace_bytes': (c_size_t, [c_int, c_int, ctypes.POINTER(c_int), c_int]),
    'ddsp_spectral_loss_f32': (c_int, [c_f32p] * 3 + [c_voidp, c_size_t, c_int, c_int,
                                                      ctypes.POINTER(c_int), c_int, c_float, c_float,
                                                      c_voidp]),
    'ddsp_spectral_loss_backward_f32': (c_int, [c_f32p] * 4 + [c_int, c_int, ctypes.POINTER(c_int), c_int,
                                                           c_float, c_float, c_voidp]),
    'ddsp_spectral_loss_value_and_grad_f32': (c_int, [c_f32p] * 4 + [c_voidp, c_size_t, c_int, c_int,
                                                                 ctypes.POINTER(c_int), c_int, c_float,
                                                                 c_float, c_voidp]),
    'ddsp_stft_mag_f32': (c_int, [c_f32p] * 4 + [c_int] * 3 + [c_voidp]),
    'ddsp_spectral_terms_workspace_bytes': (c_size_t, [c_int] * 2),
    'ddsp_spectral_terms_f32': (c_int, [c_f32p] * 3 + [c_int] * 3 + [c_f32p, c_voidp, c_f32p, c_voidp, c_size_t] +
                                [c_int] * 4 + [ctypes.c_float] * 5 + [c_int, c_voidp]),
    'ddsp_stft_mag_backward_f32': (c_int, [c_f32p] * 3 + [c_int] * 3 + [c_voidp]),
    'ddsp_window_impulse_response_size': (c_int, [c_int, c_int]),
    'ddsp_apply_window_to_impulse_response_f32': (c_int, [c_f32p, c_f32p, ctypes.c_long, c_int, c_int, c_int, c_voidp]),
    'ddsp_stft_frames_mag_f32': (c_int, [c_f32p] * 2 + [c_int] * 6 + [c_voidp]),
    'ddsp_stft_frames_f32': (c_int, [c_f32p] * 2 + [c_int] * 7 + [c_voidp]),
    'ddsp_stft_frames_mag_ex_f32': (c_int, [c_f32p] * 2 + [c_int] * 7 + [c_voidp]),
    'ddsp_stft_frames_mag_backward_f32': (c_int, [c_f32p] * 3 + [c_int] * 6 + [c_voidp]),
    'ddsp_loudness_from_mag_f32': (c_int, [c_f32p] * 3 + [c_int] * 3 + [ctypes.c_float] * 2 + [c_voidp]),
    'ddsp_loudness_from_mag_backward_f32': (c_int, [c_f32p] * 4 + [c_int] * 3 + [ctypes.c_float] * 2 + [c_voidp]),
    'ddsp_uniform_noise_f32': (c_int, [c_f32p, c_int, c_int, c_u64, c_u64, c_voidp]),
    'ddsp_prepare': (c_int, [c_int, c_int, c_int]),
    'ddsp_uniform_noise_ex_f32': (c_int, [c_f32p, c_int, c_int, c_u64, c_u64, c_int, c_voidp]),
    'ddsp_add_f32': (c_int, [c_f32p] * 3 + [c_size_t, c_voidp]),
    'ddsp_exp_sigmoid_f32': (c_int, [c_f32p] * 2 + [c_size_t] + [c_float] * 3 + [c_voidp]),
    'ddsp_oscillator_bank_workspace_bytes': (c_size_t, [c_int] * 3),
    'ddsp_oscillator_bank_f32': (c_int, [c_f32p] * 3 + [c_voidp, c_size_t] + [c_int] * 5 + [c_voidp]),
    'ddsp_resample_f32': (c_int, [c_f32p] * 2 + [c_int] * 5 + [c_voidp]),
    'ddsp_resample_ex_f32': (c_int, [c_f32p] * 2 + [c_int] * 6 + [c_voidp]),
    'ddsp_sum_rows_f32': (c_int, [c_f32p] * 2 + [c_int] * 3 + [c_voidp]),
    'ddsp_resample_ex_backward_f32': (c_int, [c_f32p] * 2 + [c_int] * 6 + [c_voidp]),
    'ddsp_oscillator_bank_grad_amplitudes_f32': (c_int, [c_f32p] * 3 + [c_voidp, c_size_t] + [c_int] * 4 + [c_voidp]),
    'ddsp_oscillator_bank_grad_frequencies_f32': (c_int, [c_f32p] * 4 + [c_voidp, c_size_t] + [c_int] * 4 + [c_voidp]),
    'ddsp_harmonic_frequencies_backward_f32': (c_int, [c_f32p] * 3 + [c_int] * 3 + [c_voidp]),
    'ddsp_harmonic_controls_backward_f32': (c_int, [c_f32p] * 6 + [c_int] * 4 + [c_uint, c_int, c_voidp]),
    'ddsp_fft_convolve_f32': (c_int, [c_f32p] * 3 + [c_int] * 7 + [c_voidp]),
    'ddsp_harmonic_envelopes_f32': (c_int, [c_f32p] * 6 + [c_int] * 3 + [c_voidp]),
    'ddsp_scale_f32': (c_int, [c_f32p] * 3 + [c_size_t, c_voidp]),
    'ddsp_harmonic_oscillator_bank_workspace_bytes': (c_size_t, [c_int] * 2),
    'ddsp_harmonic_oscillator_bank_f32': (c_int, [c_f32p] * 5 + [c_voidp, c_size_t] + [c_int] * 5 + [c_voidp]),
    'ddsp_harmonic_f0_grad_workspace_bytes': (c_size_t, [c_int] * 4),
    'ddsp_harmonic_f0_grad_f32': (c_int, [c_f32p] * 5 + [c_voidp, c_size_t] + [c_int] * 5 +
                                  [c_uint, c_voidp]),
    'ddsp_exp_decay_ir_f32': (c_int, [c_f32p] * 4 + [c_int] * 2 + [c_uint, c_voidp]),
    'ddsp_exp_decay_ir_backward_workspace_bytes': (c_size_t, [c_int] * 2),
    'ddsp_exp_decay_ir_backward_f32': (c_int, [c_f32p] * 6 + [c_voidp, c_size_t] + [c_int] * 2 + [c_uint, c_voidp]),
    'ddsp_sigmoid_f32': (c_int, [c_f32p] * 2 + [c_size_t, c_voidp]),
    'ddsp_mix_f32': (c_int, [c_f32p] * 4 + [c_size_t, c_int, c_voidp]),
    'ddsp_sigmoid_backward_f32': (c_int, [c_f32p] * 3 + [c_size_t, c_voidp]),
    'ddsp_mix_backward_f32': (c_int, [c_f32p] * 7 + [c_size_t, c_int, c_voidp]),
    'ddsp_safe_divide_f32': (c_int, [c_f32p] * 3 + [c_size_t, c_int, c_int, ctypes.c_float, c_voidp]),
    'ddsp_safe_log_f32': (c_int, [c_f32p] * 2 + [c_size_t, ctypes.c_float, c_voidp]),
    'ddsp_harmonic_frequencies_f32': (c_int, [c_f32p] * 2 + [c_size_t, c_int, c_voidp]),
    'ddsp_remove_above_nyquist_f32': (c_int, [c_f32p] * 3 + [c_size_t, c_float, c_voidp]),
    'ddsp_angular_cumsum_workspace_bytes': (c_size_t, [c_int] * 3),
    'ddsp_angular_cumsum_f32': (c_int, [c_f32p] * 2 + [c_voidp, c_size_t] + [c_int] * 3 + [c_voidp]),
    'ddsp_profile_kernel_count': (c_int, []),
    'ddsp_profile_kernel_name': (ctypes.c_char_p, [c_int]),
    'ddsp_profile_begin': (c_int, [c_uint, c_int]),
    'ddsp_profile_begin_sampled': (c_int, [c_uint, c_int, c_int]),
    'ddsp_profile_end': (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
}

# flags (mirror include/ddsp_amd.h)
HARM_SCALE_EXP_SIGMOID = 0x1
HARM_NORMALIZE_NYQUIST = 0x2
HARM_AMP_LINEAR = 0x4
HARM_ANGULAR_CUMSUM = 0x8
HARM_INPUTS_ARE_AMPLITUDES = 0x20
HARM_DIRECT_SUM = 0x40
NOISE_SCALE_EXP_SIGMOID = 0x1
NOISE_FIR_VECTOR_ALU = 0x8
NOISE_BITS_23 = 0x10
DECAY_SCALE_EXP_SIGMOID = 0x1
RESAMPLE_METHODS = {'nearest': 0, 'linear': 1, 'cubic': 2, 'window': 3}
LOSS_TYPES = {'L1': 0, 'L2': 1, 'COSINE': 2}
CONV_ADD_DRY = 0x1
CONV_MASK_TAP0 = 0x2
CONV_REVERSE_AUDIO = 0x4
CONV_REVERSE_IR = 0x8
CONV_REVERSE_OUT = 0x10
CONV_ZERO_OUT0 = 0x20

ERR_UNSUPPORTED = -3
ERRORS = {-1: 'DDSP_ERR_NULL_POINTER', -2: 'DDSP_ERR_BAD_SHAPE', -3: 'DDSP_ERR_UNSUPPORTED',
          -4: 'DDSP_ERR_WORKSPACE', -5: 'DDSP_ERR_LAUNCH'}

_lib = None


class DdspLibraryError(RuntimeError):
  pass


def load():
  """Load libddsp_amd.so (once) and type every entry point.  Raises if unavailable."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise DdspLibraryError(
        'HIP library not built: %s is missing. Run `python -c "import __graft_entry__ as g; '
        'g.build()"` (or `python -m ddsp_amd.build`). There is no CPU fallback.' % LIB_PATH)
  # torch must be imported first so that libamdhip64.so.7 resolves to the HIP runtime torch
  # already loaded: one runtime per process, device pointers are shared with torch.
  import torch  # noqa: F401
  lib = ctypes.CDLL(LIB_PATH)
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)       # AttributeError here == a declared symbol is not exported
    fn.restype = restype
    fn.argtypes = argtypes
  _lib = lib
  return lib


def check(rc, what):
  if rc != 0:
    raise DdspLibraryError('%s failed: %s (%d)' % (what, ERRORS.get(rc, 'unknown'), rc))


def profile_begin(kernel_names=None, max_records=4096, stride=1):
  """Start per-kernel HIP-event tracing; kernel_names=None traces every kernel.

  stride=n brackets only every n-th launch of each selected kernel (a bracketed launch costs
  ~5 us of queue time, so a timed region samples instead of bracketing everything)."""
  lib = load()
  n = lib.ddsp_profile_kernel_count()
  names = [lib.ddsp_profile_kernel_name(i).decode() for i in range(n)]
  mask = 0
  for i, nm in enumerate(names):
    if kernel_names is None or nm in kernel_names:
      mask |= 1 << i
  check(lib.ddsp_profile_begin_sampled(mask, int(max_records), int(stride)),
        'ddsp_profile_begin_sampled')


def profile_end():
  """Stop tracing; returns {kernel_name: (total_ms, count)} for kernels that ran."""
  lib = load()
  n = lib.ddsp_profile_kernel_count()
  ms = (ctypes.c_double * n)()
  cnt = (c_int * n)()
  check(lib.ddsp_profile_end(ms, cnt), 'ddsp_profile_end')
  return {lib.ddsp_profile_kernel_name(i).decode(): (ms[i], cnt[i])
          for i in range(n) if cnt[i] > 0}
