"""The GPU parity tests, re-run on the CPU through the SIMT emulation of tests/hip_emu (TEST INFRASTRUCTURE).

tests/hip_emu/emu_simt.py compiles the gfx950 kernel sources of ddsp_amd/csrc unchanged for the host (threads of
a block as fibers, wavefront operations - DPP, readlane, swizzle, MFMA - evaluated when all live lanes have
arrived); this module points the Python layer of ddsp_amd at that build and at host memory, then runs the very
test functions of test_gpu_parity.py / test_gpu_parity_general.py whose shapes finish in seconds.  What it
checks is kernel LOGIC (indexing, LDS layouts, barriers, cross-lane traffic, MFMA fragment layouts, launch
geometry) and the host glue; it says nothing about performance and does not replace the `-m gpu` run, which is
the parity gate.  Every kernel exercised here has also passed on the MI355X, which is what pins the emulation
itself; the point of keeping it is that a kernel change can be checked without GPU time.

The product never runs this way: ddsp_amd has no CPU path (test_host_api.test_no_gpu_fails_loudly_not_silently)."""
import os

import pytest
import torch

import test_gpu_parity as P
import test_gpu_parity_general as G
import test_gpu_reference_tests as R
from ddsp_amd import _lib, core
from tests.hip_emu import emu_simt


@pytest.fixture(scope='module')
def ddsp():
  import ddsp_amd
  os.environ.setdefault('DDSP_EMU_CUS', '4')          # the emulated chip's CU count (persistent kernels size their grid by it)
  if not os.path.exists(emu_simt.CLANG):
    pytest.skip('the SIMT emulation builds with the ROCm clang++ (%s), which this machine does not have' % emu_simt.CLANG)
  lib = emu_simt.load()
  saved = (_lib.load, core._device, core._stream, dict(core._ws_bytes_cache), P.DEV, G.DEV, R.DEV)
  _lib.load = lambda: lib
  core._device = lambda: torch.device('cpu')
  core._stream = lambda: None
  core._ws_bytes_cache.clear()
  P.DEV = G.DEV = R.DEV = 'cpu'
  yield ddsp_amd
  _lib.load, core._device, core._stream = saved[0], saved[1], saved[2]
  core._ws_bytes_cache.clear()
  core._ws_bytes_cache.update(saved[3])
  P.DEV, G.DEV, R.DEV = saved[4], saved[5], saved[6]


@pytest.fixture(params=['auto', 'direct'])
def harm_kernel(request, ddsp):
  old = ddsp.synths.Harmonic.kernel
  ddsp.synths.Harmonic.kernel = request.param
  yield request.param
  ddsp.synths.Harmonic.kernel = old


@pytest.fixture(params=['auto', 'vector'])
def noise_kernel(request, ddsp):
  old = ddsp.synths.FilteredNoise.kernel
  ddsp.synths.FilteredNoise.kernel = request.param
  yield request.param
  ddsp.synths.FilteredNoise.kernel = old


# Every test function of the two GPU modules is re-exported under its own name (so its parametrisation comes
# along); the cases below are left to the GPU run - minutes each under the emulation (clips of 4 s at batch 32).
# DDSP_EMU_ALL=1 runs them too (about 17 minutes in all; every one of them passes).
for _module in (P, G, R):
  for _name in dir(_module):
    if _name.startswith('test_') and callable(getattr(_module, _name)):
      globals()[_name] = getattr(_module, _name)

SLOW_UNDER_EMULATION = () if os.environ.get('DDSP_EMU_ALL') == '1' else (
    'test_processors_group_dag_construction',                          # 4 x 64 000 samples, 256 bands, a 48 000-tap reverb
    'test_synths_filtered_noise_output_shape',                         # 26 s: 16 000 frames of one sample on the plain sum
    'test_synths_harmonic_output_shape',
    # the reference's own accuracy tests at its own sizes (batch 16 x 4000 x 1 .. 2 x 32 000 x 5 on the stand-alone oscillator bank,
    # one thread per sample and sinusoid under the emulation): 28 - 55 s each, on the GPU milliseconds
    'test_core_oscillator_bank_is_accurate', 'test_core_harmonic_synthesis_multiple_harmonics', 'test_core_silent_above_nyquist',
    'test_core_oscillator_bank_shape_is_correct',
    'test_vst_48k_configuration_full_size',                            # minutes: 2 x 192 960 samples through every kernel
    'test_spectral_loss_on_the_synth_output_batch32',                  # 293 s
    'test_harmonic_backward_full_size_properties',                     # 210 s
    'test_full_size_properties_batch32',                               # 49 + 33 s
    'test_reverb_properties_full_size_batch32',                        # 46 s
    'test_tf_op_order_kernel_matches_faithful_oracle_full_length',     # 4 x 36 s
    'test_spectral_loss_vs_fp64_oracle[2-64000]',                      # 30 s
    'test_standalone_oscillator_bank',                                 # 26 s
    'test_training_loop_with_native_loss',                             # 19 s
    'test_reference_shape_tests',                                      # 17 s
    'test_harmonic_fused_unit_edges[auto-5-1000]', 'test_harmonic_fused_unit_edges[direct-5-1000]',   # 16 + 11 s
    'test_harmonic_canonical_vs_truth_and_faithful[auto-200.0]',       # 4 x 6-8 s: one of the four stays
    'test_harmonic_canonical_vs_truth_and_faithful[direct-70.0]',
    'test_harmonic_canonical_vs_truth_and_faithful[direct-200.0]',
    'test_spectral_loss_every_term_golden_and_gradient',               # 46 s (round 4: the CPU suite is run serially by the driver)
    'test_exp_decay_reverb_reference_tests_and_gradients',             # 37 s
    'test_filtered_noise_matrix_core_kernel_vs_oracle_and_vector_kernel[770-1-64]',      # 34 s: the two smaller cases stay
    'test_spectral_loss_vs_fp64_oracle[3-12345]',                      # 16 s
)
