"""Run-to-run and schedule determinism of the default kernels on the MI355X.

Every case of tools/stress_determinism.py - the north-star shape at f0 = 70 / 200 / 333 Hz and with vibrato, config 5's shapes (the
129 .. 200-harmonic instances), frames of 64 / 128 / 192 samples, FilteredNoise with supplied and generated noise, Harmonic with
processors.Add fused in - is launched `ITERS` times on the same inputs with the other synth kernel running on a second stream, as
bench.py runs them; every result is compared BIT FOR BIT with the first on the device, and every fourth launch random rows are run
alone and as a sub-batch and compared with the rows of the full batch (the reference's batch rows are independent and its op chain
deterministic, ddsp/core.py:912-962).

The round-3 binaries fail this test in seconds (profiles/r04_packed_fma_glitch.txt: packed-FMA assembly statements in phase B,
wrong in the last sixteen lanes now and then while a tabulator of the same SIMD runs MFMAs: 3 - 20 % of the launches of config 5's
shapes differed).  300 launches per case take ~10 s of GPU in all."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu

ITERS = 300
GROUPS = {
    'north_star_batch128': 'harm_b128_f0_70,harm_b128_f0_200,harm_b128_f0_333,harm_b128_vibrato_220,fused_add_b128',
    'batch32_and_frame_sizes': 'harm_b32_f0_70,harm_b32_k60_f0_440,harm_b32_k128_hop128,harm_b32_k128_hop64,harm_b32_k100_hop128,'
                               'harm_b32_k64_hop64,harm_b8_k100_hop192,harm_b8_k128_hop192',
    'config5_129_to_200_harmonics': 'harm_config5_b32,harm_config5_b8_f0_120,harm_b32_k160_hop64,harm_b8_k136_hop192,'
                                    'harm_b8_k200_hop192_window,harm_config5_b8_controls',
    'filtered_noise': 'noise_b128,noise_b32',
}


@pytest.mark.parametrize('group', sorted(GROUPS))
def test_launches_are_bit_identical_run_to_run_and_rows_independent(group):
  if not torch.cuda.is_available():
    pytest.skip('gpu tests need a GPU (run with -m gpu on an MI355X)')
  from ddsp_amd import build
  build.build()
  import stress_determinism
  summary = stress_determinism.main(['--iters', str(ITERS), '--cases', GROUPS[group], '--label', group])
  assert summary['cases_run'] >= len(GROUPS[group].split(',')), 'a case name matched nothing: %r' % (summary,)
  assert summary['launches_with_a_difference'] == 0, summary
