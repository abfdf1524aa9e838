#!/usr/bin/env python
"""bench.py - Msamples/s of the Harmonic + FilteredNoise hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): bench.py starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`, one
rank per GPU over RCCL) and passes rank 0's JSON line through; under an external torchrun it is one of the ranks.
It never prints `n_gpus: 1` when more were asked for.

One "step" = one pass of the hot path over one batch of synthetic control tensors:
`synths.Harmonic(amplitudes, harmonic_distribution, f0_hz)` + `synths.FilteredNoise(magnitudes)`
(raw network outputs in, both get_controls prologues included, noise generated on chip),
each output sample counted once.  Inputs are resident in HBM before the timed region.
Workload (per GPU, weak scaling): the shape BASELINE.json's metric and target are quoted on - batch 128 of
configs[1]'s clips: 4 s @ 16 kHz, F=1000 frames, K=100 harmonics (all below Nyquist: f0 = 70 + N(0,1) Hz), M=65
noise bands.  configs[1] itself (batch 32) is timed the same way right after and reported as `configs_1`.

Timing.  After W warm-up steps (and `--settle` seconds of untimed load: clock settle) the region "exactly K steps, barrier +
torch.cuda.synchronize() on both sides" is run `--repeats` times (>= 10 by default).  Each region is timed
twice: by the host clock around the bracket, and by HIP events recorded on the stream(s) at its first and after
its last launch (the two-stream mode forks from / joins into the base stream with events).  `ms_per_step` and
`value` are the MEDIAN event-timed region (max over ranks) - at the driver's K = 20 a region is under a
millisecond and the host-side synchronize alone is 2-8 % of it; the host-clock median and the first region are
reported beside it (`timing`).

The JSON line also carries
  roofline         : dominant kernel, algorithmic bytes per launch / its mean duration measured
                     with HIP events on the launch stream during the timed regions (ddsp_profile_*),
                     against the 8 TB/s HBM peak (the path is in fact ALU-bound).
  configs_1        : the same step at BASELINE.json configs[1] (batch 32 per GPU; two streams, `one_stream` beside it),
                     timed the same way right after the headline: ms_per_step, value, whole-step and dominant-kernel
                     fractions of the HBM roofline.  (Rounds 1-2 had the two shapes the other way round.)
  configs_4        : likewise BASELINE.json configs[4] per GPU (48 kHz, 200 harmonics, 10 s clips, batch 32): the Harmonic
                     kernel's instances for 129 .. 200 harmonics.
  configs_2        : BASELINE.json configs[2] - the synths' DAG (FilteredNoise, Harmonic with processors.Add fused in) followed by
                     losses.SpectralLoss (six STFT scales 2048 .. 64, mag + logmag: gin/models/ae.gin:36-41), batch 128: one stream,
                     regions like configs_4's; per-kernel breakdown, algorithmic bytes, whole-step fraction; the loss's value +
                     gradient w.r.t. the audio (what a training step runs) beside the forward value.
  configs_3        : BASELINE.json configs[3] per GPU - the same DAG followed by effects.Reverb with the ONE trainable 48 000-tap
                     impulse response of gin/models/solo_instrument.gin:26-40, batch 128 (1024 over 8 GPUs); an impulse response
                     per clip beside it.
  fnoise_11_bit_levels : the headline step with FilteredNoise(noise_bits=11) - generated noise of 2048 levels, every sample an fp16
                     number - beside the headline, which since round 6 draws the 2^23 levels of the reference's tf.random.uniform
                     (FilteredNoise's default).  (--noise-bits 11 swaps them: the side block is then
                     fnoise_full_resolution.)
  cpu_baseline     : the numpy fp32 oracle ("port" of the TF op chain; TF itself cannot run
                     here) timed on this host's cores (concurrent worker processes) on a bounded
                     sample of the same workload.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DTYPE = ('f32 (Harmonic wavetables and the FilteredNoise FIR/IR products: 22-bit operands - fp16 hi/lo pairs on the MFMA matrix cores - '
         'with fp32 accumulation; everything else plain fp32, the frame phase prefix fp64)')


def parse_args(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--warmup', type=int, default=500)   # ~20 ms of load: the GPU needs that long to reach its sustained clock (43 vs 37 us/step)
  ap.add_argument('--settle', type=float, default=0.5,
                  help='seconds of untimed load after the W warm-up steps and before the timed regions: a fresh box reaches '
                       'its sustained clocks only after some tenths of a second (r03p: the same kernels 4 %% slower in a process '
                       'that timed its 20-step regions 50 ms after its first launch)')
  ap.add_argument('--repeats', type=int, default=0,
                  help='timed regions of K steps each (0: at least 10, more when a region is short, so that '
                       'about 0.1 s is measured in all); the median is reported')
  ap.add_argument('--batch', type=int, default=128,
                  help='clips per GPU (128: the shape the north-star target is quoted on; configs[1] has 32)')
  ap.add_argument('--n-frames', type=int, default=1000)
  ap.add_argument('--n-harmonics', type=int, default=100)
  ap.add_argument('--n-bands', type=int, default=65)
  ap.add_argument('--n-samples', type=int, default=64000)
  ap.add_argument('--sample-rate', type=int, default=16000)
  ap.add_argument('--f0', type=float, default=70.0, help='f0 centre in Hz (70: all harmonics live)')
  ap.add_argument('--cpu-clips', type=int, default=6, help='clips per worker process of the CPU oracle leg')
  ap.add_argument('--cpu-procs', type=int, default=0,
                  help='worker processes of the CPU oracle leg (0: the physical cores this process may run on, '
                       'bounded by free memory)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--event-stride', type=int, default=8,
                  help='bracket every n-th launch of the dominant kernel with HIP events inside the '
                       'timed regions (a bracketed launch costs ~5 us of queue time; 1 = all of them)')
  ap.add_argument('--streams', choices=['auto', '1', '2'], default='auto',
                  help='issue the two Processor calls on one stream or on two free-running streams; auto = two '
                       '(the two persistent kernels overlap only where one has left CUs idle)')
  ap.add_argument('--no-overlap', action='store_true',
                  help='issue the two Processor calls back to back on one stream instead of on two '
                       'free-running HIP streams')
  ap.add_argument('--also-other-mode', action='store_true',
                  help='after the timed regions, run K more steps in the other stream mode and report '
                       'them as other_issue_mode (off by default so that a rocprofv3 trace of the '
                       'default command sees one mode only)')
  ap.add_argument('--no-aux', action='store_true',
                  help='skip the auxiliary yardsticks after the timed regions (measured device-copy bandwidth, '
                       'the f0 = 200 Hz regime of SURVEY.md 8d)')
  ap.add_argument('--no-second-shape', '--no-north-star', dest='no_second_shape', action='store_true',
                  help='skip the second block (configs[1], batch 32 per GPU)')
  ap.add_argument('--no-other-configs', action='store_true',
                  help='skip the configs_2 (SpectralLoss), configs_3 (Reverb) and fnoise_full_resolution blocks')
  ap.add_argument('--noise-bits', type=int, default=23, choices=[11, 23],
                  help="FilteredNoise(noise_bits=...) of the headline step (23, the class default: the 2^23 levels of "
                       "tf.random.uniform; 11: 2048 levels, every sample an fp16 number - reported beside the headline as "
                       "fnoise_11_bit_levels in the default run)")
  ap.add_argument('--second-batch', '--north-star-batch', dest='second_batch', type=int, default=32,
                  help='clips per GPU of the second block (32 = BASELINE configs[1], reported as `configs_1`)')
  ap.add_argument('--harm-kernel', default='auto', help="Harmonic.kernel ('auto', 'direct', ...)")
  ap.add_argument('--noise-kernel', default='auto', help="FilteredNoise.kernel ('auto', ...)")
  ap.add_argument('--allgather', action='store_true',
                  help='also time an RCCL all_gather of the audio (reported separately)')
  ap.add_argument('--dry-run', action='store_true',
                  help='PLUMBING TEST ONLY (tests/test_bench_contract.py): no device, no kernels - the step is a '
                       'sleep, the process group is gloo; exercises the launcher, the rank logic, the collectives '
                       'and the JSON line on a CPU box.  The line says "dry_run": true and measures nothing.')
  return ap.parse_args(argv)


def make_inputs(batch, a, seed):
  rng = np.random.default_rng(seed)
  return dict(
      amplitudes=rng.standard_normal((batch, a.n_frames, 1)).astype(np.float32),
      harmonic_distribution=rng.standard_normal((batch, a.n_frames, a.n_harmonics)).astype(np.float32),
      f0_hz=(a.f0 + rng.standard_normal((batch, a.n_frames, 1))).astype(np.float32),
      magnitudes=rng.standard_normal((batch, a.n_frames, a.n_bands)).astype(np.float32))


def algorithmic_bytes(a, batch):
  """controls read once, each synth's audio written once, noise on chip."""
  harm = 4 * batch * (a.n_frames * (a.n_harmonics + 2) + a.n_samples)
  noise = 4 * batch * (a.n_frames * a.n_bands + a.n_samples)
  return harm, noise


FP32_VECTOR_PEAK_TFLOPS = 157.3   # 256 CUs x 2.4 GHz x 256 flop/clk


def algorithmic_flops(a, batch):
  """SURVEY.md 8(d), the ALU note beside the HBM roofline: 3 FMAs per (sample, live harmonic) for the
  oscillator bank, 2 L flops per sample for the L-tap time-varying FIR (L = 2 (M - 1))."""
  live = min(a.n_harmonics, int((a.sample_rate / 2.0) / max(a.f0, 1e-6)))
  harm = batch * a.n_samples * 3 * live * 2
  noise = batch * a.n_samples * 2 * 2 * (a.n_bands - 1)
  return harm, noise


def cpu_baseline(a):
  """The oracle (numpy, fp32, op by op as TF executes it) on the host's cores: oracle/cpu_baseline.py runs
  P concurrent worker processes of `cpu_clips` clips each (TF's CPU kernels are multi-threaded; `cores` = the
  workers actually used, the one-process figure is reported beside it).  The op chain is memory-bound (it
  materialises the [N, K] tensors as TF does), so more workers is not monotonically faster - measured on the
  2 x 64-core GPU host: 32 workers 10.5 Msamples/s, 128 workers 3.4.  With --cpu-procs 0 the leg therefore tries
  32, 64 and the physical core count (each bounded by the cores there are) and reports the BEST as the baseline,
  listing every count it tried."""
  from oracle import cpu_baseline as leg
  shape = (a.n_frames, a.n_harmonics, a.n_bands, a.n_samples, a.sample_rate, a.f0)
  if a.cpu_procs > 0:
    return leg.measure(a.cpu_clips, a.cpu_procs, *shape)
  top = leg.default_procs()
  tried, best = [], None
  for procs in sorted({min(32, top), min(64, top), top}):
    r = leg.measure(a.cpu_clips if procs <= 64 else max(2, a.cpu_clips // 2), procs, *shape)
    tried.append({'workers': r['cores'], 'value': r['value']})
    if best is None or r['value'] > best['value']:
      best = r
    elif r['value'] < 0.7 * best['value']:
      break                       # past the memory-bandwidth knee: larger counts only get slower (and take longer)
  best['worker_counts_tried'] = tried
  # BASELINE configs[0] - "synths.Harmonic on reference TF CPU: batch=1, 16kHz, 1000 frames, 60 harmonics" - is this leg alone:
  # one process, Harmonic only, 60 harmonics (gin/models/solo_instrument.gin:18-20)
  best['configs_0'] = leg.measure_config0()
  return best


DEFAULT_SHAPE = {'n_frames': 1000, 'n_harmonics': 100, 'n_samples': 64000, 'sample_rate': 16000}


def _pmc_records():
  """The committed PMC passes (profiles/pmc_*.json): one record per (batch, shape) - tools/pmc_traffic.py, tools/pmc_summary.py."""
  import glob
  out = []
  for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_*.json'))):
    try:
      with open(path) as f:
        rec = json.load(f)
      if isinstance(rec, dict) and 'batch' in rec:
        out.append(rec)
    except (ValueError, OSError):
      pass
  return out


def _pmc_match(rec, a, batch):
  """A PMC record belongs to a launch of the same kernel at the same batch AND shape (a record without shape fields is
  one of the default shape: rounds 1-3 wrote those)."""
  shape = dict(DEFAULT_SHAPE, **rec.get('shape', {}))
  return rec.get('batch') == batch and all(shape[k] == getattr(a, k) for k in DEFAULT_SHAPE)


def load_traffic(a, dominant, batch):
  """HBM bytes per launch of `dominant` from the committed PMC passes of THIS shape, or None (round 3 keyed the lookup on
  the batch alone and handed configs[4]'s 126 MB launch the 16 kHz figure: VERDICT r3, weak #6)."""
  for rec in _pmc_records():
    if _pmc_match(rec, a, batch) and rec.get('kernels', {}).get(dominant) is not None:
      return rec['kernels'][dominant]
  return None


def load_issue_counters(a, dominant, batch, avg_launch_s):
  """What the dominant kernel EXECUTES, from the committed SQ counter pass of this shape (profiles/pmc_sq_*.json): wavefront
  instructions per launch by class and per SIMD clock.  None when no such pass is committed."""
  for rec in _pmc_records():
    c = rec.get('sq', {}).get(dominant)
    if _pmc_match(rec, a, batch) and c:
      insts = {k: c[k] for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM', 'SQ_INSTS_VMEM_RD',
                                 'SQ_INSTS_VMEM_WR') if k in c}
      total = sum(insts.values())
      simd_clocks = 1024 * avg_launch_s * 2.4e9                 # 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock
      out = {'source': rec.get('source', 'profiles/pmc_sq_*.json'), 'wave_instructions_per_launch': total,
             'by_class': insts, 'per_simd_clock': total / simd_clocks,
             'note': 'a SIMD issues at most one instruction per clock; with four busy wavefronts one per 2.15 (plain fp32) to '
                     '3.3 (fp64, conversions, DPP, packed) to 5.5 clocks (transcendental), HISTORY.md section 4'}
      if c.get('SQ_WAVE_CYCLES') and c.get('SQ_WAIT_INST_ANY') is not None:
        out['wave_cycles_waiting_for_an_instruction'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
      if c.get('SQ_LDS_IDX_ACTIVE') and c.get('SQ_LDS_BANK_CONFLICT') is not None:
        out['lds_bank_conflict_share_of_lds_cycles'] = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']
      return out
  return None


def kernel_roofline(a, batch, dominant, dom_ms, dom_n):
  """achieved = algorithmic bytes of the kernel's Processor / its mean launch duration."""
  harm_bytes, noise_bytes = algorithmic_bytes(a, batch)
  harm_flops, noise_flops = algorithmic_flops(a, batch)
  dom_avg_s = dom_ms / dom_n * 1e-3
  is_harm = dominant.startswith('harm')
  dom_bytes = harm_bytes if is_harm else noise_bytes
  if dominant.startswith('synth'):          # a fused Harmonic + FilteredNoise launch carries both
    dom_bytes = harm_bytes + noise_bytes
  achieved = dom_bytes / dom_avg_s / 1e9
  dom_flops = (harm_flops + noise_flops) if dominant.startswith('synth') else (harm_flops if is_harm else noise_flops)
  return {'kernel': dominant, 'achieved': achieved, 'frac': achieved / HBM_PEAK_GBS,
          'traffic': load_traffic(a, dominant, batch), 'algorithmic_bytes_per_launch': dom_bytes,
          'avg_launch_us': dom_avg_s * 1e6, 'launches': dom_n, '_flops': dom_flops, '_avg_s': dom_avg_s}


def build_result(a, world, B, elapsed, prof, breakdown, dominant, overlap, aux=None, alt_elapsed=None,
                 gather_ms=None, cpu_baseline_fn=None, timing=None, second=None, one_stream_elapsed=None,
                 roofline_timing=None, fused_add=None):
  """The JSON line of the bench contract from the measured quantities (pure: unit-tested on the CPU).
  elapsed: seconds of the median K-step region, max over ranks; prof / breakdown: {kernel: (total_ms, launches)}
  of the timed regions' sampled dispatch events / of the untimed single-stream diagnostic pass."""
  aux = aux or {}
  cpu_baseline_fn = cpu_baseline_fn or cpu_baseline
  total_samples = world * B * a.n_samples * a.steps
  value = total_samples / elapsed / 1e6
  harm_bytes, noise_bytes = algorithmic_bytes(a, B)
  roof = kernel_roofline(a, B, dominant, *prof[dominant])
  dom_flops, dom_avg_s = roof.pop('_flops'), roof.pop('_avg_s')
  step_bytes = harm_bytes + noise_bytes
  harm_flops, noise_flops = algorithmic_flops(a, B)
  result = {
      'metric': 'Msamples/s (Harmonic+FilteredNoise, 16kHz, 100 harmonics)',
      'value': value, 'unit': 'Msamples/s', 'n_gpus': world, 'steps': a.steps,
      'warmup': a.warmup, 'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
      'config': {
          'workload': '%s: Harmonic+FilteredNoise, batch=%d per GPU, %d samples '
                      '@ %d Hz, %d frames, %d harmonics (f0=%g+N(0,1) Hz), %d noise bands, raw '
                      'controls in (get_controls fused), noise generated on chip (%s)' %
                      (shape_name(B), B, a.n_samples, a.sample_rate, a.n_frames, a.n_harmonics, a.f0, a.n_bands,
                       '2048 levels: FilteredNoise(noise_bits=11)' if a.noise_bits == 11
                       else "2^23 levels, tf.random.uniform's resolution: FilteredNoise(noise_bits=23), the default"),
          'batch_per_gpu': B, 'global_batch': world * B, 'parallelism': 'batch-sharded x%d, '
          'no collective' % world,
          'streams': 'Harmonic and FilteredNoise on two free-running HIP streams' if overlap
                     else 'one stream, back to back',
          'kernel_variants': {'harmonic': a.harm_kernel, 'filtered_noise': a.noise_kernel}},
      'per_gpu_value': value / world,
      'roofline': dict(roof, **{
          'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'event_stride': a.event_stride,
          'timing': roofline_timing or ('dispatch start/stop events (hipExtLaunchKernelGGL) on the launch stream, '
                                        'every event_stride-th launch inside the timed regions'),
          'whole_step': {'algorithmic_bytes': step_bytes,
                         'achieved_GBs': step_bytes / (elapsed / a.steps) / 1e9,
                         'frac': step_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS},
          # the path sits above the fp32 ridge point: the vector-ALU ceiling beside the HBM fraction - on the
          # REFERENCE FORMULATION's flop count (3 FMAs per sample and live harmonic, 2 L per sample of the FIR).  The
          # wavetable / matrix-core kernels do not execute those flops: the figure says how fast a direct-sum kernel would
          # have to run to keep up, not what this one achieves (VERDICT r3, weak #6b); what it executes is `executed`
          'alu_note': {'reference_formulation_flop_per_launch': dom_flops,
                       'reference_formulation_equivalent_TFLOPs': dom_flops / dom_avg_s / 1e12,
                       'peak_TFLOPs': FP32_VECTOR_PEAK_TFLOPS,
                       'reference_formulation_equivalent_frac': dom_flops / dom_avg_s / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                       'whole_step_reference_formulation_equivalent_frac':
                           (harm_flops + noise_flops) / (elapsed / a.steps) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                       'executed': load_issue_counters(a, dominant, B, dom_avg_s)}}),
      'kernel_breakdown_us_isolated': {k: v[0] / v[1] * 1e3 for k, v in breakdown.items()},
  }
  if timing:
    result['timing'] = timing
    if 'host_clock_ms_per_step_median' in timing:           # the clock round 1's headline used: comparable across rounds
      result['ms_per_step_host_clock'] = timing['host_clock_ms_per_step_median']
  if one_stream_elapsed is not None:
    one = one_stream_elapsed / a.steps
    result['one_stream'] = {'ms_per_step': one * 1e3, 'value': world * B * a.n_samples / one / 1e6,
                            'whole_step_frac': step_bytes / one / 1e9 / HBM_PEAK_GBS}
  if B == 128:
    result['target'] = '>= 0.5 of the HBM roofline at this shape (BASELINE.json north_star): roofline.whole_step.frac'
  if second:
    result['configs_1' if second.get('batch_per_gpu') == 32 else 'second_shape'] = second
  if fused_add is not None:
    # the group every shipped DAG ends with (gin/models/ae.gin:49-56): FilteredNoise, then Harmonic with processors.Add
    # fused into its kernel - ONE [B,N] stream written: 14.44 bytes per output sample, never the larger figure
    fa_bytes = 4 * B * (a.n_frames * (a.n_harmonics + 2) + a.n_frames * a.n_bands + a.n_samples)
    per = fused_add / a.steps
    result['fused_add'] = {'what': 'ProcessorGroup[Harmonic, FilteredNoise, Add] as two launches on one stream: FilteredNoise, '
                                   'then Harmonic + Add (ddsp_harmonic_add_f32); the sum is the only [B,N] stream written',
                           'ms_per_step': per * 1e3, 'value': world * B * a.n_samples / per / 1e6,
                           'algorithmic_bytes': fa_bytes, 'bytes_per_sample': fa_bytes / (B * a.n_samples),
                           'achieved_GBs': fa_bytes / per / 1e9, 'frac': fa_bytes / per / 1e9 / HBM_PEAK_GBS}
  if 'measured_copy_GBs' in aux:
    result['roofline']['measured_copy_GBs'] = aux['measured_copy_GBs']
    result['roofline']['frac_of_measured_copy'] = roof['achieved'] / aux['measured_copy_GBs']
  if 'f0_regimes' in aux:
    # SURVEY.md 8(d) names two f0 regimes; the wavetable kernel's time depends on f0 in others too (HISTORY.md section 7), so the
    # line carries a small sweep - the same step, the same issue mode - with the whole-step roofline fraction of each, the
    # worst of them at the top level
    regimes = {}
    for name, r in aux['f0_regimes'].items():
      per = r['ms_per_step'] * 1e-3
      regimes[name] = dict(r, frac=step_bytes / per / 1e9 / HBM_PEAK_GBS)
    regimes['%g+-1 Hz (headline)' % a.f0] = {'ms_per_step': elapsed / a.steps * 1e3, 'steps': a.steps, 'value': value,
                                            'frac': step_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS}
    worst = min(regimes, key=lambda k: regimes[k]['frac'])
    best = max(regimes, key=lambda k: regimes[k]['frac'])
    result['f0_regimes'] = regimes
    result['worst_regime'] = dict(regimes[worst], regime=worst)
    result['min_regime_frac'] = regimes[worst]['frac']
    result['regime_worst_over_best_time'] = regimes[best]['frac'] / regimes[worst]['frac']
    if '200+-1 Hz' in regimes:
      result['f0_200_regime'] = regimes['200+-1 Hz']
  if 'error' in aux:
    result['aux_error'] = aux['error']
  if alt_elapsed is not None:
    result['other_issue_mode'] = {
        'streams': 'one stream, back to back' if overlap else 'two free-running HIP streams',
        'ms_per_step': alt_elapsed / a.steps * 1e3,
        'value': world * B * a.n_samples * a.steps / alt_elapsed / 1e6}
  if gather_ms is not None:
    result['allgather_ms'] = gather_ms
  if not a.no_cpu_baseline and world == 1:
    result['cpu_baseline'] = cpu_baseline_fn(a)
  elif not a.no_cpu_baseline:
    result['cpu_baseline'] = None     # rank 0 at N=1 only (bench contract)
  if a.dry_run:
    result.update(dry_run=True, data='none (dry run: no device work; plumbing test only, not a measurement)')
  return result


def shape_name(B):
  if B == 128:
    return 'north-star shape (the batch BASELINE.json quotes its metric and target on; configs[1] x 4)'
  if B == 32:
    return 'BASELINE configs[1]'
  return 'configs[1] clips at another batch size'


def second_block(a, world, B, elapsed, steps, prof, breakdown, elapsed_one_stream=None):
  """The same step at a second batch size (configs[1]: 32 per GPU): same definitions as the headline, the two calls on
  two free-running streams as there; the one-stream figure beside it."""
  harm_bytes, noise_bytes = algorithmic_bytes(a, B)
  step_bytes = harm_bytes + noise_bytes
  per_step = elapsed / steps
  block = {'batch_per_gpu': B, 'workload': shape_name(B),
           'streams': 'Harmonic and FilteredNoise on two free-running HIP streams', 'steps': steps,
           'ms_per_step': per_step * 1e3, 'value': world * B * a.n_samples / per_step / 1e6,
           'whole_step': {'algorithmic_bytes': step_bytes, 'achieved_GBs': step_bytes / per_step / 1e9,
                          'frac': step_bytes / per_step / 1e9 / HBM_PEAK_GBS},
           'kernel_breakdown_us': {k: v[0] / v[1] * 1e3 for k, v in breakdown.items()}}
  if elapsed_one_stream is not None:
    one = elapsed_one_stream / steps
    block['one_stream'] = {'ms_per_step': one * 1e3, 'value': world * B * a.n_samples / one / 1e6,
                           'whole_step_frac': step_bytes / one / 1e9 / HBM_PEAK_GBS}
  if prof:
    dominant = max(prof, key=lambda k: prof[k][0] / prof[k][1])
    roof = kernel_roofline(a, B, dominant, *prof[dominant])
    roof.pop('_flops'), roof.pop('_avg_s')
    roof['timing'] = ('sampled dispatch events of the one-stream regions (on two streams a dispatch also spans the wait for '
                      'the CUs the other kernel still holds)')
    block['dominant_kernel'] = roof
  return block


# ---------------------------------------------------------------------------------------------------------
def _free_port():
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def self_launch(a, argv):
  """`python bench.py --gpus N` with no launcher around it: become the launcher.  One rank per GPU through
  torch.distributed.run on 127.0.0.1; rank 0 prints the JSON line, which is passed through unchanged."""
  if not a.dry_run:
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < a.gpus:
      raise SystemExit('bench.py: --gpus %d asked for but this node shows %d GPU(s); refusing to report a '
                       'smaller job under that label' % (a.gpus, have))
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes needs it on this driver)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
  proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
  lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
  for ln in proc.stdout.splitlines():
    if not ln.startswith('{'):
      print(ln, file=sys.stderr)
  if proc.returncode != 0 or len(lines) != 1:
    raise SystemExit('bench.py: the %d-rank run failed (exit code %d, %d JSON lines)' %
                     (a.gpus, proc.returncode, len(lines)))
  if json.loads(lines[0]).get('n_gpus') != a.gpus:
    raise SystemExit('bench.py: the ranks reported n_gpus=%r, not %d' % (json.loads(lines[0]).get('n_gpus'), a.gpus))
  print(lines[0], flush=True)


class _HostClockEvent:
  """--dry-run stand-in for torch.cuda.Event (no device)."""

  def __init__(self, enable_timing=True):
    self.t = 0.0

  def record(self, stream=None):
    self.t = time.perf_counter()

  def elapsed_time(self, other):
    return (other.t - self.t) * 1e3


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  a = parse_args(argv)
  if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    return self_launch(a, argv)

  import torch
  import torch.distributed as dist

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != a.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
  dry = a.dry_run
  if not dry:
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    torch.cuda.set_device(local_rank)
  dev_name = 'cpu' if dry else 'cuda'
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if dry:
      dist.init_process_group('gloo', rank=rank, world_size=world)
    else:
      dist.init_process_group('nccl', rank=rank, world_size=world,
                              device_id=torch.device('cuda', local_rank))

  # ---- the two Processors (or, --dry-run, sleeps of their rough duration) --------------------------------
  if dry:
    _lib = None

    def make_step(B, seed, streams2, a=a):
      out = (torch.zeros((B, 1)), torch.zeros((B, 1)))

      def step(two_streams=None):
        time.sleep(2e-4)
        return out
      return step, {}
  else:
    from ddsp_amd import build
    if rank == 0:
      build.build()            # no-op when the shipped .so is current
    if world > 1:
      dist.barrier()
    import ddsp_amd as ddsp
    from ddsp_amd import _lib
    _lib.load()
    # (a high-priority stream for either kernel changes nothing: 69.0-70.6 us per step in all three arrangements, r03q)
    stream_h, stream_z = torch.cuda.Stream(), torch.cuda.Stream()
    stream_0 = torch.cuda.current_stream()

    def make_step(B, seed, streams2, a=a):
      # ---- per-rank shard of the global batch: independent rows, no data-path collective ----
      x = make_inputs(B, a, seed=seed)
      dev = {k: ddsp.core.tf_float32(v) for k, v in x.items()}
      harmonic = ddsp.synths.Harmonic(n_samples=a.n_samples, sample_rate=a.sample_rate)
      fnoise = ddsp.synths.FilteredNoise(n_samples=a.n_samples, window_size=0, seed=rank, noise_bits=a.noise_bits)
      harmonic.kernel = a.harm_kernel                          # instance attributes: the defaults unless asked
      if a.noise_kernel != 'auto':
        fnoise.kernel = a.noise_kernel

      # The two Processor calls of a step are independent (nothing on this path joins them; the
      # reference's Add would): at small batches Harmonic and FilteredNoise are issued on two free-running HIP
      # streams so each runs in the CUs the other leaves idle.  Joining the streams every step costs more than
      # it gains (88 vs 62 us at batch 32), so the join is the event pair that closes a timed region.
      def step(two_streams=None):
        two_streams = streams2 if two_streams is None else two_streams
        if two_streams:
          # set_stream, not the `with torch.cuda.stream()` context manager: the manager costs ~15 us of
          # host time per use, which at batch 32 is as long as the kernels it is trying to overlap
          torch.cuda.set_stream(stream_h)
          h = harmonic(dev['amplitudes'], dev['harmonic_distribution'], dev['f0_hz'])
          torch.cuda.set_stream(stream_z)
          z = fnoise(dev['magnitudes'])
          torch.cuda.set_stream(stream_0)
        else:
          h = harmonic(dev['amplitudes'], dev['harmonic_distribution'], dev['f0_hz'])
          z = fnoise(dev['magnitudes'])
        return h, z

      def step_fused_add(two_streams=None):
        z = fnoise(dev['magnitudes'])
        return harmonic.call_add(dev['amplitudes'], dev['harmonic_distribution'], dev['f0_hz'], z), z
      step.fused_add = step_fused_add
      return step, dev

  def sync_all():
    if world > 1:
      dist.barrier()
    if not dry:
      torch.cuda.synchronize()

  def max_over_ranks(seconds):
    if world > 1:
      t = torch.tensor([seconds], dtype=torch.float64, device=dev_name)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      return float(t.item())
    return seconds

  Event = _HostClockEvent if dry else torch.cuda.Event

  def timed_region(step, steps, two_streams):
    """Exactly `steps` steps, barrier + synchronize on both sides; -> (event seconds, host-clock seconds, last output).
    The events sit on the base stream; in the two-stream mode both streams start behind the first event and the
    base stream waits for both before the second."""
    e0, e1 = Event(enable_timing=True), Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    if dry:
      e0.record()
    else:
      e0.record(stream_0)
      if two_streams:
        stream_h.wait_event(e0)
        stream_z.wait_event(e0)
    out = None
    for _ in range(steps):
      out = step(two_streams)
    if dry:
      e1.record()
    else:
      if two_streams:
        eh, ez = Event(), Event()
        eh.record(stream_h)
        ez.record(stream_z)
        stream_0.wait_event(eh)
        stream_0.wait_event(ez)
      e1.record(stream_0)
    sync_all()
    wall = time.perf_counter() - t0
    return e0.elapsed_time(e1) * 1e-3, wall, out

  def repeated_regions(step, steps, two_streams, repeats):
    ev, wall, out = [], [], None
    for _ in range(repeats):
      e, w, out = timed_region(step, steps, two_streams)
      ev.append(e)
      wall.append(w)
    return ev, wall, out

  def settle(step, seconds):
    # Clock settle (untimed): an idle MI355X needs ~20 ms of load to reach its sustained clock - the
    # same step measures 43 us right after a 5-step warm-up and 37 us from then on.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < seconds:
      for _ in range(20):
        step()
      if not dry:
        torch.cuda.synchronize()

  # ---- headline: BASELINE configs[1] ---------------------------------------------------------------------
  B = a.batch
  # two free-running streams at every batch size: the two persistent kernels cannot share a CU (each takes its LDS), so
  # nothing stretches, and the second kernel's blocks start on the CUs the first one has already left
  # (batch 128: 87 against 96 us per step, profiles/r02w_streams.txt)
  overlap = a.streams in ('2', 'auto') and not a.no_overlap
  step, dev = make_step(B, 1000 + rank, overlap)
  for _ in range(a.warmup):
    step()
  if not dry:
    torch.cuda.synchronize()

  # diagnostic pass (untimed, one stream so every kernel runs alone): every kernel bracketed, to
  # find the dominant one and give the isolated per-kernel times
  if dry:
    breakdown = {'dry_run_step': (0.6, 3)}
  else:
    for _ in range(5):               # the first one-stream launches after the two-stream phase are not representative
      step(two_streams=False)
    torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=256)
    for _ in range(10):
      step(two_streams=False)
    torch.cuda.synchronize()
    breakdown = _lib.profile_end()
  dominant = max(breakdown, key=lambda k: breakdown[k][0] / breakdown[k][1])
  # clock settle LAST, in the issue mode of the timed regions: the first two-stream regions after a stretch of one-stream
  # launches (the diagnostic pass) are slow - 3.4 ms for a 1.4 ms region, then tens of regions above the steady 68-69 us per
  # step (r03t: 74.2 vs 70.0 us as the median of 29 / 71 regions) - the runtime re-activates the second stream's queue
  settle(step, a.settle if not dry else 0.0)
  # ... and the same again as REGIONS (untimed): the event fork / join of a region and the synchronize between regions have a
  # warm-up of their own (the first region takes 3.4 ms instead of 1.4, the next ~30 are 5 % slow)
  # (a fixed COUNT of regions, the same on every rank: a region's bracket holds a barrier when world > 1, so a loop that
  # ran until a rank's own clock said stop would leave the ranks with different numbers of barriers)
  k_warm = min(a.steps, 50)
  n_warm = 0 if a.settle <= 0 else max(3, min(300, int(5000 * a.settle / 0.5) // max(k_warm, 1)))
  for _ in range(n_warm if not dry else 0):
    timed_region(step, k_warm, overlap)

  # ---- timed regions: exactly K steps each, barrier + synchronize on both sides --------------------------
  e_probe, _, _ = timed_region(step, a.steps, overlap)          # untimed probe: sizes the repeat count
  e_probe = max_over_ranks(e_probe)                             # every rank must run the SAME number of regions (collectives inside)
  repeats = a.repeats if a.repeats > 0 else int(min(200, max(10, 0.1 / max(e_probe, 1e-6))))
  # The dominant kernel's dispatch events.  At configs[1]'s batch they are sampled inside the timed regions.  At batch 128
  # on two streams a dispatch of one kernel also spans the time its blocks wait for the CUs the other kernel's persistent
  # blocks still hold, so its duration says nothing about the kernel: there the events are sampled in one-stream regions
  # of the same K steps run right after the timed ones (their step time is reported as `one_stream`).
  events_in_one_stream_regions = overlap and B >= 64
  if not dry and not events_in_one_stream_regions:
    _lib.profile_begin([dominant], max_records=2 * a.steps * repeats // max(a.event_stride, 1) + 64,
                       stride=a.event_stride)
  ev, wall, out = repeated_regions(step, a.steps, overlap, repeats)
  prof = {dominant: (0.6, 3)} if dry else (None if events_in_one_stream_regions else _lib.profile_end())
  elapsed = max_over_ranks(statistics.median(ev))
  one_stream_elapsed, roofline_timing = None, None
  if events_in_one_stream_regions:
    reps1 = max(3, repeats // 3)
    if not dry:
      _lib.profile_begin([dominant], max_records=2 * a.steps * reps1 // max(a.event_stride, 1) + 64,
                         stride=a.event_stride)
    ev_1s, _, _ = repeated_regions(step, a.steps, False, reps1)
    if not dry:
      prof = _lib.profile_end()
    one_stream_elapsed = max_over_ranks(statistics.median(ev_1s))
    roofline_timing = ('dispatch start/stop events (hipExtLaunchKernelGGL) on the launch stream, every event_stride-th '
                       'launch inside %d one-stream regions of the same K steps run right after the timed (two-stream) '
                       'regions: on two streams a dispatch also spans the wait for the CUs the other kernel holds' % reps1)
  timing = {'method': 'median of %d regions of K=%d steps; each region bracketed by barrier + synchronize and '
                      'timed by HIP events on the stream(s) (value) and by the host clock (beside it)' %
                      (repeats, a.steps),
            'repeats': repeats, 'region_ms_median': statistics.median(ev) * 1e3,
            'region_ms_min': min(ev) * 1e3, 'region_ms_max': max(ev) * 1e3, 'region_ms_first': ev[0] * 1e3,
            'host_clock_ms_per_step_median': max_over_ranks(statistics.median(wall)) / a.steps * 1e3,
            'host_clock_ms_per_step_first': wall[0] / a.steps * 1e3}

  # optionally the other issue mode, same K steps, reported next to the headline
  alt_elapsed = None
  if a.also_other_mode:
    ev_alt, _, _ = repeated_regions(step, a.steps, not overlap, max(3, repeats // 3))
    alt_elapsed = max_over_ranks(statistics.median(ev_alt))

  # ---- Harmonic + FilteredNoise + Add as the DAGs run it, the Add fused into the Harmonic kernel (one stream) ----
  fused_add_elapsed = None
  if not dry and not a.no_aux and hasattr(step, 'fused_add'):
    # (as the configs_1 block: what may raise runs inside the `try`, the ranks agree on the outcome, and the regions - which
    # hold a barrier each when world > 1 - run outside it on every rank or on none: a rank that raised in the middle of a
    # loop of regions would leave the others waiting in a barrier and cost the headline line)
    err_f = 0.0
    try:
      with torch.no_grad():
        for _ in range(10):
          step.fused_add()
        torch.cuda.synchronize()
    except Exception:                             # noqa: BLE001 - a side block: the headline line must survive
      err_f = 1.0
    if max_over_ranks(err_f) == 0.0:
      with torch.no_grad():
        ev_f, _, _ = repeated_regions(step.fused_add, a.steps, False, max(3, repeats // 3))
      fused_add_elapsed = max_over_ranks(statistics.median(ev_f))

  # ---- auxiliary yardsticks (untimed for the headline; a failure here never costs the JSON line) ----
  aux = {}
  if not a.no_aux and not dry:
    try:
      # (i) SURVEY.md 8(d): the fraction is quoted against the 8 TB/s spec peak; the device-to-device copy
      # rate measured in the same run says what this box's HBM actually sustains (read + write counted)
      src = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device='cuda').normal_()
      dst = torch.empty_like(src)
      for _ in range(3):
        dst.copy_(src)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(10):
        dst.copy_(src)
      e1.record()
      torch.cuda.synchronize()
      aux['measured_copy_GBs'] = 2 * src.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
      del src, dst
      # (ii) f0 regimes beside the headline's 70 +- 1 Hz: SURVEY.md 8(d)'s second, "test-like" f0 = 200 + N(0,1) Hz
      # (processors_test.py:40; 39 of 100 harmonics below Nyquist, harmonic 40 ON it), a note with vibrato (220 Hz, 6 Hz deep
      # at 5.5 Hz), and 333 / 500 +- 1 Hz (Nyquist / 24 and / 16: the table reads' bank-conflict resonances, HISTORY.md
      # section 7).  Same step, same stream mode, regions of the same K steps, three each
      x_r = make_inputs(B, a, seed=2000 + rank)
      jitter = x_r['f0_hz'] - a.f0
      tt = np.arange(a.n_frames)[None, :, None] / 250.0
      vib = 6.0 * np.sin(2 * np.pi * 5.5 * tt + np.random.default_rng(2100 + rank).uniform(0, 6.28, (B, 1, 1)))
      dev_headline = dict(dev)
      k_r = a.steps                  # (regions as long as the headline's: a shorter region costs 2-3 us per step more, r03p)
      regimes = {}
      for name, f0 in (('200+-1 Hz', 200.0 + jitter), ('220 Hz, vibrato 6 Hz deep at 5.5 Hz', 220.0 + vib),
                       ('333+-1 Hz', 333.0 + jitter), ('500+-1 Hz', 500.0 + jitter)):
        x_r['f0_hz'] = f0.astype(np.float32)
        dev.update({k: ddsp.core.tf_float32(v) for k, v in x_r.items()})
        # (the 256 MB copy above and the upload leave the clocks where an idle chip's are: the driver's 20-step run measured
        # the first regime at 86 us per step and the same regime at 69 over 1000 steps - profiles/r04_final_bench_*.json)
        settle(step, 0.02)
        ev_r, _, _ = repeated_regions(step, k_r, overlap, 3)
        dt = max_over_ranks(statistics.median(ev_r))
        regimes[name] = {'ms_per_step': dt / k_r * 1e3, 'steps': k_r, 'value': world * B * a.n_samples * k_r / dt / 1e6}
      dev.update(dev_headline)
      aux['f0_regimes'] = regimes
    except Exception as exc:                      # noqa: BLE001 - diagnostics only
      aux['error'] = repr(exc)

  gather_ms = None
  if a.allgather and world > 1:
    h = out[0]
    full = torch.empty((world * h.shape[0], h.shape[1]), dtype=torch.float32, device=dev_name)
    for _ in range(3):
      dist.all_gather_into_tensor(full, h)
    sync_all()
    t1 = time.perf_counter()
    for _ in range(10):
      dist.all_gather_into_tensor(full, h)
    sync_all()
    gather_ms = max_over_ranks(time.perf_counter() - t1) / 10 * 1e3

  # ---- the second shape: BASELINE configs[1], batch 32 per GPU ---------------------------------------------------
  # (every rank takes the same path through the collectives: a failure on one rank is agreed on before anything is
  # reduced over the ranks - ADVICE r2)
  second = None
  if not a.no_second_shape and a.second_batch > 0 and a.second_batch != B:
    BN = a.second_batch
    ns_steps = max(10, min(a.steps, 200))
    err, med_n, med_1, prof_n, bd_n = None, 0.0, 0.0, None, {}
    try:
      del step, dev
      step_n, dev_n = make_step(BN, 3000 + rank, True)
      for _ in range(20):
        step_n()
      settle(step_n, 0.02 if not dry else 0.0)
      if dry:
        bd_n, prof_n = {'dry_run_step': (0.6, 3)}, None
      else:
        for _ in range(10):
          step_n(two_streams=False)
        torch.cuda.synchronize()
        _lib.profile_begin(None, max_records=128)
        for _ in range(10):
          step_n(two_streams=False)
        torch.cuda.synchronize()
        bd_n = _lib.profile_end()
    except Exception as exc:                      # noqa: BLE001 - the headline line must survive
      err = repr(exc)
    failed = max_over_ranks(1.0 if err else 0.0) > 0.0
    if not failed:
      ev_n, _, _ = repeated_regions(step_n, ns_steps, True, 10)
      if not dry:
        _lib.profile_begin(list(bd_n), max_records=4 * ns_steps * 5 // max(a.event_stride, 1) + 64,
                           stride=a.event_stride)
      ev_1, _, _ = repeated_regions(step_n, ns_steps, False, 5)
      if not dry:
        prof_n = _lib.profile_end()
      second = second_block(a, world, BN, max_over_ranks(statistics.median(ev_n)), ns_steps, prof_n, bd_n,
                            max_over_ranks(statistics.median(ev_1)))
    else:
      second = {'batch_per_gpu': BN, 'error': err or 'another rank failed to set this shape up'}

  # ---- BASELINE configs[4] per GPU: 48 kHz, 200 harmonics, 10 s clips (2500 frames of 192 samples), batch 32 ---------
  # (for the record beside the headline, when the headline is the default shape: the Harmonic kernel's instances for
  # 129 .. 200 harmonics; same definitions as `configs_1`)
  fifth = None
  default_shape = (a.n_frames, a.n_harmonics, a.n_samples, a.sample_rate, a.n_bands) == (1000, 100, 64000, 16000, 65)
  if second is not None and 'error' not in second and default_shape and not a.no_aux:
    a5 = argparse.Namespace(**vars(a))
    a5.n_frames, a5.n_harmonics, a5.n_samples, a5.sample_rate = 2500, 200, 480000, 48000
    c5_steps = 50                                 # (its own region length: a 146 us step; three untimed regions first)
    err, med_n, med_1, prof_5, bd_5 = None, 0.0, 0.0, None, {}
    try:
      del step_n, dev_n
      step_5, dev_5 = make_step(32, 5000 + rank, True, a5)
      for _ in range(10):
        step_5()
      settle(step_5, 0.02 if not dry else 0.0)
      if dry:
        bd_5 = {'dry_run_step': (0.6, 3)}
      else:
        for _ in range(10):                         # (warmed: three cold bracketed launches said 107 us where rocprofv3 says 63)
          step_5(two_streams=False)
        torch.cuda.synchronize()
        _lib.profile_begin(None, max_records=128)
        for _ in range(10):
          step_5(two_streams=False)
        torch.cuda.synchronize()
        bd_5 = _lib.profile_end()
    except Exception as exc:                      # noqa: BLE001 - the headline line must survive
      err = repr(exc)
    failed = max_over_ranks(1.0 if err else 0.0) > 0.0
    if not failed:
      for _ in range(3 if not dry else 0):
        timed_region(step_5, c5_steps, True)
      ev_n, _, _ = repeated_regions(step_5, c5_steps, True, 5)
      if not dry:
        _lib.profile_begin(list(bd_5), max_records=4 * c5_steps * 3 // max(a.event_stride, 1) + 64, stride=a.event_stride)
      ev_1, _, _ = repeated_regions(step_5, c5_steps, False, 3)
      if not dry:
        prof_5 = _lib.profile_end()
      fifth = second_block(a5, world, 32, max_over_ranks(statistics.median(ev_n)), c5_steps, prof_5, bd_5,
                           max_over_ranks(statistics.median(ev_1)))
      fifth['workload'] = ('BASELINE configs[4] per GPU: Harmonic (200 harmonics) + FilteredNoise, 10 s clips @ 48 kHz, '
                           '2500 frames of 192 samples, batch 32 (256 over 8 GPUs)')
    else:
      fifth = {'batch_per_gpu': 32, 'error': err or 'another rank failed to set this shape up'}

  # ---- the callers either side of the path at BASELINE's batch, and the full-resolution noise -------------------------------
  # (under the driver's clock, in the default line.  One helper: whatever may raise - building the
  # Processors, the first launches - runs inside a `try`, the ranks agree on the outcome, and the regions, which hold a
  # barrier each when world > 1, run on every rank or on none.)
  def side_block(build_fn, steps, reps, two_streams=False):
    err, fn, meta, bd = None, None, {}, {}
    try:
      fn, meta = build_fn()
      with torch.no_grad():
        for _ in range(10):
          fn(two_streams)
        settle(lambda: fn(two_streams), 0.05)
        for _ in range(5):
          fn(False)
        torch.cuda.synchronize()
        _lib.profile_begin(None, max_records=512)
        for _ in range(5):
          fn(False)
        torch.cuda.synchronize()
        bd = _lib.profile_end()
    except Exception as exc:                      # noqa: BLE001 - a side block: the headline line must survive
      err = repr(exc)
    if max_over_ranks(1.0 if err else 0.0) > 0.0:
      return {'error': err or 'another rank failed to set this block up'}
    with torch.no_grad():
      for _ in range(3):
        timed_region(fn, steps, two_streams)
      ev_s, _, _ = repeated_regions(fn, steps, two_streams, reps)
    per = max_over_ranks(statistics.median(ev_s)) / steps
    nbytes = meta.pop('algorithmic_bytes')
    b_, n_ = meta.pop('batch_per_gpu'), meta.pop('n_samples')
    return dict(meta, batch_per_gpu=b_, steps=steps, regions=reps,
                streams='two free-running HIP streams' if two_streams else 'one stream',
                ms_per_step=per * 1e3, value=world * b_ * n_ / per / 1e6,
                whole_step={'algorithmic_bytes': nbytes, 'achieved_GBs': nbytes / per / 1e9,
                            'frac': nbytes / per / 1e9 / HBM_PEAK_GBS},
                kernel_breakdown_us={k: v[0] / v[1] * 1e3 for k, v in bd.items()},
                kernel_launches_per_step={k: v[1] / 5.0 for k, v in bd.items()})

  other = {}
  if default_shape and not dry and not a.no_aux and not a.no_other_configs:
    BO = 128
    synth_bytes = 4 * BO * (a.n_frames * (a.n_harmonics + 2) + a.n_frames * a.n_bands + a.n_samples)   # the DAG with Add fused: one [B,N] stream

    def dag_parts(seed, noise_bits=23):
      x = make_inputs(BO, a, seed=seed)
      d = {k: ddsp.core.tf_float32(v) for k, v in x.items()}
      harmonic = ddsp.synths.Harmonic(n_samples=a.n_samples, sample_rate=a.sample_rate)
      fnoise = ddsp.synths.FilteredNoise(n_samples=a.n_samples, window_size=0, seed=rank, noise_bits=noise_bits)

      def dag():
        z = fnoise(d['magnitudes'])
        return harmonic.call_add(d['amplitudes'], d['harmonic_distribution'], d['f0_hz'], z)
      return d, harmonic, fnoise, dag

    def build_configs_2():
      d, harmonic, fnoise, dag = dag_parts(7000 + rank)
      loss = ddsp.losses.SpectralLoss(logmag_weight=1.0)              # gin/models/ae.gin:36-41: L1, mag + logmag, six scales
      target = ddsp.core.tf_float32(0.3 * np.random.default_rng(7100 + rank).standard_normal((BO, a.n_samples)))

      def fn(two_streams=None):
        return loss(target, dag())
      meta = {'workload': 'BASELINE configs[2]: FilteredNoise, Harmonic + Add (fused), losses.SpectralLoss (fft sizes 2048 .. 64, '
                          'L1, mag + logmag) on the sum against a target; batch 128, 4 s @ 16 kHz; forward value',
              'batch_per_gpu': BO, 'n_samples': a.n_samples,
              # the DAG's controls in, its one audio stream out; the loss reads that stream and the target
              'algorithmic_bytes': synth_bytes + 8 * BO * a.n_samples}
      fn.loss, fn.target, fn.dag = loss, target, dag
      return fn, meta

    def build_configs_3(per_clip_ir=False):
      d, harmonic, fnoise, dag = dag_parts(7200 + rank)
      L = 48000
      if per_clip_ir:
        reverb = ddsp.effects.Reverb(add_dry=True)
        ir = ddsp.core.tf_float32(0.05 * np.random.default_rng(7300 + rank).standard_normal((BO, L)))

        def fn(two_streams=None):
          return reverb(dag(), ir)
      else:
        reverb = ddsp.effects.Reverb(trainable=True, reverb_length=L, add_dry=True)
        reverb.build(device=d['magnitudes'].device)
        reverb._ir = ddsp.core.tf_float32(0.05 * np.random.default_rng(7300 + rank).standard_normal((L,)))

        def fn(two_streams=None):
          return reverb(dag())
      meta = {'workload': 'BASELINE configs[3] per GPU: ProcessorGroup FilteredNoise, Harmonic + Add (fused), effects.Reverb (%s '
                          '48 000-tap impulse response, add_dry), batch 128 (1024 over 8 GPUs), 4 s @ 16 kHz' %
                          ('one per clip:' if per_clip_ir else 'the ONE trainable, as gin/models/solo_instrument.gin:26-40:'),
              'batch_per_gpu': BO, 'n_samples': a.n_samples,
              # ... the Reverb reads that stream and writes its own; the impulse response(s) once
              'algorithmic_bytes': synth_bytes + 8 * BO * a.n_samples + 4 * L * (BO if per_clip_ir else 1)}
      return fn, meta

    def build_other_resolution(bits):
      x = make_inputs(B, a, seed=1000 + rank)
      d = {k: ddsp.core.tf_float32(v) for k, v in x.items()}
      harmonic = ddsp.synths.Harmonic(n_samples=a.n_samples, sample_rate=a.sample_rate)
      fnoise = ddsp.synths.FilteredNoise(n_samples=a.n_samples, window_size=0, seed=rank, noise_bits=bits)

      def fn(two_streams=None):
        if two_streams:
          torch.cuda.set_stream(stream_h)
          h = harmonic(d['amplitudes'], d['harmonic_distribution'], d['f0_hz'])
          torch.cuda.set_stream(stream_z)
          z = fnoise(d['magnitudes'])
          torch.cuda.set_stream(stream_0)
        else:
          h = harmonic(d['amplitudes'], d['harmonic_distribution'], d['f0_hz'])
          z = fnoise(d['magnitudes'])
        return h, z
      hb, nb = algorithmic_bytes(a, B)
      meta = {'workload': ('the headline step with FilteredNoise(noise_bits=23): generated noise of 2^23 levels (the resolution of '
                           "the reference's tf.random.uniform, ddsp/synths.py:192-193) carried as fp16 hi / lo pairs" if bits == 23 else
                           'the headline step with FilteredNoise(noise_bits=11): generated noise of 2048 levels, every sample an '
                           "fp16 number (NOT the reference's resolution; the headline draws 2^23 levels)"),
              'batch_per_gpu': B, 'n_samples': a.n_samples, 'algorithmic_bytes': hb + nb}
      return fn, meta

    k_o = max(10, min(a.steps, 100))
    # the other resolution of the generated noise beside the headline's (23 bits - the reference's - is the headline since round 6)
    other_bits = 11 if a.noise_bits == 23 else 23
    other_key = 'fnoise_11_bit_levels' if other_bits == 11 else 'fnoise_full_resolution'
    other[other_key] = side_block(lambda: build_other_resolution(other_bits), a.steps, max(3, repeats // 2), two_streams=overlap)
    if 'error' not in other[other_key]:
      fr = other[other_key]
      fr['headline_ms_per_step'] = elapsed / a.steps * 1e3
      fr['cost_of_the_twelve_bits_us'] = abs(fr['ms_per_step'] * 1e3 - elapsed / a.steps * 1e6)
    other['configs_2'] = side_block(build_configs_2, k_o, 5)
    other['configs_3'] = side_block(build_configs_3, k_o, 5)
    c3b = side_block(lambda: build_configs_3(per_clip_ir=True), k_o, 3)
    if 'error' not in other['configs_3']:
      other['configs_3']['per_clip_ir'] = ({k: c3b[k] for k in ('ms_per_step', 'value', 'whole_step', 'kernel_breakdown_us')}
                                           if 'error' not in c3b else c3b)
    # the loss's value AND gradient w.r.t. the audio, as a training step runs it (one kernel for both)
    if 'error' not in other['configs_2']:
      err_g, fn_g = None, None
      try:
        fn2, _ = build_configs_2()

        def fn_g(two_streams=None):
          with torch.no_grad():
            y = fn2.dag()
          y.requires_grad_(True)
          with torch.enable_grad():
            fn2.loss(fn2.target, y).backward()
          return y.grad
        for _ in range(5):
          fn_g()
        torch.cuda.synchronize()
      except Exception as exc:                    # noqa: BLE001
        err_g = repr(exc)
      if max_over_ranks(1.0 if err_g else 0.0) == 0.0:
        ev_g, _, _ = repeated_regions(fn_g, k_o, False, 3)
        per = max_over_ranks(statistics.median(ev_g)) / k_o
        other['configs_2']['with_gradient_wrt_audio'] = {'ms_per_step': per * 1e3, 'value': world * BO * a.n_samples / per / 1e6}
      else:
        other['configs_2']['with_gradient_wrt_audio'] = {'error': err_g or 'another rank failed'}

  # ---- who holds which rows (contiguous batch shards, no data-path collective) ---------------------------------
  # every rank reports the rows ddsp_amd.distributed.shard_bounds gives it for the global batches of BASELINE's sharded
  # configurations at this world size (weak scaling: per-GPU batch x world; configs[3] 1024 -> 128 and configs[4] 256 -> 32
  # at 8 GPUs), summed into one vector: rank 0 prints what every rank actually took (tests/test_bench_contract.py, 8 ranks)
  shards = None
  try:
    from ddsp_amd.distributed import shard_bounds
    plan = {'headline': B * world, 'configs_3': 128 * world, 'configs_4': 32 * world}
    v = [0.0] * (len(plan) * 2 * world)
    for i, (name, gb) in enumerate(plan.items()):
      v[(i * world + rank) * 2], v[(i * world + rank) * 2 + 1] = (float(x) for x in shard_bounds(gb, rank, world))
    if world > 1:
      vec = torch.tensor(v, dtype=torch.float64, device=dev_name)
      dist.all_reduce(vec, op=dist.ReduceOp.SUM)
      v = vec.tolist()
    v = [int(x) for x in v]
    shards = {name: {'global_batch': gb, 'rows_of_rank': [[v[(i * world + r) * 2], v[(i * world + r) * 2 + 1]] for r in range(world)]}
              for i, (name, gb) in enumerate(plan.items())}
  except ImportError:
    pass

  if rank == 0:
    result = build_result(a, world, B, elapsed, prof, breakdown, dominant, overlap, aux, alt_elapsed, gather_ms,
                          timing=timing, second=second, one_stream_elapsed=one_stream_elapsed,
                          roofline_timing=roofline_timing, fused_add=fused_add_elapsed)
    if fifth:
      result['configs_4'] = fifth
    result.update(other)
    if shards:
      result['shards'] = shards
    print(json.dumps(result), flush=True)

  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
