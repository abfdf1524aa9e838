#!/usr/bin/env python
"""bench.py - Msamples/s of the Harmonic + FilteredNoise hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N)

One "step" = one pass of the hot path over one batch of synthetic control tensors:
`synths.Harmonic(amplitudes, harmonic_distribution, f0_hz)` + `synths.FilteredNoise(magnitudes)`
(raw network outputs in, both get_controls prologues included, noise generated on chip),
each output sample counted once.  Inputs are resident in HBM before the timed region.
Workload (per GPU, weak scaling): BASELINE.json configs[1] - batch 32, 4 s @ 16 kHz,
F=1000 frames, K=100 harmonics (all below Nyquist: f0 = 70 + N(0,1) Hz), M=65 noise bands.

The JSON line also carries
  roofline     : dominant kernel, algorithmic bytes per launch / its mean duration measured
                 with HIP events on the launch stream during the timed region (ddsp_profile_*),
                 against the 8 TB/s HBM peak (the path is in fact ALU-bound).
  cpu_baseline : the numpy fp32 oracle ("port" of the TF op chain; TF itself cannot run
                 here) timed on this host's cores (concurrent worker processes) on a bounded
                 sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--warmup', type=int, default=500)   # ~20 ms of load: the GPU needs that long to reach its sustained clock (43 vs 37 us/step)
  ap.add_argument('--batch', type=int, default=32, help='clips per GPU (configs[1]: 32)')
  ap.add_argument('--n-frames', type=int, default=1000)
  ap.add_argument('--n-harmonics', type=int, default=100)
  ap.add_argument('--n-bands', type=int, default=65)
  ap.add_argument('--n-samples', type=int, default=64000)
  ap.add_argument('--sample-rate', type=int, default=16000)
  ap.add_argument('--f0', type=float, default=70.0, help='f0 centre in Hz (70: all harmonics live)')
  ap.add_argument('--cpu-clips', type=int, default=6, help='clips per worker process of the CPU oracle leg')
  ap.add_argument('--cpu-procs', type=int, default=0,
                  help='worker processes of the CPU oracle leg (0: the logical CPUs, at most 32)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--event-stride', type=int, default=8,
                  help='bracket every n-th launch of the dominant kernel with HIP events inside the '
                       'timed region (a bracketed launch costs ~5 us of queue time; 1 = all of them)')
  ap.add_argument('--streams', choices=['auto', '1', '2'], default='auto',
                  help='issue the two Processor calls on one stream or on two free-running streams; auto = two '
                       'when the per-GPU batch is below 64 (the kernels leave CUs idle for each other), one above')
  ap.add_argument('--no-overlap', action='store_true',
                  help='issue the two Processor calls back to back on one stream instead of on two '
                       'free-running HIP streams')
  ap.add_argument('--also-other-mode', action='store_true',
                  help='after the timed region, run K more steps in the other stream mode and report '
                       'them as other_issue_mode (off by default so that a rocprofv3 trace of the '
                       'default command sees one mode only)')
  ap.add_argument('--no-aux', action='store_true',
                  help='skip the auxiliary yardsticks after the timed region (measured device-copy bandwidth, '
                       'the f0 = 200 Hz regime of SURVEY.md 8d)')
  ap.add_argument('--harm-kernel', choices=['auto', 'direct', 'table_tphase'], default='auto',
                  help="Harmonic.kernel: 'auto' (default), 'direct' (sum harmonic by harmonic) or the experimental "
                       "'table_tphase' variant of the wavetable kernel")
  ap.add_argument('--noise-ir', choices=['vector', 'matrix', 'matrix_direct'], default='vector',
                  help="FilteredNoise.ir_design: 'vector' (default) or the experimental matrix-core designs")
  ap.add_argument('--allgather', action='store_true',
                  help='also time an RCCL all_gather of the audio (reported separately)')
  return ap.parse_args()


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


FP32_VECTOR_PEAK_TFLOPS = 157.3   # 256 CUs x 2.4 GHz x 256 flop/clk, packed fp32


def algorithmic_flops(a, batch):
  """SURVEY.md 8(d), the ALU note beside the HBM roofline: 3 FMAs per (sample, live harmonic) for the
  oscillator bank, 2 L flops per sample for the L-tap time-varying FIR (L = 2 (M - 1))."""
  live = min(a.n_harmonics, int((a.sample_rate / 2.0) / max(a.f0, 1e-6)))
  harm = batch * a.n_samples * 3 * live * 2
  noise = batch * a.n_samples * 2 * 2 * (a.n_bands - 1)
  return harm, noise


def cpu_baseline(a):
  """The oracle (numpy, fp32, op by op as TF executes it) on the host's cores: oracle/cpu_baseline.py runs
  `cpu_procs` concurrent worker processes of `cpu_clips` clips each (TF's CPU kernels are multi-threaded;
  `cores` = the workers actually used, the one-process figure is reported beside it)."""
  from oracle import cpu_baseline as leg
  procs = a.cpu_procs if a.cpu_procs > 0 else leg.default_procs()
  return leg.measure(a.cpu_clips, procs, a.n_frames, a.n_harmonics, a.n_bands, a.n_samples, a.sample_rate,
                     a.f0)


def build_result(a, world, B, elapsed, prof, breakdown, dominant, overlap, aux=None, alt_elapsed=None,
                 gather_ms=None, cpu_baseline_fn=None):
  """The JSON line of the bench contract from the measured quantities (pure: unit-tested on the CPU).

  elapsed: seconds for a.steps steps, max over ranks; prof / breakdown: {kernel: (total_ms, launches)} of the
  timed region's sampled dispatch events / of the untimed single-stream diagnostic pass."""
  aux = aux or {}
  cpu_baseline_fn = cpu_baseline_fn or cpu_baseline
  total_samples = world * B * a.n_samples * a.steps
  value = total_samples / elapsed / 1e6
  harm_bytes, noise_bytes = algorithmic_bytes(a, B)
  dom_ms, dom_n = prof[dominant]
  dom_avg_s = dom_ms / dom_n * 1e-3
  # algorithmic bytes of the launch = those of the Processor the kernel belongs to
  dom_bytes = harm_bytes if dominant.startswith('harm') else noise_bytes   # its Processor's bytes
  achieved = dom_bytes / dom_avg_s / 1e9
  traffic = None
  for tname in ('pmc_traffic_b%d.json' % B, 'pmc_traffic.json'):     # PMC passes are per batch size
    tpath = os.path.join(ROOT, 'profiles', tname)
    if traffic is None and os.path.exists(tpath):
      try:
        rec = json.load(open(tpath))
        if rec.get('batch') == B:
          traffic = rec.get('kernels', {}).get(dominant)
      except (ValueError, OSError):
        traffic = None
  step_bytes = harm_bytes + noise_bytes
  harm_flops, noise_flops = algorithmic_flops(a, B)
  dom_flops = harm_flops if dominant.startswith('harm') else noise_flops
  result = {
      'metric': 'Msamples/s (Harmonic+FilteredNoise, 16kHz, 100 harmonics)',
      'value': value, 'unit': 'Msamples/s', 'n_gpus': world, 'steps': a.steps,
      'warmup': a.warmup, 'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {
          'workload': 'BASELINE configs[1]: Harmonic+FilteredNoise, batch=%d per GPU, %d samples '
                      '@ %d Hz, %d frames, %d harmonics (f0=%g+N(0,1) Hz), %d noise bands, raw '
                      'controls in (get_controls fused), noise generated on chip' %
                      (B, a.n_samples, a.sample_rate, a.n_frames, a.n_harmonics, a.f0, a.n_bands),
          'batch_per_gpu': B, 'global_batch': world * B, 'parallelism': 'batch-sharded x%d, '
          'no collective' % world,
          'streams': 'Harmonic and FilteredNoise on two free-running HIP streams' if overlap
                     else 'one stream, back to back',
          'kernel_variants': {'harmonic': a.harm_kernel, 'noise_ir_design': a.noise_ir}},
      'roofline': {
          'bound': 'hbm', 'kernel': dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
          'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
          'algorithmic_bytes_per_launch': dom_bytes, 'avg_launch_us': dom_avg_s * 1e6,
          'launches': dom_n, 'event_stride': a.event_stride,
          'timing': 'dispatch start/stop events (hipExtLaunchKernelGGL) on the launch stream, '
                    'every event_stride-th launch inside the timed region',
          'whole_step': {'algorithmic_bytes': step_bytes,
                         'achieved_GBs': step_bytes / (elapsed / a.steps) / 1e9,
                         'frac': step_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS},
          # the path sits above the fp32 ridge point: the vector-ALU ceiling beside the
          # HBM fraction, on the reference formulation's flop count (the wavetable kernel does fewer)
          'alu_note': {'algorithmic_flop_per_launch': dom_flops,
                       'achieved_TFLOPs': dom_flops / dom_avg_s / 1e12,
                       'peak_TFLOPs': FP32_VECTOR_PEAK_TFLOPS,
                       'frac': dom_flops / dom_avg_s / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                       'whole_step_frac': (harm_flops + noise_flops) / (elapsed / a.steps) / 1e12 /
                                          FP32_VECTOR_PEAK_TFLOPS}},
      'kernel_breakdown_us_isolated': {k: v[0] / v[1] * 1e3 for k, v in breakdown.items()},
  }
  if 'measured_copy_GBs' in aux:
    result['roofline']['measured_copy_GBs'] = aux['measured_copy_GBs']
    result['roofline']['frac_of_measured_copy'] = achieved / aux['measured_copy_GBs']
  if 'f0_200_regime' in aux:
    result['f0_200_regime'] = aux['f0_200_regime']
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
  return result


def main():
  a = parse_args()
  import torch
  import torch.distributed as dist

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != a.gpus and world > 1:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
  assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
  torch.cuda.set_device(local_rank)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=torch.device('cuda', local_rank))

  from ddsp_amd import build
  if rank == 0:
    build.build()            # no-op when the shipped .so is current
  if world > 1:
    dist.barrier()
  import ddsp_amd as ddsp
  from ddsp_amd import _lib
  _lib.load()

  # ---- per-rank shard of the global batch: independent rows, no data-path collective ----
  B = a.batch
  x = make_inputs(B, a, seed=1000 + rank)
  dev = {k: ddsp.core.tf_float32(v) for k, v in x.items()}
  harmonic = ddsp.synths.Harmonic(n_samples=a.n_samples, sample_rate=a.sample_rate)
  fnoise = ddsp.synths.FilteredNoise(n_samples=a.n_samples, window_size=0, seed=rank)
  harmonic.kernel, fnoise.ir_design = a.harm_kernel, a.noise_ir          # instance attributes: the defaults unless asked

  # The two Processor calls of a step are independent (nothing on this path joins them; the
  # reference's Add would): Harmonic and FilteredNoise are issued on two free-running HIP streams so
  # FilteredNoise's latency-bound stages run under Harmonic's ALU-bound synthesis (the persistent
  # harmonic kernel hands its units out dynamically and uses whatever share of the CUs it gets).
  # Joining the streams every step costs more than it gains (88 vs 62 us at batch 32), so the join is
  # the barrier + synchronize that closes the timed region.  --no-overlap: one stream, back to back.
  stream_h, stream_z = torch.cuda.Stream(), torch.cuda.Stream()
  stream_0 = torch.cuda.current_stream()
  overlap = (a.streams == '2' or (a.streams == 'auto' and a.batch < 64)) and not a.no_overlap

  def step(two_streams=None):
    two_streams = overlap if two_streams is None else two_streams
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

  def sync_all():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(a.warmup):
    step()
  torch.cuda.synchronize()
  # Clock settle (untimed): an idle MI355X needs ~20 ms of load to reach its sustained clock - the
  # same step measures 43 us right after a 5-step warm-up and 37 us from then on.  Whatever W is,
  # keep the GPU busy for 50 ms before anything is measured.
  t_settle = time.perf_counter()
  while time.perf_counter() - t_settle < 0.05:
    for _ in range(20):
      step()
    torch.cuda.synchronize()

  # diagnostic pass (untimed, one stream so every kernel runs alone): every kernel bracketed, to
  # find the dominant one and give the isolated per-kernel times
  _lib.profile_begin(None, max_records=64)
  for _ in range(3):
    step(two_streams=False)
  torch.cuda.synchronize()
  breakdown = _lib.profile_end()
  dominant = max(breakdown, key=lambda k: breakdown[k][0] / breakdown[k][1])

  # ---- timed region: exactly K steps, barrier + synchronize on both sides ----------------
  _lib.profile_begin([dominant], max_records=2 * a.steps + 8, stride=a.event_stride)
  sync_all()
  t0 = time.perf_counter()
  for _ in range(a.steps):
    out = step()
  sync_all()
  elapsed = time.perf_counter() - t0
  prof = _lib.profile_end()

  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  # optionally the other issue mode, same K steps, reported next to the headline
  alt_elapsed = None
  if a.also_other_mode:
    sync_all()
    t_alt = time.perf_counter()
    for _ in range(a.steps):
      step(two_streams=not overlap)
    sync_all()
    alt_elapsed = time.perf_counter() - t_alt
    if world > 1:
      t = torch.tensor([alt_elapsed], dtype=torch.float64, device='cuda')
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      alt_elapsed = float(t.item())

  # ---- auxiliary yardsticks (untimed for the headline; a failure here never costs the JSON line) ----
  aux = {}
  if not a.no_aux:
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
      # (ii) SURVEY.md 8(d)'s second f0 regime: "test-like" f0 = 200 + N(0,1) Hz (processors_test.py:40),
      # 39 of 100 harmonics below Nyquist; same step, same stream mode, a fifth of the steps
      x200 = make_inputs(B, a, seed=2000 + rank)
      x200['f0_hz'] = (200.0 + (x200['f0_hz'] - a.f0)).astype(np.float32)
      dev200 = {k: ddsp.core.tf_float32(v) for k, v in x200.items()}
      dev_headline = dict(dev)
      k200 = max(a.steps // 5, 10)
      dev.update(dev200)
      for _ in range(20):
        step()
      sync_all()
      t200 = time.perf_counter()
      for _ in range(k200):
        step()
      sync_all()
      dt200 = time.perf_counter() - t200
      dev.update(dev_headline)
      if world > 1:
        t = torch.tensor([dt200], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt200 = float(t.item())
      aux['f0_200_regime'] = {'ms_per_step': dt200 / k200 * 1e3, 'steps': k200,
                              'value': world * B * a.n_samples * k200 / dt200 / 1e6}
    except Exception as exc:                      # noqa: BLE001 - diagnostics only
      aux['error'] = repr(exc)

  gather_ms = None
  if a.allgather and world > 1:
    h = out[0]
    full = torch.empty((world * B, a.n_samples), dtype=torch.float32, device='cuda')
    for _ in range(3):
      dist.all_gather_into_tensor(full, h)
    sync_all()
    t1 = time.perf_counter()
    for _ in range(10):
      dist.all_gather_into_tensor(full, h)
    sync_all()
    gather_ms = (time.perf_counter() - t1) / 10 * 1e3

  if rank == 0:
    result = build_result(a, world, B, elapsed, prof, breakdown, dominant, overlap, aux, alt_elapsed, gather_ms)
    print(json.dumps(result), flush=True)

  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
