"""The CPU leg of bench.py: the numpy fp32 restatement of the TF op chain (oracle/ddsp_oracle.py) timed on the
host's cores.  TEST / MEASUREMENT INFRASTRUCTURE - the product never imports this.

One clip = synths.Harmonic + synths.FilteredNoise of BASELINE.json configs[1] (4 s @ 16 kHz, 100 harmonics,
65 bands), op by op with the [N, K] tensors materialised as TensorFlow does.  TensorFlow's CPU kernels are
multi-threaded, so the port is run as P concurrent worker processes (plain child interpreters, one clip at a
time each) and `cores` reports P; the single-process figure is kept beside it."""
import os
import subprocess
import sys
import time

import numpy as np


def _make_inputs(n_clips, n_frames, n_harmonics, n_bands, f0, seed):
  rng = np.random.default_rng(seed)
  return dict(
      amplitudes=rng.standard_normal((n_clips, n_frames, 1)).astype(np.float32),
      harmonic_distribution=rng.standard_normal((n_clips, n_frames, n_harmonics)).astype(np.float32),
      f0_hz=(f0 + rng.standard_normal((n_clips, n_frames, 1))).astype(np.float32),
      magnitudes=rng.standard_normal((n_clips, n_frames, n_bands)).astype(np.float32))


def run_clips(n_clips, n_frames, n_harmonics, n_bands, n_samples, sample_rate, f0, seed):
  """n_clips clips through the oracle, one at a time; returns the seconds spent in the op chain."""
  from oracle import ddsp_oracle as O
  x = _make_inputs(n_clips, n_frames, n_harmonics, n_bands, f0, seed)
  rng = np.random.default_rng(seed + 1)
  t0 = time.perf_counter()
  for i in range(n_clips):
    s = slice(i, i + 1)
    O.harmonic(x['amplitudes'][s], x['harmonic_distribution'][s], x['f0_hz'][s], n_samples, sample_rate)
    noise = rng.uniform(-1, 1, (1, n_samples)).astype(np.float32)     # tf.random.uniform stand-in
    O.filtered_noise(x['magnitudes'][s], noise, 0)
  return time.perf_counter() - t0


def _physical_cores():
  """Physical cores this process may run on: its CPU affinity with SMT siblings counted once."""
  try:
    cpus = sorted(os.sched_getaffinity(0))
  except AttributeError:
    cpus = list(range(os.cpu_count() or 1))
  cores = set()
  for c in cpus:
    try:
      with open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c) as f:
        cores.add(f.read().strip())
    except OSError:
      cores.add(str(c))
  return max(1, len(cores))


def default_procs():
  """Worker processes: one per physical core this process may use (TF's own CPU kernels would use them all), bounded
  by memory - a clip's op chain holds ~0.5 GB of materialised [N, K] tensors, so at most half of MemAvailable is
  committed - and by 256."""
  procs = _physical_cores()
  try:
    with open('/proc/meminfo') as f:
      avail_kb = [int(line.split()[1]) for line in f if line.startswith('MemAvailable:')][0]
    procs = min(procs, max(1, int(avail_kb / 1024 / 1024 * 0.5 / 0.5)))
  except (OSError, IndexError, ValueError):
    procs = min(procs, 32)
  return max(1, min(256, procs))


def measure(clips_per_proc, procs, n_frames, n_harmonics, n_bands, n_samples, sample_rate, f0, timeout_s=180.0):
  """{'value', 'unit', 'cores', 'kind', 'sample', ...}: Msamples/s of `procs` concurrent workers.

  Workers are plain child interpreters (`python -m oracle.cpu_baseline ...`: no fork of a parent that holds an
  initialised HIP runtime, no re-import of the parent's main module); each runs one warm-up clip, then reports
  the seconds its own `clips_per_proc` clips took.  They do equal work and start together, so the throughput is
  all their clips over the slowest worker's time.  Any failure falls back to the single-process figure."""
  shape = (n_frames, n_harmonics, n_bands, n_samples, sample_rate, f0)
  run_clips(1, *shape, seed=99)                                           # warm-up: imports, page faults
  single_s = run_clips(2, *shape, seed=100) / 2.0                        # this process alone, per clip
  single = n_samples / single_s / 1e6
  result = {'unit': 'Msamples/s', 'kind': 'port', 'single_process_value': single}
  value, cores, wall, note = single, 1, single_s, ''
  if procs > 1:
    children = []
    try:
      root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
      cmd = [sys.executable, '-m', 'oracle.cpu_baseline', str(clips_per_proc)] + [repr(v) for v in shape]
      env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
      for p in range(procs):
        children.append(subprocess.Popen(cmd + [str(300 + p)], cwd=root, env=env, stdout=subprocess.PIPE,
                                         stderr=subprocess.DEVNULL, text=True))
      deadline = time.monotonic() + timeout_s
      times = []
      for child in children:
        out, _ = child.communicate(timeout=max(1.0, deadline - time.monotonic()))
        if child.returncode != 0:
          raise RuntimeError('worker exit code %d' % child.returncode)
        times.append(float(out.strip().splitlines()[-1]))
      wall = max(times)
      value, cores = procs * clips_per_proc * n_samples / wall / 1e6, procs
    except Exception as exc:                       # noqa: BLE001 - the single-process figure stands
      note = '; workers failed (%r), single process reported' % (exc,)
    finally:
      for child in children:                       # our own children, by handle
        if child.poll() is None:
          child.kill()
          child.wait()
  result.update(value=value, cores=cores, sample=(
      '%d clip(s) of the same workload (one clip = %d samples, K=%d, M=%d) through oracle/ddsp_oracle.py '
      '(numpy fp32 restatement of the TF op chain; TF is not installable here) on %d concurrent worker '
      'process(es), slowest worker %.1f s; one process alone: %.2f Msamples/s; host has %d logical CPUs%s' %
      (cores * (clips_per_proc if cores > 1 else 1), n_samples, n_harmonics, n_bands, cores, wall, single,
       os.cpu_count() or 1, note)))
  return result


def measure_config0(n_clips=3, n_frames=1000, n_harmonics=60, n_samples=64000, sample_rate=16000, f0=200.0):
  """BASELINE.json configs[0]: synths.Harmonic alone, batch 1, 16 kHz, 1000 frames, 60 harmonics
  (gin/models/solo_instrument.gin:18-20), one process - the reference's own CPU-runnable case, as the numpy fp32 port of
  its op chain runs it (f0 = 200 + N(0,1) Hz, the reference tests' regime: processors_test.py:40)."""
  from oracle import ddsp_oracle as O
  x = _make_inputs(n_clips + 1, n_frames, n_harmonics, 2, f0, seed=60)
  O.harmonic(x['amplitudes'][:1], x['harmonic_distribution'][:1], x['f0_hz'][:1], n_samples, sample_rate)     # warm-up
  t0 = time.perf_counter()
  for i in range(1, n_clips + 1):
    s = slice(i, i + 1)
    O.harmonic(x['amplitudes'][s], x['harmonic_distribution'][s], x['f0_hz'][s], n_samples, sample_rate)
  dt = (time.perf_counter() - t0) / n_clips
  return {'value': n_samples / dt / 1e6, 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port', 'ms_per_clip': dt * 1e3,
          'sample': '%d clips of BASELINE configs[0] (synths.Harmonic, batch 1, %d samples @ %d Hz, %d frames, %d harmonics, '
                    'f0 = %g + N(0,1) Hz) through oracle/ddsp_oracle.py in one process' %
                    (n_clips, n_samples, sample_rate, n_frames, n_harmonics, f0)}


if __name__ == '__main__':
  # worker: python -m oracle.cpu_baseline <clips> <n_frames> <n_harmonics> <n_bands> <n_samples> <sample_rate> <f0> <seed>
  _clips, _f, _k, _m, _n, _sr = (int(v) for v in sys.argv[1:7])
  _f0, _seed = float(sys.argv[7]), int(sys.argv[8])
  run_clips(1, _f, _k, _m, _n, _sr, _f0, _seed + 1000)                   # warm-up: imports, page faults
  print(run_clips(_clips, _f, _k, _m, _n, _sr, _f0, _seed), flush=True)
