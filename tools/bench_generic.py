"""The kernels BEHIND the specialised fast paths, timed on the shapes the reference's own tests use (ddsp/processors_test.py:28-73,
synths_test.py:43-50: 99 harmonics, 100 or 256 noise magnitudes, 64 000 samples, 1000 frames) and on shapes the fast paths
refuse (a hop that is not a multiple of 64, more than 128 harmonics, a cropped window): per call (host clock over back-to-back
calls) and per kernel (dispatch events).  For the record - the round-2 verdict's "specialisation cliff".

    python tools/bench_generic.py [batch]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)


def timed(fn, reps=50):
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.3:
    for _ in range(5): fn()
    torch.cuda.synchronize()
  t1 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t1) / reps
  _lib.profile_begin(None, max_records=1024)
  for _ in range(20): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  return dt, {k: round(v[0] / v[1] * 1e3, 1) for k, v in bd.items()}


def tf(x):
  return ddsp.core.tf_float32(x.astype(np.float32))

cases = []
def harmonic_case(name, k, f, n, sr=16000, f0c=70.0):
  amps, hd, f0 = tf(rng.standard_normal((B, f, 1))), tf(rng.standard_normal((B, f, k))), tf(f0c + rng.standard_normal((B, f, 1)))
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  dt, bd = timed(lambda: synth(amps, hd, f0))
  cases.append({'what': name, 'batch': B, 'us_per_call': round(dt * 1e6, 1), 'Msamples_per_s': round(B * n / dt / 1e6), 'kernel_us': bd})
def noise_case(name, m, f, n, window_size=0):
  mags = tf(rng.standard_normal((B, f, m)))
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=window_size)
  dt, bd = timed(lambda: synth(mags))
  cases.append({'what': name, 'batch': B, 'us_per_call': round(dt * 1e6, 1), 'Msamples_per_s': round(B * n / dt / 1e6), 'kernel_us': bd})

harmonic_case('Harmonic, 100 harmonics, 1000 frames of 64 (the fast path, for comparison)', 100, 1000, 64000)
harmonic_case('Harmonic, 99 harmonics (processors_test.py): the wavetable kernel since the end of round 3', 99, 1000, 64000)
harmonic_case('Harmonic, 100 harmonics, 250 frames of 256 samples', 100, 250, 64000)
harmonic_case('Harmonic, 100 harmonics, 640 frames of 100 samples (hop not a multiple of 64: the wavetable kernel with masked tiles since round 4)', 100, 640, 64000)
harmonic_case('Harmonic, 160 harmonics at 32 kHz (K > 128: the ten-tap wavetable instances)', 160, 1000, 64000, sr=32000, f0c=70.0)
harmonic_case('Harmonic, 200 harmonics at 48 kHz (the last band count of the wavetable kernel)', 200, 1000, 64000, sr=48000, f0c=70.0)
harmonic_case('Harmonic, 256 harmonics at 48 kHz (201 … 256: the direct sum, harm_fused_kernel)', 256, 1000, 64000, sr=48000, f0c=70.0)
harmonic_case('Harmonic, 400 harmonics at 48 kHz, f0 = 50 Hz (257 … 512: the plain kernels, envelopes through HBM)', 400, 1000, 64000, sr=48000, f0c=50.0)
noise_case('FilteredNoise, 65 magnitudes, 1000 frames of 64 (the fast path, for comparison)', 65, 1000, 64000)
noise_case('FilteredNoise, 100 magnitudes (synths_test.py): 198-tap IR - one launch, taps designed per tile in LDS (filtered_noise_general.hip)', 100, 1000, 64000)
noise_case('FilteredNoise, 256 magnitudes: 510-tap IR - IR design as a matrix product + Toeplitz FIR on the matrix cores (two launches)', 256, 1000, 64000)
noise_case('FilteredNoise, 65 magnitudes, window_size 257 (the constructor default: cropped window)', 65, 1000, 64000, window_size=257)
noise_case('FilteredNoise, 65 magnitudes, 640 frames of 100 samples (pieces of 64 + 36: filtered_noise_general.hip)', 65, 640, 64000)
for c in cases:
  print(json.dumps(c))
