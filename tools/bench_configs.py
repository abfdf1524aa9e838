"""BASELINE.json's other configurations, per GPU (they are parity-test cases, not bench lines: this is for the record).

  configs[2]  Harmonic + FilteredNoise + losses.SpectralLoss (6 scales), batch 128
  configs[3]  ProcessorGroup Harmonic + FilteredNoise + effects.Reverb (48 000-tap IR), batch 128 per GPU (1024 over 8)
  configs[4]  48 kHz, 200 harmonics, 10 s clips, batch 32 per GPU (256 over 8): the wavetable kernel (129 .. 200 harmonics: ten taps) since the end of round 3

    python tools/bench_configs.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
rng = np.random.default_rng(0)


def timed(fn, reps):
  for _ in range(5): fn()
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.05:
    fn(); torch.cuda.synchronize()
  torch.cuda.synchronize(); t1 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t1) / reps


def inputs(b, f, k, f0=70.0):
  return (ddsp.core.tf_float32(rng.standard_normal((b, f, 1))), ddsp.core.tf_float32(rng.standard_normal((b, f, k))),
          ddsp.core.tf_float32(f0 + rng.standard_normal((b, f, 1))), ddsp.core.tf_float32(rng.standard_normal((b, f, 65))))

out = []
# configs[2]
b, f, k, n = 128, 1000, 100, 64000
a, hd, f0, mags = inputs(b, f, k)
harm, noise = ddsp.synths.Harmonic(n_samples=n), ddsp.synths.FilteredNoise(n_samples=n, window_size=0)
loss = ddsp.losses.SpectralLoss(logmag_weight=1.0)
target = ddsp.core.tf_float32(0.3 * rng.standard_normal((b, n)))
add = ddsp.processors.Add()
dt = timed(lambda: loss(target, add(harm(a, hd, f0), noise(mags))), 50)
out.append({'config': 'configs[2]: Harmonic + FilteredNoise + Add + SpectralLoss(mag + logmag, 6 scales), batch 128, 4 s @ 16 kHz',
            'ms_per_step': dt * 1e3, 'Msamples_per_s': b * n / dt / 1e6})
# configs[3]
ir = ddsp.core.tf_float32(0.05 * rng.standard_normal((b, 48000)))
reverb = ddsp.effects.Reverb(add_dry=True)
dt = timed(lambda: reverb(add(harm(a, hd, f0), noise(mags)), ir), 50)
out.append({'config': 'configs[3] per GPU: Harmonic + FilteredNoise + Add + Reverb(48 000-tap IR per clip), batch 128 (1024 over 8 GPUs)',
            'ms_per_step': dt * 1e3, 'Msamples_per_s': b * n / dt / 1e6})
del a, hd, f0, mags, target, ir
# configs[4]
b, f, k, n, sr = 32, 2500, 200, 480000, 48000
a, hd, f0, mags = inputs(b, f, k)
harm = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)
dt = timed(lambda: (harm(a, hd, f0), noise(mags)), 30)
out.append({'config': 'configs[4] per GPU: Harmonic (200 harmonics: wavetable kernel, ten taps) + FilteredNoise, 10 s @ 48 kHz, frame size 192, batch 32 (256 over 8 GPUs)',
            'ms_per_step': dt * 1e3, 'Msamples_per_s': b * n / dt / 1e6,
            'algorithmic_GBs': 4 * b * (f * (k + 2) + n + f * 65 + n) / dt / 1e9})
for o in out:
  print(json.dumps(o))
