"""Harmonic and FilteredNoise on two free-running HIP streams (no per-step join) vs back to back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
F, K, M, N = 1000, 100, 65, 64000
dev = {k: ddsp.core.tf_float32(v) for k, v in dict(
    a=rng.standard_normal((B, F, 1)), hd=rng.standard_normal((B, F, K)),
    f0=70 + rng.standard_normal((B, F, 1)), m=rng.standard_normal((B, F, M))).items()}
harm, noise = ddsp.synths.Harmonic(), ddsp.synths.FilteredNoise(window_size=0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, steps=100):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    if mode == 'seq':
      harm(dev['a'], dev['hd'], dev['f0']); noise(dev['m'])
    else:
      with torch.cuda.stream(s1): harm(dev['a'], dev['hd'], dev['f0'])
      with torch.cuda.stream(s2): noise(dev['m'])
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / steps * 1e6
for mode in ('seq', 'free', 'seq', 'free'):
  run(mode, 10)
  print('B=%d %-5s %.1f us/step' % (B, mode, run(mode)))
