"""FilteredNoise (noise_mfma65_kernel) per-launch time at 23 and 11 bits and with supplied noise, and the headline two-stream step.

    python tools/exp_noise_ab.py [batches]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddsp_amd import _lib
import ddsp_amd as ddsp

def settle(fn, secs=0.05):
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize()

def per_launch_us(fn, n=200):
  _lib.profile_begin(None, max_records=4096)
  for _ in range(n): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  return {k: round(v[0] / v[1] * 1e3, 2) for k, v in bd.items()}

F, K, N = 1000, 100, 64000
for B in [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['32', '128'])]:
  rng = np.random.default_rng(0)
  mags = ddsp.core.tf_float32(rng.standard_normal((B, F, 65)))
  amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
  hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
  f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
  x = ddsp.core.tf_float32(rng.uniform(-1, 1, (B, N)))
  harm = ddsp.synths.Harmonic(n_samples=N)
  rec = {'batch': B}
  for bits in (23, 11):
    z = ddsp.synths.FilteredNoise(n_samples=N, window_size=0, noise_bits=bits)
    fn = lambda: z(mags)
    settle(fn)
    rec['noise_%d_bits_us' % bits] = per_launch_us(fn)
    if bits == 23:
      fs = lambda: z(mags, noise=x)
      settle(fs)
      rec['noise_supplied_us'] = per_launch_us(fs)
      s_h, s_z, s_0 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
      def step():
        torch.cuda.set_stream(s_h); harm(amps, hd, f0)
        torch.cuda.set_stream(s_z); z(mags)
        torch.cuda.set_stream(s_0)
      settle(step)
      torch.cuda.synchronize(); t0 = time.perf_counter(); n = 500
      for _ in range(n): step()
      torch.cuda.synchronize()
      rec['two_stream_step_us'] = round((time.perf_counter() - t0) / n * 1e6, 2)
  print('NOISE_AB', json.dumps(rec), flush=True)
