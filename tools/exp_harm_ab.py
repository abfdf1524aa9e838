"""Harmonic (harm_table_kernel) per-launch time at several f0 / batches and the headline two-stream step - for same-session A/B of
library variants (tools/build_variant.sh + tools/with_lib.py).

    python tools/exp_harm_ab.py [batches]
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
  rec = {'batch': B, 'lib': os.path.basename(_lib.LIB_PATH) if getattr(_lib, 'LIB_PATH', None) else 'product'}
  rng = np.random.default_rng(0)
  mags = ddsp.core.tf_float32(rng.standard_normal((B, F, 65)))
  amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
  hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
  harm = ddsp.synths.Harmonic(n_samples=N)
  z = ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
  for f0c in (70.0, 200.0, 333.0):
    f0 = ddsp.core.tf_float32(f0c + rng.standard_normal((B, F, 1)))
    fn = lambda: harm(amps, hd, f0)
    settle(fn)
    rec['harm_%g_us' % f0c] = per_launch_us(fn)['harm_table_kernel']
    if f0c == 70.0:
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
  print('HARM_AB', json.dumps(rec), flush=True)
