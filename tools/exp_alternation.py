"""What does alternating with another kernel cost harm_table_kernel / noise_mfma65_kernel?  Per-launch dispatch events at batch 128
for: the kernel back to back; alternating with the other synth; alternating with a large elementwise torch kernel (streams
150 MB through L2 / MALL, tiny code); alternating with a one-element torch kernel (no data, tiny code).

    python tools/exp_alternation.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib
B, F, K, N = 128, 1000, 100, 64000
rng = np.random.default_rng(0)
amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
mags = ddsp.core.tf_float32(rng.standard_normal((B, F, 65)))
harm = ddsp.synths.Harmonic(n_samples=N)
noise = ddsp.synths.FilteredNoise(n_samples=N)
big = torch.zeros(37_500_000, device='cuda')       # 150 MB
one = torch.zeros(1, device='cuda')
others = {'nothing (back to back)': lambda: None, 'the other synth': None, 'a 150 MB elementwise kernel': lambda: big.add_(1.0),
          'a one-element kernel': lambda: one.add_(1.0)}
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
  for _ in range(20): harm(amps, hd, f0); noise(mags)
  torch.cuda.synchronize()
for name, fn, other_synth in (('harm_table_kernel', lambda: harm(amps, hd, f0), lambda: noise(mags)),
                              ('noise_mfma65_kernel', lambda: noise(mags), lambda: harm(amps, hd, f0))):
  for label, other in others.items():
    if other is None: other = other_synth
    for _ in range(20): fn(); other()
    torch.cuda.synchronize()
    _lib.profile_begin([name], max_records=512)
    for _ in range(200): fn(); other()
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    print(json.dumps({'kernel': name, 'alternating_with': label, 'us_per_launch': round(bd[name][0] / bd[name][1] * 1e3, 2)}))
# --- what is it about "another kernel in between"?  (1) the same kernel with the GPU left idle between launches (the host waits
# for each launch, then sleeps); (2) harm_table_kernel alternating with its own fused-Add instance (other code, same work)
def timed(name, fn, between, n=100):
  for _ in range(10): fn(); between()
  torch.cuda.synchronize()
  _lib.profile_begin([name], max_records=512)
  for _ in range(n): fn(); between()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  return round(bd[name][0] / bd[name][1] * 1e3, 2)
def idle(us):
  def f():
    torch.cuda.synchronize()
    t = time.perf_counter()
    while time.perf_counter() - t < us * 1e-6: pass
  return f
out_add = torch.zeros(B, N, device='cuda')
for us in (0, 20, 100, 1000):
  print(json.dumps({'kernel': 'harm_table_kernel', 'gpu_idle_between_launches_us': us, 'us_per_launch': timed('harm_table_kernel', lambda: harm(amps, hd, f0), idle(us))}))
for us in (0, 100):
  print(json.dumps({'kernel': 'noise_mfma65_kernel', 'gpu_idle_between_launches_us': us, 'us_per_launch': timed('noise_mfma65_kernel', lambda: noise(mags), idle(us))}))
print(json.dumps({'kernel': 'harm_table_kernel', 'alternating_with': 'its own fused-Add instance (both counted)',
                  'us_per_launch': timed('harm_table_kernel', lambda: harm(amps, hd, f0), lambda: harm.call_add(amps, hd, f0, out_add))}))
print(json.dumps({'kernel': 'harm_table_kernel', 'alternating_with': 'nothing, 200 launches', 'us_per_launch': timed('harm_table_kernel', lambda: harm(amps, hd, f0), lambda: None, 200)}))
