"""losses.SpectralLoss forward on its own: time per call and per-scale kernel time.

    python tools/bench_spectral_loss.py [batch] [n_samples]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
rng = np.random.default_rng(0)
t = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N)))
a = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N)))
loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
for _ in range(5): loss(t, a)
t_settle = time.perf_counter()
while time.perf_counter() - t_settle < 0.05:      # the GPU needs ~20 ms of load to reach its sustained clock
  for _ in range(5): loss(t, a)
  torch.cuda.synchronize()
_lib.profile_begin(None, max_records=64)
for _ in range(10): loss(t, a)
torch.cuda.synchronize()
bd = _lib.profile_end()
steps = 100
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): out = loss(t, a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
ag = a.clone().requires_grad_(True)
for _ in range(5):
  ag.grad = None; loss(t, ag).backward()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(50):
  ag.grad = None; loss(t, ag).backward()
torch.cuda.synchronize(); dt_fb = (time.perf_counter() - t1) / 50
alg = 4.0 * 2 * B * N                  # both signals read once (the 6 scales x 4 overlaps re-read from L2)
print(json.dumps({'workload': 'SpectralLoss(mag+logmag, 6 scales) batch=%d, %d samples' % (B, N),
                  'ms_per_call': dt * 1e3, 'ms_per_fwd_bwd': dt_fb * 1e3, 'Msamples_per_s': B * N / dt / 1e6,
                  'kernel_us': {k: v[0] / v[1] * 1e3 for k, v in bd.items()},
                  'algorithmic_bytes': alg, 'hbm_frac': alg / dt / 8e12}))
