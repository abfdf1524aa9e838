import json, os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib
B, N = 32, 64000
rng = np.random.default_rng(0)
a = ddsp.core.tf_float32(rng.standard_normal((B, N)))
t = ddsp.core.tf_float32(rng.standard_normal((B, N)))
ag = a.clone().requires_grad_(True)
for sizes in ([2048], [1024], [512], [256], [128], [64], [2048, 1024, 512, 256, 128, 64]):
  loss = ddsp.losses.SpectralLoss(fft_sizes=tuple(sizes), mag_weight=1.0, logmag_weight=1.0)
  for _ in range(20): loss(t, a)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.2:
    for _ in range(10): loss(t, a)
    torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=1024)
  for _ in range(50): loss(t, a)
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  fw = {k: round(v[0] / 50 * 1e3, 1) for k, v in bd.items()}
  _lib.profile_begin(None, max_records=1024)
  for _ in range(50):
    ag.grad = None
    loss(t, ag).backward()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  bw = {k: round(v[0] / 50 * 1e3, 1) for k, v in bd.items()}
  print(json.dumps({'fft_sizes': sizes, 'forward_us': fw, 'fwd_bwd_us': bw}))
