"""Time of the SpectralLoss kernels per FFT size, batch 32 x 64000: forward (stft_l1_kernel) and value + gradient in one
pass (stft_l1_bwd_kernel, through torch.autograd)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
rng = np.random.default_rng(0)
B, N = 32, 64000
t = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N))); a = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, N)))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.05:
  ddsp.losses.SpectralLoss()(t, a); torch.cuda.synchronize()
for S in (2048, 1024, 512, 256, 128, 64):
  loss = ddsp.losses.SpectralLoss(fft_sizes=(S,), logmag_weight=1.0)
  for _ in range(5): loss(t, a)
  torch.cuda.synchronize(); t1 = time.perf_counter()
  for _ in range(50): loss(t, a)
  torch.cuda.synchronize()
  fwd = (time.perf_counter() - t1) / 50 * 1e6
  ag = a.clone().requires_grad_(True)
  for _ in range(5):
    ag.grad = None; loss(t, ag).backward()
  torch.cuda.synchronize(); t1 = time.perf_counter()
  for _ in range(50):
    ag.grad = None; loss(t, ag).backward()
  torch.cuda.synchronize()
  print('S=%4d  forward %.1f us   value + gradient %.1f us' % (S, fwd, (time.perf_counter() - t1) / 50 * 1e6))
