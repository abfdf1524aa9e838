"""A short SpectralLoss run for tools/pmc.sh (PMC_CMD): a few forward and value + gradient calls at the given batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
t = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, 64000)))
a = ddsp.core.tf_float32(0.3 * rng.standard_normal((B, 64000)))
loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
for _ in range(3): loss(t, a)
ag = a.clone().requires_grad_(True)
for _ in range(3):
  ag.grad = None; loss(t, ag).backward()
torch.cuda.synchronize()
