"""Forward + backward of the synths through torch.autograd: time per call, kernels.

    python tools/bench_backward.py [batch]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F, K, M, N = 1000, 100, 65, 64000
rng = np.random.default_rng(0)
amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1))).requires_grad_(True)
hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K))).requires_grad_(True)
f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
mags = ddsp.core.tf_float32(rng.standard_normal((B, F, M))).requires_grad_(True)
g = ddsp.core.tf_float32(rng.standard_normal((B, N)))
harm = ddsp.synths.Harmonic(n_samples=N)
noise = ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
spec = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)       # ae.gin:36-41
add = ddsp.processors.Add()
def step(with_noise):
  # every op on the path is this library's (processors.Add, not torch's +; the upstream gradient handed to backward(),
  # not made by torch's mul / sum): what is timed beside the kernels is the autograd node's host code, nothing of torch's
  amps.grad = hd.grad = mags.grad = None
  if with_noise:
    # Harmonic + Add as one launch and one autograd node (what ProcessorGroup does for the ae.gin DAG since round 4;
    # DDSP_BENCH_UNFUSED_ADD=1: the three processors one by one, as rounds 1-3 timed it)
    y = add(harm(amps, hd, f0), noise(mags)) if os.environ.get('DDSP_BENCH_UNFUSED_ADD') else harm.call_add(amps, hd, f0, noise(mags))
  else:
    y = harm(amps, hd, f0)
  if with_noise == 'loss':
    spec(g, y).backward()           # the whole differentiable path of ae.gin, every kernel native
  else:
    y.backward(g)
res = {}
for name, wn in (('harmonic', False), ('harmonic+noise', True), ('harmonic+noise+spectral_loss', 'loss')):
  if wn and not hasattr(noise, '_backward'):
    continue
  t_settle = time.perf_counter()
  while time.perf_counter() - t_settle < 0.05:    # clock settle, as bench.py
    for _ in range(5): step(wn)
    torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=512)
  for _ in range(10): step(wn)
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  steps = 200
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): step(wn)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
  res[name] = {'ms_per_fwd_bwd': dt * 1e3, 'Msamples_per_s': B * N / dt / 1e6,
               'kernel_us': {k: round(v[0] / v[1] * 1e3, 1) for k, v in bd.items()}}
print(json.dumps({'workload': 'forward + backward through torch.autograd, batch=%d, 64000 samples, K=100, M=65' % B, **res}))
