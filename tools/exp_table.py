"""harm_table_kernel vs harm_fused_kernel on the bench workload: per-launch time (dispatch events) and agreement.

    python tools/exp_table.py [batch ...]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
batches = [int(a) for a in sys.argv[1:]] or [32, 128]
for B in batches:
  F, K, N = 1000, 100, 64000
  rng = np.random.default_rng(0)
  amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
  hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
  f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
  res = {'batch': B}
  outs = {}
  for kernel in ('auto', 'direct'):
    synth = ddsp.synths.Harmonic(n_samples=N)
    synth.kernel = kernel
    for _ in range(20): synth(amps, hd, f0)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.05:
      for _ in range(20): synth(amps, hd, f0)
      torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=512)
    for _ in range(50): outs[kernel] = synth(amps, hd, f0)
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    steps = 300
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): synth(amps, hd, f0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    res[kernel] = {'us_per_call_back_to_back': dt * 1e6, 'kernel_us': {k: v[0] / v[1] * 1e3 for k, v in bd.items()}}
  res['max_abs_diff'] = float((outs['auto'] - outs['direct']).abs().max())
  print(json.dumps(res))
