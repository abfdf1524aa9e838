"""FilteredNoise.ir_design 'vector' (lanes = frames on the vector ALUs) vs 'matrix' (cosine transform on the fp16
matrix cores) vs 'matrix_direct' (the same, magnitudes from HBM to the fragments without LDS staging, noise tile
generated under the load latency) on the bench workload: per-launch time (dispatch events), back-to-back time and agreement.

    python tools/exp_noise_ir.py [batch ...]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
batches = [int(a) for a in sys.argv[1:]] or [32, 128]
for B in batches:
  F, M, N = 1000, 65, 64000
  rng = np.random.default_rng(0)
  mags = ddsp.core.tf_float32(rng.standard_normal((B, F, M)))
  noise = ddsp.core.uniform_noise(B, N, seed=1)
  res = {'batch': B}
  outs = {}
  for design in ('vector', 'matrix', 'matrix_direct'):
    synth = ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
    synth.ir_design = design
    for _ in range(20): synth(mags)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.05:
      for _ in range(20): synth(mags)
      torch.cuda.synchronize()
    outs[design] = synth(mags, noise=noise)
    _lib.profile_begin(None, max_records=512)
    for _ in range(50): synth(mags)
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    steps = 300
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): synth(mags)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    res[design] = {'us_per_call_back_to_back': dt * 1e6, 'kernel_us': {k: v[0] / v[1] * 1e3 for k, v in bd.items()}}
  res['max_abs_diff'] = {d: float((outs['vector'] - outs[d]).abs().max()) for d in ('matrix', 'matrix_direct')}
  print(json.dumps(res))
