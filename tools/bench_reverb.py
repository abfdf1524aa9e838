"""Reverb (effects.Reverb) on its own: time, kernel breakdown, HBM roofline fraction.

    python tools/bench_reverb.py [batch] [n_samples] [ir_size] [ir_batch]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 48000
BIR = int(sys.argv[4]) if len(sys.argv) > 4 else B
rng = np.random.default_rng(0)
audio = ddsp.core.tf_float32(rng.standard_normal((B, N)))
ir = ddsp.core.tf_float32(rng.standard_normal((BIR, L)) * np.exp(-np.arange(L) / (0.2 * L)))
rev = ddsp.effects.Reverb(add_dry=True)
for _ in range(5): rev(audio, ir)
t_settle = time.perf_counter()
while time.perf_counter() - t_settle < 0.05:      # the GPU needs ~20 ms of load to reach its sustained clock
  for _ in range(5): rev(audio, ir)
  torch.cuda.synchronize()
_lib.profile_begin(None, max_records=256)
for _ in range(10): rev(audio, ir)
torch.cuda.synchronize()
bd = _lib.profile_end()
steps = 100
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): out = rev(audio, ir)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
alg = 4.0 * (B * 2 * N + BIR * L)             # audio in, audio out, IR in
print(json.dumps({
    'workload': 'effects.Reverb(add_dry=True): batch=%d, %d samples, %d-tap IR (%s)' % (B, N, L, 'one per row' if BIR == B else 'shared'),
    'ms_per_call': dt * 1e3, 'Msamples_per_s': B * N / dt / 1e6,
    'kernel_us': {k: v[0] / v[1] * 1e3 for k, v in bd.items()},
    'algorithmic_bytes': alg, 'hbm_frac': alg / dt / 8e12}))
