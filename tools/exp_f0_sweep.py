"""f0 sweep of the default Harmonic kernel: per-launch time (dispatch events) of
harm_table_kernel at f0 = 70 ... 1000 Hz, batch 32 and 128.  The table reads of one wavefront instruction are an
arithmetic progression of positions with stride 512 f0 / sr, so LDS bank conflicts depend on f0.

    python tools/exp_f0_sweep.py [batch ...]
DDSP_SWEEP_JITTER (default 1.0): the standard deviation of f0 around its centre, in Hz - 0 keeps every harmonic on one side of
Nyquist for the whole clip (no frame with a harmonic crossing it: the kernel's per-sample correction path never runs), which
separates the cost of the scattered table reads from the cost of that path.  DDSP_SWEEP_F0S: a comma-separated list of centres.
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib, build
build.build()
batches = [int(a) for a in sys.argv[1:]] or [32, 128]
F0S = [float(v) for v in os.environ['DDSP_SWEEP_F0S'].split(',')] if os.environ.get('DDSP_SWEEP_F0S') else \
    [70.0, 125.0, 200.0, 250.0, 333.0, 400.0, 500.0, 666.0, 1000.0]
JITTER = float(os.environ.get('DDSP_SWEEP_JITTER', '1.0'))
for B in batches:
  F, K, N = 1000, 100, 64000
  rng = np.random.default_rng(0)
  amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
  hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
  res = {'batch': B, 'jitter_hz': JITTER, 'us': {}}
  for f0c in F0S:
    f0 = ddsp.core.tf_float32(f0c + JITTER * rng.standard_normal((B, F, 1)))
    synth = ddsp.synths.Harmonic(n_samples=N)
    synth.kernel = os.environ.get('DDSP_SWEEP_KERNEL', 'auto')
    # (clock settle: a box reaches its sustained clocks after tenths of a second of load - without it the first f0 of
    # the list is measured 25 % slow, r03q)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < (0.5 if f0c == F0S[0] else 0.1):
      for _ in range(30): synth(amps, hd, f0)
      torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=512)
    for _ in range(50): synth(amps, hd, f0)
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    res['us'][str(f0c)] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in bd.items()}
  vals = [list(v.values())[0] for v in res['us'].values()]
  res['worst_over_best'] = round(max(vals) / min(vals), 3)
  print(json.dumps(res))
