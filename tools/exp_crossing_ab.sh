# Same-session A/B of two library variants (tools/build_variant_lib.sh base / new) on the Nyquist-crossing cases of harm_table_kernel:
# per-launch time at batch 128 for f0 centre +- jitter pairs, then the GPU tests of the Harmonic paths.  gpurun -- bash tools/exp_crossing_ab.sh
python tools/exp_ab.py base new --rounds 3 --only harm 2>&1 | tail -2 | cut -c1-260
for v in base new; do python - <<PY
import os, sys, json, time
sys.path.insert(0, '.')
import numpy as np, torch
from ddsp_amd import _lib
_lib.LIB_PATH = 'tools/bin/libddsp_amd_$v.so'
import ddsp_amd as ddsp
B, F, K, N = 128, 1000, 100, 64000
rng = np.random.default_rng(0)
amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1))); hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
res = {}
for f0c, jit in ((70.0, 1.0), (200.0, 1.0), (250.0, 0.0), (250.0, 1.0), (333.0, 1.0), (1000.0, 1.0), (220.0, 6.0)):
  f0 = ddsp.core.tf_float32(f0c + jit * rng.standard_normal((B, F, 1)))
  synth = ddsp.synths.Harmonic(n_samples=N)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.15:
    for _ in range(30): synth(amps, hd, f0)
    torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=512)
  for _ in range(50): synth(amps, hd, f0)
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  res['%g+-%g' % (f0c, jit)] = round(sum(v[0] for v in bd.values()) / 50 * 1e3, 1)
print('$v', json.dumps(res))
PY
done
python -m pytest tests -m gpu -q -k "harmonic or nyquist or fused_add or config" 2>&1 | tail -1
