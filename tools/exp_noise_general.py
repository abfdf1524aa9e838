"""Time accounting of the general FilteredNoise kernels (filtered_noise_general.hip): per-launch time (dispatch events) of the
IR design and of the FIR on the reference tests' shapes, for library variants with parts of a kernel compiled out
(-DDDSP_GF_NO_TAPS / _NOISE / _MFMA / _STORE / _ZERO, -DDDSP_GI_NO_STORE; tools/build_variant_lib.sh), inside ONE gpurun call.

    python tools/exp_noise_general.py product notaps nonoise ...
"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from ddsp_amd import _lib
if %(lib)r: _lib.LIB_PATH = %(lib)r
import ddsp_amd as ddsp
T = ddsp.core.tf_float32
res = {}
for name, B, F, M, N, ws, given in [('m100', 32, 1000, 100, 64000, 0, False), ('m100_given', 32, 1000, 100, 64000, 0, True),
                                    ('m256', 32, 1000, 256, 64000, 0, False), ('m65_hop100', 32, 640, 65, 64000, 0, False),
                                    ('m100_b128', 128, 1000, 100, 64000, 0, False)]:
  rng = np.random.default_rng(0)
  mags = T(rng.standard_normal((B, F, M)))
  z = T(rng.uniform(-1, 1, (B, N))) if given else None
  synth = ddsp.synths.FilteredNoise(n_samples=N, window_size=ws)
  fn = (lambda: synth(mags, noise=z)) if given else (lambda: synth(mags))
  for _ in range(10): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.1:
    for _ in range(5): fn()
    torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=512)
  for _ in range(30): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  res[name] = {k.replace('_kernel', ''): round(v[0] / v[1] * 1e3, 1) for k, v in bd.items()}
print('AB ' + json.dumps(res))
'''
for v in sys.argv[1:] or ['product']:
  env = dict(os.environ)
  if v.startswith('plan='):                      # plan=W,R: the FIR's cut pinned (DDSP_EXP_GF_PLAN), product library
    env['DDSP_EXP_GF_PLAN'] = v[5:]
    lib = ''
  else:
    lib = '' if v == 'product' else os.path.join(HERE, 'bin', 'libddsp_amd_%s.so' % v)
  out = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, lib=lib)], capture_output=True, text=True, env=env)
  line = [l for l in out.stdout.splitlines() if l.startswith('AB ')]
  print(v, line[0][3:] if line else 'FAILED ' + out.stderr[-400:], flush=True)
