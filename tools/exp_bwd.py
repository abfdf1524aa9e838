"""Per-launch time (dispatch events) of the kernels of Harmonic's backward pass for library variants inside ONE gpurun call.

    python tools/exp_bwd.py product bt4 plain ...      (plain = the product library with DDSP_EXP_HARM_BWD=plain: harm_bwd_pq_kernel's sums)
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
for name, B, F, K, N, sr, f0c in [('b32_70', 32, 1000, 100, 64000, 16000, 70.0), ('b32_300', 32, 1000, 100, 64000, 16000, 300.0),
                                  ('b32_k128', 32, 1000, 128, 64000, 16000, 55.0), ('b32_hop192', 8, 2500, 100, 480000, 48000, 120.0),
                                  ('b128_70', 128, 1000, 100, 64000, 16000, 70.0)]:
  rng = np.random.default_rng(0)
  amps = T(rng.standard_normal((B, F, 1))).requires_grad_(True)
  hd = T(rng.standard_normal((B, F, K))).requires_grad_(True)
  f0 = T(f0c + rng.standard_normal((B, F, 1)))
  g = T(rng.standard_normal((B, N)))
  synth = ddsp.synths.Harmonic(n_samples=N, sample_rate=sr)
  def fn():
    amps.grad = None; hd.grad = None
    synth(amps, hd, f0).backward(g)
  for _ in range(5): fn()
  torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=512)
  for _ in range(20): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  res[name] = {k.replace('_kernel', '').replace('harm_', ''): round(v[0] / v[1] * 1e3, 1) for k, v in bd.items()}
print('AB ' + json.dumps(res))
'''
for v in sys.argv[1:] or ['product']:
  env = dict(os.environ)
  lib = ''
  if v == 'plain':
    env['DDSP_EXP_HARM_BWD'] = 'plain'
  elif v != 'product':
    lib = os.path.join(HERE, 'bin', 'libddsp_amd_%s.so' % v)
  out = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, lib=lib)], capture_output=True, text=True, env=env)
  line = [l for l in out.stdout.splitlines() if l.startswith('AB ')]
  print(v, line[0][3:] if line else 'FAILED ' + out.stderr[-600:], flush=True)
