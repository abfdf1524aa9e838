import json, os, sys, time, subprocess
HERE='/root/repo/tools'
CHILD = r'''
import json, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from ddsp_amd import _lib
if %(lib)r: _lib.LIB_PATH = %(lib)r
import ddsp_amd as ddsp
T = ddsp.core.tf_float32
res = {}
for name, B, F, N in [('fs192_b32', 32, 2500, 480000), ('fs128_b32', 32, 500, 64000), ('fs256_b32', 32, 250, 64000), ('fs64_b128', 128, 1000, 64000)]:
  rng = np.random.default_rng(0)
  mags = T(rng.standard_normal((B, F, 65)))
  synth = ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
  fn = lambda: synth(mags)
  for _ in range(10): fn()
  torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=256)
  for _ in range(40): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  res[name] = round(sum(v[0] for v in bd.values()) / max(v[1] for v in bd.values()) * 1e3, 2)
print('AB ' + json.dumps(res))
'''
for rnd in range(2):
  for v in ['product', 'prevnoise']:
    lib = '' if v == 'product' else os.path.join(HERE, 'bin', 'libddsp_amd_%s.so' % v)
    out = subprocess.run([sys.executable, '-c', CHILD % dict(lib=lib)], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('AB ')]
    print(v, line[0][3:] if line else 'FAILED ' + out.stderr[-300:], flush=True)
