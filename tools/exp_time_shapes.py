"""Per-launch time (dispatch events) of Harmonic.__call__ over a list of shapes and f0 regimes, for several library variants
taken in turn inside ONE gpurun call (boxes differ by more than most kernel changes).

    python tools/exp_time_shapes.py [--rounds 3] product r3 ...     ("product" = ddsp_amd/lib, others = tools/bin/libddsp_amd_<name>.so)
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
SHAPES = [  # name, B, F, K, N, sr, f0 centre, jitter ('vibrato' = 6 Hz at 5.5 Hz), method
  ('b128_70', 128, 1000, 100, 64000, 16000, 70.0, 1.0, 'window'),
  ('b128_200', 128, 1000, 100, 64000, 16000, 200.0, 1.0, 'window'),
  ('b128_220vib', 128, 1000, 100, 64000, 16000, 220.0, 'vibrato', 'window'),
  ('b128_333', 128, 1000, 100, 64000, 16000, 333.0, 1.0, 'window'),
  ('b128_500', 128, 1000, 100, 64000, 16000, 500.0, 1.0, 'window'),
  ('b128_203', 128, 1000, 100, 64000, 16000, 203.0, 1.0, 'window'),
  ('b32_70', 32, 1000, 100, 64000, 16000, 70.0, 1.0, 'window'),
  ('b32_k128', 32, 1000, 128, 64000, 16000, 55.0, 0.5, 'window'),
  ('b32_k128_hop128', 32, 500, 128, 64000, 16000, 55.0, 0.5, 'window'),
  ('config5_b32', 32, 2500, 200, 480000, 48000, 100.0, 1.0, 'linear'),
]
only = %(only)r
res = {}
for name, B, F, K, N, sr, f0c, jit, method in SHAPES:
  if only and name not in only.split(','): continue
  rng = np.random.default_rng(0)
  amps = T(rng.standard_normal((B, F, 1)))
  hd = T(rng.standard_normal((B, F, K)))
  if jit == 'vibrato':
    t = np.arange(F)[None, :, None] / 250.0
    f0 = T(f0c + 6.0 * np.sin(2 * np.pi * 5.5 * t + rng.uniform(0, 6.28, (B, 1, 1))))
  else:
    f0 = T(f0c + jit * rng.standard_normal((B, F, 1)))
  harm = ddsp.synths.Harmonic(n_samples=N, sample_rate=sr, amp_resample_method=method)
  fn = lambda: harm(amps, hd, f0)
  for _ in range(20): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 0.08:
    for _ in range(10): fn()
    torch.cuda.synchronize()
  _lib.profile_begin(None, max_records=512)
  for _ in range(60): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  res[name] = round(sum(v[0] for v in bd.values()) / max(v[1] for v in bd.values()) * 1e3, 2)
print('AB ' + json.dumps(res))
'''

def main():
  args = sys.argv[1:]
  rounds, only = 3, ''
  if '--rounds' in args:
    i = args.index('--rounds'); rounds = int(args[i + 1]); del args[i:i + 2]
  if '--only' in args:
    i = args.index('--only'); only = args[i + 1]; del args[i:i + 2]
  names = args or ['product']
  runs = {n: [] for n in names}
  for r in range(rounds):
    for n in names:
      lib = '' if n == 'product' else os.path.join(HERE, 'bin', 'libddsp_amd_%s.so' % n)
      out = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'lib': lib, 'only': only}], capture_output=True, text=True)
      line = [l for l in out.stdout.split('\n') if l.startswith('AB ')]
      if not line:
        print(n, 'FAILED', out.stderr[-800:]); continue
      runs[n].append(json.loads(line[0][3:]))
  import statistics
  for n in names:
    if runs[n]:
      print(json.dumps({'variant': n, 'median_us': {k: statistics.median(r[k] for r in runs[n]) for k in runs[n][0]},
                        'all': {k: [r[k] for r in runs[n]] for k in runs[n][0]}}))

if __name__ == '__main__':
  main()
