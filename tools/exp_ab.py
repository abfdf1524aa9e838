"""Same-session A/B of library variants (tools/build_variant.sh): per-launch time (dispatch events) of the two
default kernels at batch 32 and 128, the variants taken in turn, `rounds` times; medians.

    python tools/exp_ab.py base new [--rounds 3] [--only harm|noise]
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
only = %(only)r
res = {}
for B in ((32, 128) if only != 'loss' else ()):
  F, K, N = 1000, 100, 64000
  rng = np.random.default_rng(0)
  amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
  hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
  f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
  mags = ddsp.core.tf_float32(rng.standard_normal((B, F, 65)))
  harm = ddsp.synths.Harmonic(n_samples=N)
  noise = ddsp.synths.FilteredNoise(n_samples=N)
  for name, fn in (('harm', lambda: harm(amps, hd, f0)), ('noise', lambda: noise(mags))):
    if only and only != name: continue
    for _ in range(30): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
      for _ in range(20): fn()
      torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=512)
    for _ in range(100): fn()
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    res['%%s_b%%d' %% (name, B)] = round(sum(v[0] for v in bd.values()) / max(v[1] for v in bd.values()) * 1e3, 2)
if not only or only == 'loss':
  B, N = 32, 64000
  rng = np.random.default_rng(1)
  a = ddsp.core.tf_float32(rng.standard_normal((B, N)))
  t = ddsp.core.tf_float32(rng.standard_normal((B, N)))
  for sizes in ((2048,), (256,), (2048, 1024, 512, 256, 128, 64)):
    loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, mag_weight=1.0, logmag_weight=1.0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
      for _ in range(10): loss(t, a)
      torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=1024)
    for _ in range(50): loss(t, a)
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    res['loss_fwd_%%s' %% ('all' if len(sizes) > 1 else sizes[0])] = round(sum(v[0] for v in bd.values()) / 50 * 1e3, 1)
    if len(sizes) > 1:                      # value + gradient (the training path: ddsp_spectral_loss_value_and_grad_f32)
      at = a.clone().requires_grad_(True)
      def fb():
        at.grad = None
        loss(t, at).backward()
      for _ in range(20): fb()
      torch.cuda.synchronize()
      _lib.profile_begin(None, max_records=2048)
      for _ in range(50): fb()
      torch.cuda.synchronize()
      bd = _lib.profile_end()
      res['loss_fwdbwd_all'] = round(sum(v[0] for v in bd.values()) / 50 * 1e3, 1)
      res['loss_fwdbwd_kernels'] = {k: round(v[0] / 50 * 1e3, 1) for k, v in bd.items()}
print('AB ' + json.dumps(res))
'''

def main():
  args = [a for a in sys.argv[1:]]
  rounds, only = 3, ''
  if '--rounds' in args:
    i = args.index('--rounds'); rounds = int(args[i + 1]); del args[i:i + 2]
  if '--only' in args:
    i = args.index('--only'); only = args[i + 1]; del args[i:i + 2]
  names = args or ['base', 'new']
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
      num = [k for k in runs[n][0] if not isinstance(runs[n][0][k], dict)]
      print(json.dumps({'variant': n, 'median_us': {k: statistics.median(r[k] for r in runs[n]) for k in num},
                        'all': {k: [r[k] for r in runs[n]] for k in num},
                        'kernels': {k: runs[n][0][k] for k in runs[n][0] if isinstance(runs[n][0][k], dict)}}))

if __name__ == '__main__':
  main()
