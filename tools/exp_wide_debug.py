"""Where the WIDE wavetable instance differs from the direct sum (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()

def case(b, f, hop, k, sr, f0c, seed=0, jitter=0.0):
  rng = np.random.default_rng(seed)
  n = f * hop
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = np.abs(f0c + jitter * rng.standard_normal((b, f, 1))).astype(np.float32)
  s1 = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  s2 = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr); s2.kernel = 'direct'
  outs = []
  for rep in range(3):
    outs.append(s1(amps, hd, f0).cpu().numpy())
  d = s2(amps, hd, f0).cpu().numpy()
  err = np.abs(outs[0] - d).reshape(b, f, hop)
  per_frame = err.max(axis=2)
  bad = np.argwhere(per_frame > 1e-4)
  print('B%d F%d hop%d K%d f0 %g: max err %.3g, bad frames %d of %d, repeatable %s' % (
      b, f, hop, k, f0c, err.max(), len(bad), b * f, all(np.array_equal(outs[0], o) for o in outs[1:])))
  if len(bad):
    print('   bad (clip, frame):', [tuple(x) for x in bad[:24]])
    c, j = bad[0]
    e = err[c, j]
    print('   first bad frame: samples with err > 1e-4:', np.flatnonzero(e > 1e-4)[:20], 'max at', int(e.argmax()), 'of', hop)
    full = s1(amps, hd, f0, return_outputs_dict=True)
    ref = s2(amps, hd, f0, return_outputs_dict=True)
    for key in ('amplitudes', 'harmonic_distribution'):
      print('   controls', key, float((full['controls'][key] - ref['controls'][key]).abs().max()))

case(1, 10, 64, 200, 48000, 60.0)
case(1, 10, 192, 200, 48000, 60.0)
case(1, 31, 64, 136, 48000, 60.0)
case(1, 31, 64, 192, 48000, 60.0)
case(1, 31, 64, 196, 48000, 60.0)
case(3, 45, 192, 199, 48000, 60.0)
case(2, 100, 64, 200, 48000, 100.0, jitter=1.0)
