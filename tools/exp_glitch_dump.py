"""Which intermediate of phase B differs when a launch of harm_table_kernel differs from the last one?  Needs a library built
with -DDDSP_EXP_DEBUG_DUMP (tools/build_variant_lib.sh): the interpolators then write eight floats per sample - theta, z, the
two accumulators of either table row, the envelope weight, the table offset - to the buffer named by $DDSP_EXP_DUMP_PTR.
    python tools/exp_glitch_dump.py tools/bin/libddsp_amd_<variant>.so [launches]"""
import os, sys, json
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, torch
from ddsp_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import ddsp_amd as ddsp
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 30
b, f, k, n = 32, 500, 100, 64000
rng = np.random.default_rng(9)
T = ddsp.core.tf_float32
amps, hd = T(rng.standard_normal((b, f, 1))), T(rng.standard_normal((b, f, k)))
f0 = T(70.0 + 0.5 * rng.standard_normal((b, f, 1)))
harm = ddsp.synths.Harmonic(n_samples=n)
dumps = [torch.zeros((b, n, 8), device='cuda') for _ in range(2)]
names = ['theta', 'z', 'acc0.x', 'acc0.y', 'acc1.x', 'acc1.y', 'w_next', 'tab_off']
def run(i):
  os.environ['DDSP_EXP_DUMP_PTR'] = str(dumps[i].data_ptr())
  out = harm(amps, hd, f0)
  torch.cuda.synchronize()
  return out
ref = run(0)
shown = 0
for it in range(n_launch):
  out = run(1)
  d = (out != ref)
  if not bool(d.any()):
    continue
  idx = d.nonzero().tolist()
  dd = (dumps[1] != dumps[0])
  fields = dd.any(dim=0).any(dim=0).tolist()
  print('launch %d: %d samples differ; fields that differ anywhere: %s; samples with a differing field: %d' % (
      it, len(idx), [nm for nm, x in zip(names, fields) if x], int(dd.any(dim=2).sum())))
  for r, s_ in idx[:6]:
    a, c = dumps[0][r, s_].tolist(), dumps[1][r, s_].tolist()
    print('  row %d sample %d (lane %d, frame %d): audio %.9g vs %.9g' % (r, s_, s_ % 64, s_ // (n // f), float(ref[r, s_]), float(out[r, s_])))
    for nm, x, y in zip(names, a, c):
      print('     %-8s %.9g  %.9g  %s' % (nm, x, y, '' if x == y else '<-- differs by %.3g' % (y - x)))
  # samples whose dump differs but whose audio does not, and the other way round
  only_dump = (dd.any(dim=2) & ~d).sum().item(); only_audio = (d & ~dd.any(dim=2)).sum().item()
  print('  dump differs / audio equal: %d   audio differs / dump equal: %d' % (only_dump, only_audio))
  shown += 1
  if shown >= 4: break
print('done, launches with a difference shown:', shown)
