"""Host-side cost of a Processor call: time K calls at B=1 (GPU work is tiny) without syncing each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
rng = np.random.default_rng(0)
B, F, K, M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 1000, 100, 65, 64000
dev = {k: ddsp.core.tf_float32(v) for k, v in dict(
    a=rng.standard_normal((B, F, 1)), hd=rng.standard_normal((B, F, K)),
    f0=70 + rng.standard_normal((B, F, 1)), m=rng.standard_normal((B, F, M))).items()}
harm, noise = ddsp.synths.Harmonic(), ddsp.synths.FilteredNoise(window_size=0)
def loop(k):
  for _ in range(k):
    harm(dev['a'], dev['hd'], dev['f0']); noise(dev['m'])
loop(50); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(2000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('enqueue time per step (2 calls): %.1f us ; incl. final sync %.1f us' % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
s1, s2, s0 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
def loop2(k):
  for _ in range(k):
    torch.cuda.set_stream(s1); harm(dev['a'], dev['hd'], dev['f0'])
    torch.cuda.set_stream(s2); noise(dev['m'])
    torch.cuda.set_stream(s0)
def loop3(k):
  for _ in range(k):
    with torch.cuda.stream(s1): harm(dev['a'], dev['hd'], dev['f0'])
    with torch.cuda.stream(s2): noise(dev['m'])
for nm, fn in (('set_stream', loop2), ('with stream', loop3)):
  fn(50); torch.cuda.synchronize()
  t0 = time.perf_counter(); fn(2000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print('%s: enqueue per step %.1f us ; incl. final sync %.1f us' % (nm, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
if len(sys.argv) > 1: sys.exit(0)
pr = cProfile.Profile(); pr.enable(); loop(2000); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
