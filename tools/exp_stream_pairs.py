"""The headline step (Harmonic + FilteredNoise, batch 128, 2^23-level noise) on 1, 2, 4 and 6 free-running HIP streams: with 2 streams
(bench.py's mode) kernel k + 1 of a synth cannot start before the LAST block of its kernel k has left; with 2 P streams step s runs on
pair s % P, so the persistent blocks of the next step's kernels may take the CUs the current step's early finishers leave.

    python tools/exp_stream_pairs.py [steps=1000] [batch=128]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
F, K, N = 1000, 100, 64000
rng = np.random.default_rng(0)
T = ddsp.core.tf_float32
amps, hd = T(rng.standard_normal((B, F, 1))), T(rng.standard_normal((B, F, K)))
f0, mags = T(70 + rng.standard_normal((B, F, 1))), T(rng.standard_normal((B, F, 65)))
harm = ddsp.synths.Harmonic(n_samples=N)
noise = ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
base = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(8)]


def region(n_streams, k):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record(base)
  used = pool[:n_streams] if n_streams > 1 else []
  for s in used: s.wait_event(e0)
  keep = []
  for i in range(k):
    if n_streams == 1:
      h = harm(amps, hd, f0); z = noise(mags)
    else:
      pair = (2 * i) % n_streams
      torch.cuda.set_stream(used[pair]); h = harm(amps, hd, f0)
      torch.cuda.set_stream(used[pair + 1]); z = noise(mags)
    keep.append((h, z))
    if len(keep) > 16: keep.pop(0)            # (outputs stay alive until their stream's later launches are queued: the caching allocator is per stream)
  torch.cuda.set_stream(base)
  for s in used:
    ev = torch.cuda.Event(); ev.record(s); base.wait_event(ev)
  e1.record(base)
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / k


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0: region(2, 50)
res = {}
for rep in range(3):
  for n in (1, 2, 4, 6, 8):
    res.setdefault(str(n), []).append(round(region(n, steps), 2))
print('STREAMS ' + json.dumps({'batch': B, 'steps': steps, 'us_per_step': res}))
