"""The headline step (Harmonic + FilteredNoise, batch 128, two streams) issued K at a time as ONE HIP graph against the same K
steps launched one by one: what the launches and the stream fork / join of the eager loop cost per step.

    python tools/exp_graph_step.py [batch] [K] [regions]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 60
F, KH, M, N = 1000, 100, 65, 64000
rng = np.random.default_rng(0)
T = ddsp.core.tf_float32
amps, hd = T(rng.standard_normal((B, F, 1))), T(rng.standard_normal((B, F, KH)))
f0, mags = T(70.0 + rng.standard_normal((B, F, 1))), T(rng.standard_normal((B, F, M)))
harm = ddsp.synths.Harmonic(n_samples=N)
fnoise = ddsp.synths.FilteredNoise(n_samples=N, window_size=0)
ddsp.core.prepare(KH, M, 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def k_steps():
  """K steps on two free-running streams (bench.py's loop): s2 forks off s1 and joins it at the end."""
  s2.wait_stream(s1)
  for _ in range(K):
    with torch.cuda.stream(s1):
      a = harm(amps, hd, f0)
    with torch.cuda.stream(s2):
      z = fnoise(mags)
  s1.wait_stream(s2)
  return a, z


def timed(fn):
  ts = []
  for _ in range(R):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
      e0.record(s1)
      fn()
      e1.record(s1)
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / K * 1e3)
  ts = np.sort(np.array(ts))
  return float(np.median(ts)), float(ts[0])


with torch.cuda.stream(s1):
  for _ in range(5):
    k_steps()
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
  with torch.cuda.stream(s1):
    k_steps()
  torch.cuda.synchronize()
eager = timed(lambda: k_steps())
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s1):
  k_steps()
torch.cuda.synchronize()
for _ in range(5):
  graph.replay()
torch.cuda.synchronize()


def replay():
  graph.replay()          # (on the current stream = s1 inside timed())


replayed = timed(replay)
eager2 = timed(lambda: k_steps())
print(json.dumps({'batch': B, 'steps_per_region': K, 'regions': R, 'us_per_step_eager_median_min': eager, 'us_per_step_graph_median_min': replayed,
                  'us_per_step_eager_again': eager2}))
