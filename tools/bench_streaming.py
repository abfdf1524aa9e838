"""Latency of the streaming path: the VST model's per-frame call
(ddsp/training/inference.py:446-472) - core.streaming_harmonic_synthesis on two frames with the phase
carried in and out, FilteredNoise.get_signal on two frames, their sum - one call after the other the way a
real-time host drives it (batch 1, one hop of audio per call).  A latency path: what is reported is the time
per call against the real-time budget of one hop, not a throughput.

    python tools/bench_streaming.py [sample_rate] [hop] [n_harmonics] [n_noise] [calls]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import build
build.build()
SR = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
HOP = int(sys.argv[2]) if len(sys.argv) > 2 else 320          # vst.gin: 50 frames per second
K = int(sys.argv[3]) if len(sys.argv) > 3 else 60
M = int(sys.argv[4]) if len(sys.argv) > 4 else 65
CALLS = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
rng = np.random.default_rng(0)
f0 = ddsp.core.tf_float32(220.0 + rng.standard_normal((1, 2, 1)))
amps = ddsp.core.tf_float32(rng.uniform(0.1, 1.0, (1, 2, 1)))
hd = ddsp.core.tf_float32(rng.uniform(0.0, 1.0, (1, 2, K)))
mags = ddsp.core.tf_float32(rng.uniform(0.0, 1.0, (1, 2, M)))
noise = ddsp.synths.FilteredNoise(n_samples=HOP, window_size=0, scale_fn=None)
phase = torch.zeros((1, 1, 1), device='cuda')


def frame(phase):
  harm, phase = ddsp.core.streaming_harmonic_synthesis(f0, amps, hd, initial_phase=phase, n_samples=HOP,
                                                       sample_rate=SR, amp_resample_method='linear')
  return harm + noise.get_signal(mags), phase


for _ in range(200):
  out, phase = frame(phase)
torch.cuda.synchronize()
# (a) free-running: calls issued back to back, one synchronize at the end (the device-side cost per call)
t0 = time.perf_counter()
for _ in range(CALLS):
  out, phase = frame(phase)
torch.cuda.synchronize()
pipelined = (time.perf_counter() - t0) / CALLS
# (b) real-time use: the host needs the samples of every call before it issues the next
lat = []
for _ in range(CALLS):
  t1 = time.perf_counter()
  out, phase = frame(phase)
  out_host = out.cpu()
  lat.append(time.perf_counter() - t1)
lat = np.sort(np.array(lat))
budget = HOP / SR
print(json.dumps({
    'workload': 'VST frame call: streaming_harmonic_synthesis (2 frames, K=%d) + FilteredNoise.get_signal (M=%d) + sum, '
                'batch 1, hop %d @ %d Hz' % (K, M, HOP, SR),
    'us_per_call_pipelined': pipelined * 1e6,
    'us_per_call_with_readback': {'median': float(np.median(lat)) * 1e6, 'p99': float(lat[int(0.99 * len(lat))]) * 1e6,
                                  'max': float(lat[-1]) * 1e6},
    'real_time_budget_us': budget * 1e6, 'budget_used_median': float(np.median(lat)) / budget}))
