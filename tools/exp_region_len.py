"""Why are 20-step regions slower per step than 1000-step regions?  Per-kernel dispatch durations and region time per step
as a function of the region length (one stream; every region bracketed by synchronize), batch 128.

    python tools/exp_region_len.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ddsp_amd as ddsp
from ddsp_amd import _lib
B, F, K, N = 128, 1000, 100, 64000
rng = np.random.default_rng(0)
amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
f0 = ddsp.core.tf_float32(70 + rng.standard_normal((B, F, 1)))
mags = ddsp.core.tf_float32(rng.standard_normal((B, F, 65)))
harm = ddsp.synths.Harmonic(n_samples=N)
noise = ddsp.synths.FilteredNoise(n_samples=N)
def step():
  harm(amps, hd, f0); noise(mags)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
  for _ in range(20): step()
  torch.cuda.synchronize()
for steps in (1, 2, 5, 10, 20, 50, 200, 1000):
  reps = max(3, 400 // steps)
  ev, tot = [], {}
  for r in range(reps):
    torch.cuda.synchronize()
    _lib.profile_begin(None, max_records=2 * steps + 16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): step()
    e1.record()
    torch.cuda.synchronize()
    bd = _lib.profile_end()
    ev.append(e0.elapsed_time(e1) * 1e3 / steps)
    for k, v in bd.items():
      tot.setdefault(k, []).append(v[0] / v[1] * 1e3)
  print(json.dumps({'steps_per_region': steps, 'regions': reps, 'us_per_step_median': round(float(np.median(ev)), 2),
                    'kernel_us_median': {k: round(float(np.median(v)), 2) for k, v in tot.items()}}))
# the same without per-launch events (what bench.py's regions are): region time per step, host time of the launch loop
for steps in (5, 10, 20, 50, 200, 1000):
  reps = max(5, 2000 // steps)
  ev, host = [], []
  for r in range(reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps): step()
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    ev.append(e0.elapsed_time(e1) * 1e3 / steps)
    host.append((t1 - t0) * 1e6 / steps)
  print(json.dumps({'plain_regions_steps': steps, 'regions': reps, 'us_per_step_median': round(float(np.median(ev)), 2),
                    'us_per_step_min': round(float(np.min(ev)), 2), 'host_launch_loop_us_per_step': round(float(np.median(host)), 2)}))
# two free-running streams the way bench.py brackets a region: both streams start behind e0 (recorded on the base stream), the base
# stream waits for both before e1; per region length: per-step median / min, and the same with a fixed launch-ahead of the
# first kernels (`prime`): one Harmonic + one FilteredNoise launch ahead of e0 on their streams (untimed), so that the GPU is not idle
# when the timed launches arrive
s0 = torch.cuda.current_stream()
sh, sz = torch.cuda.Stream(), torch.cuda.Stream()
def step2():
  torch.cuda.set_stream(sh); harm(amps, hd, f0)
  torch.cuda.set_stream(sz); noise(mags)
  torch.cuda.set_stream(s0)
for steps in (5, 20, 50, 200, 1000):
  reps = max(5, 2000 // steps)
  ev = []
  for r in range(reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0); sh.wait_event(e0); sz.wait_event(e0)
    for _ in range(steps): step2()
    eh, ez = torch.cuda.Event(), torch.cuda.Event()
    eh.record(sh); ez.record(sz); s0.wait_event(eh); s0.wait_event(ez)
    e1.record(s0)
    torch.cuda.synchronize()
    ev.append(e0.elapsed_time(e1) * 1e3 / steps)
  print(json.dumps({'two_stream_regions_steps': steps, 'regions': reps, 'us_per_step_median': round(float(np.median(ev)), 2),
                    'us_per_step_min': round(float(np.min(ev)), 2), 'us_per_step_p90': round(float(np.percentile(ev, 90)), 2)}))
