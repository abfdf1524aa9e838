"""Round 6 A/B on one box: Harmonic on harm_wt4_kernel (four independent blocks of four wavefronts per CU, the default) against
harm_table_kernel (one block of sixteen, Harmonic.kernel = 'table16'): per-launch time from dispatch events, bit equality of the
two at the headline regime, errors against each other at other f0, and the two-stream step (Harmonic + FilteredNoise).

    python tools/exp_w4.py [--batches 32,128] [--f0 70,200,333]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddsp_amd import _lib
import ddsp_amd as ddsp


def settle(fn, secs=0.05):
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize()


def per_launch_us(fn, n=100):
  _lib.profile_begin(None, max_records=4096)
  for _ in range(n): fn()
  torch.cuda.synchronize()
  bd = _lib.profile_end()
  return {k: round(v[0] / v[1] * 1e3, 2) for k, v in bd.items()}


def wall_us(fn, n=200):
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record()
  torch.cuda.synchronize()
  return round(e0.elapsed_time(e1) / n * 1e3, 2)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batches', default='32,128')
  ap.add_argument('--f0', default='70,200,333')
  ap.add_argument('--noise-bits', type=int, default=23)
  a = ap.parse_args()
  F, K, N = 1000, 100, 64000
  out = {}
  for B in [int(x) for x in a.batches.split(',')]:
    for f0c in [float(x) for x in a.f0.split(',')]:
      rng = np.random.default_rng(0)
      amps = ddsp.core.tf_float32(rng.standard_normal((B, F, 1)))
      hd = ddsp.core.tf_float32(rng.standard_normal((B, F, K)))
      f0 = ddsp.core.tf_float32(f0c + rng.standard_normal((B, F, 1)))
      mags = ddsp.core.tf_float32(rng.standard_normal((B, F, 65)))
      new, old = ddsp.synths.Harmonic(n_samples=N), ddsp.synths.Harmonic(n_samples=N)
      old.kernel = 'table16'
      noise = ddsp.synths.FilteredNoise(n_samples=N, window_size=0, noise_bits=a.noise_bits)
      y_new, y_old = new(amps, hd, f0), old(amps, hd, f0)
      torch.cuda.synchronize()
      rec = {'max_abs_diff_new_vs_old': float((y_new - y_old).abs().max()), 'bit_equal': bool(torch.equal(y_new, y_old)),
             'finite': bool(torch.isfinite(y_new).all()), 'max_abs': float(y_new.abs().max())}
      for name, h in (('new', new), ('old', old)):
        fn = lambda: h(amps, hd, f0)
        settle(fn)
        rec[name + '_launch_us'] = per_launch_us(fn)
        rec[name + '_wall_us'] = wall_us(fn)
      fnn = lambda: noise(mags)
      settle(fnn)
      rec['noise_launch_us'] = per_launch_us(fnn)
      s_h, s_z, s_0 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
      for name, h in (('new', new), ('old', old)):
        def step():
          torch.cuda.set_stream(s_h); h(amps, hd, f0)
          torch.cuda.set_stream(s_z); noise(mags)
          torch.cuda.set_stream(s_0)
        settle(step)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n): step()
        torch.cuda.synchronize()
        rec[name + '_two_stream_step_us'] = round((time.perf_counter() - t0) / n * 1e6, 2)
        def step1():
          h(amps, hd, f0); noise(mags)
        settle(step1)
        rec[name + '_one_stream_step_us'] = wall_us(step1)
      out['b%d_f0_%g' % (B, f0c)] = rec
      print('W4', 'b%d_f0_%g' % (B, f0c), json.dumps(rec), flush=True)
  os.makedirs('gpurun_out', exist_ok=True)
  json.dump(out, open('gpurun_out/exp_w4.json', 'w'), indent=1)


if __name__ == '__main__':
  main()
