"""The oracle against the REFERENCE'S OWN SOURCE on random inputs (TEST INFRASTRUCTURE; build container only: needs /root/reference).

tests/golden/*.npz pin oracle/ddsp_oracle.py to 36 fixtures made by executing the reference's unmodified ddsp/core.py, synths.py,
processors.py, effects.py, losses.py on the numpy stand-in for the TensorFlow ops they call (tests/golden/tf_numpy_shim.py).  This
tool widens that pin: the same reference code and the oracle's fp32 "faithful" mode run side by side on random shapes, flags and
regimes, and must agree - bit for bit where the oracle restates the reference op for op (resample, windows, controls, crops,
frame counts), within a few ulp of the output scale where summation order differs (FFT against direct sums).  It says nothing
about real TensorFlow (not installable here) and nothing about the kernels (tools/fuzz_parity.py does that on the GPU).

    python tools/fuzz_oracle_vs_reference.py [--seconds 120] [--seed 1]
"""
import argparse, json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
import tf_numpy_shim
tf_numpy_shim.install('/root/reference')
from ddsp import core, effects, losses, processors, synths      # noqa: E402  (the reference's files)
from oracle import ddsp_oracle as O                              # noqa: E402


def a(x):
  return np.asarray(x)


def close(got, ref, rtol_of_scale, what):
  got, ref = a(got), a(ref)
  assert got.shape == ref.shape, (what, 'shape', got.shape, ref.shape)
  if ref.size == 0:
    return 0.0
  scale = max(float(np.abs(ref).max()), 1e-30)
  err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max())
  assert err <= rtol_of_scale * scale, (what, err, rtol_of_scale * scale)
  return err / scale


def case_resample(rng):
  b, f, c = int(rng.integers(1, 3)), int(rng.integers(1, 40)), int(rng.integers(1, 9))
  method = str(rng.choice(['nearest', 'linear', 'cubic', 'window']))
  add_endpoint = bool(rng.integers(0, 2))
  if method == 'window':
    hops = f if add_endpoint else f - 1
    if hops < 1:
      return None
    n = hops * int(rng.integers(2, 80))
    if f + int(add_endpoint) >= n:              # the reference's own ValueError (core.py:682-685); its text is pinned in tests/test_host_api.py
      return None
  else:
    n = int(rng.integers(f, f * 100 + 1)) if rng.random() < 0.8 else int(rng.integers(1, f + 1))      # (downsampling too: core_test.py:268-290)
  x = rng.standard_normal((b, f, c)).astype(np.float32)
  what = dict(case='resample', b=b, f=f, c=c, n=n, method=method, add_endpoint=add_endpoint)
  ref = a(core.resample(x, n, method=method, add_endpoint=add_endpoint))
  got = O.resample(x, n, method=method, add_endpoint=add_endpoint, dtype=np.float32)
  # 'window': the oracle restates overlap_and_add literally; everything else is index arithmetic and one interpolation
  return what, close(got, ref, 0.0 if method in ('nearest', 'linear', 'cubic') else 4e-7, what)


def case_harmonic(rng):
  b, f = int(rng.integers(1, 3)), int(rng.integers(1, 25))
  hop = int(rng.choice([16, 37, 64, 100, 192]))
  k = int(rng.choice([1, 3, 12, 40, 100]))
  sr = int(rng.choice([8000, 16000, 48000]))
  method = str(rng.choice(['window', 'linear', 'nearest', 'cubic']))
  n = f * hop if method == 'window' else int(rng.integers(f, f * hop + 1))
  angular = bool(rng.integers(0, 2))
  scale = bool(rng.integers(0, 4))
  normalize = bool(rng.integers(0, 4))
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = rng.uniform(30.0, sr / 3.0, (b, f, 1)).astype(np.float32)
  if not scale:
    amps, hd = np.abs(amps) + 0.1, np.abs(hd) + 0.01
  what = dict(case='harmonic', b=b, f=f, n=n, k=k, sr=sr, method=method, angular=angular, scale=scale, normalize=normalize)
  synth = synths.Harmonic(n_samples=n, sample_rate=sr, scale_fn=core.exp_sigmoid if scale else None,
                          normalize_below_nyquist=normalize, amp_resample_method=method, use_angular_cumsum=angular)
  out = synth(amps, hd, f0, return_outputs_dict=True)
  sf = O.exp_sigmoid if scale else None
  ctl = O.harmonic_get_controls(amps, hd, f0, sr, scale_fn=sf, normalize_below_nyquist=normalize, dtype=np.float32)
  r1 = close(ctl['amplitudes'], out['controls']['amplitudes'], 0.0, what)
  r2 = close(ctl['harmonic_distribution'], out['controls']['harmonic_distribution'], 0.0, what)
  sig = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=sr, scale_fn=sf, normalize_below_nyquist=normalize,
                   amp_resample_method=method, use_angular_cumsum=angular, dtype=np.float32)
  ref = a(out['signal'])
  # same ops in the same order: the sum over harmonics is the one place numpy may associate differently (pairwise sums)
  assert sig.shape == ref.shape, (what, 'shape')
  err = float(np.abs(sig.astype(np.float64) - ref).max())
  tol = 4e-6 * max(1.0, float(np.abs(a(out['controls']['amplitudes'])).max()))
  assert err <= tol, (what, 'signal', err, tol)
  return what, max(r1, r2, err / tol)


def case_noise(rng):
  b, f = int(rng.integers(1, 3)), int(rng.integers(1, 30))
  m = int(rng.choice([3, 9, 17, 33, 65, 100]))
  l0 = 2 * (m - 1)
  ws = int(rng.choice([0, 257, l0 + 3, max(3, l0 // 2 + 1), max(2, l0 // 3)]))
  fs = int(rng.choice([8, 20, 64, 100, 192]))
  n = f * fs - int(rng.integers(0, fs))
  if n < 1:
    return None
  scale = bool(rng.integers(0, 4))
  mags = rng.standard_normal((b, f, m)).astype(np.float32)
  if not scale:
    mags = np.abs(mags)
  noise = rng.uniform(-1.0, 1.0, (b, n)).astype(np.float32)
  what = dict(case='noise', b=b, f=f, m=m, ws=ws, n=n, scale=scale)
  synth = synths.FilteredNoise(n_samples=n, window_size=ws, scale_fn=core.exp_sigmoid if scale else None)
  controls = synth.get_controls(mags)
  ir = a(core.frequency_impulse_response(controls['magnitudes'], window_size=ws))
  ref = a(core.frequency_filter(noise, controls['magnitudes'], window_size=ws))
  sf = O.exp_sigmoid if scale else None
  octl = O.filtered_noise_get_controls(mags, scale_fn=sf, dtype=np.float32)
  r0 = close(octl['magnitudes'], controls['magnitudes'], 0.0, what)
  oir = O.frequency_impulse_response(octl['magnitudes'], window_size=ws, dtype=np.float32)
  r1 = close(oir, ir, 3e-6, what)                 # (numpy's FFT in double against the shim's: rounding of the irfft only)
  got = O.filtered_noise(mags, noise, ws, sf, dtype=np.float32)
  return what, max(r0, r1, close(got, ref, 2e-5, what))


def case_fft_convolve(rng):
  b, f = int(rng.integers(1, 3)), int(rng.integers(1, 20))
  fs = int(rng.choice([4, 16, 50, 64]))
  n = f * fs - int(rng.integers(0, fs))
  l = int(rng.choice([1, 2, 3, 17, 64, 129]))
  if n < 1:
    return None
  padding = str(rng.choice(['same', 'valid']))
  dc = int(rng.choice([-1, -1, 0, 2, l // 2]))
  ir_b = int(rng.choice([1, b]))
  x = rng.standard_normal((b, n)).astype(np.float32)
  h = rng.standard_normal((ir_b, f, l)).astype(np.float32)
  what = dict(case='fft_convolve', b=b, f=f, n=n, l=l, padding=padding, dc=dc, ir_b=ir_b)
  try:
    ref = a(core.fft_convolve(x, h, padding=padding, delay_compensation=dc))
  except ValueError as e:                        # (one sample, one tap: an FFT of size 1 - numpy's irfft refuses n = 0 output points)
    if 'FFT data points' in str(e):
      return None
    raise
  got = O.fft_convolve(x, h, padding=padding, delay_compensation=dc, dtype=np.float32)
  return what, close(got, ref, 2e-5, what)


def case_reverb(rng):
  b = int(rng.integers(1, 4))
  n, l = int(rng.integers(1, 3000)), int(rng.integers(1, 2000))
  ir_b = int(rng.choice([1, b])) if b > 1 else 1
  add_dry = bool(rng.integers(0, 2))
  x = rng.standard_normal((b, n)).astype(np.float32)
  h = (rng.standard_normal((b, l)) * np.exp(-np.arange(l) / (0.3 * l + 1))).astype(np.float32)
  what = dict(case='reverb', b=b, n=n, l=l, add_dry=add_dry)
  ref = a(effects.Reverb(add_dry=add_dry)(x, h))
  got = O.reverb(x, h, add_dry=add_dry, dtype=np.float32)
  return what, close(got, ref, 3e-5, what)


def case_loss(rng):
  b, n = int(rng.integers(1, 4)), int(rng.choice([64, 500, 3000, int(rng.integers(64, 5000))]))
  all_sizes = [2048, 1024, 512, 256, 128, 64, 1536, 768, 384, 192, 96, 48]     # (3 * 2**k: vst_48k.gin's kind)
  sizes = tuple(int(s) for s in rng.permutation(all_sizes)[:int(rng.integers(1, 5))])
  kw = dict(mag_weight=float(rng.choice([1.0, 0.0, 0.5])), logmag_weight=float(rng.choice([1.0, 0.0, 0.5])),
            delta_time_weight=float(rng.choice([0.0, 0.0, 1.0])), delta_freq_weight=float(rng.choice([0.0, 0.0, 1.0])),
            cumsum_freq_weight=float(rng.choice([0.0, 0.0, 1.0])), loudness_weight=float(rng.choice([0.0, 0.0, 1.0])))
  loss_type = str(rng.choice(['L1', 'L1', 'L2', 'COSINE']))
  t = (0.3 * rng.standard_normal((b, n))).astype(np.float32)
  x = (0.8 * t + 0.05 * rng.standard_normal((b, n))).astype(np.float32)
  what = dict(case='loss', b=b, n=n, sizes=sizes, loss_type=loss_type, **kw)
  ref = float(a(losses.SpectralLoss(fft_sizes=sizes, loss_type=loss_type, **kw)(t, x)))
  got = float(O.spectral_loss(t, x, sizes, loss_type=loss_type, dtype=np.float32, **kw))
  if np.isnan(ref) and np.isnan(got):       # (one frame and a delta-time term: the mean of an empty difference, in both)
    return what, 0.0
  assert abs(got - ref) <= 2e-5 * max(abs(ref), 1e-12), (what, got, ref)
  return what, abs(got - ref) / max(abs(ref), 1e-12) / 2e-5


def case_oscillator_bank(rng):
  b, n, k = int(rng.integers(1, 3)), int(rng.integers(1, 2500)), int(rng.choice([1, 4, 33, 100]))
  sr = int(rng.choice([8000, 16000, 48000]))
  f = rng.uniform(0.0, sr * 0.6, (b, n, k)).astype(np.float32)
  amp = rng.standard_normal((b, n, k)).astype(np.float32)
  angular, sum_s = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
  what = dict(case='oscillator_bank', b=b, n=n, k=k, sr=sr, angular=angular, sum_sinusoids=sum_s)
  ref = a(core.oscillator_bank(f, amp, sample_rate=sr, sum_sinusoids=sum_s, use_angular_cumsum=angular))
  got = O.oscillator_bank(f, amp, sample_rate=sr, sum_sinusoids=sum_s, use_angular_cumsum=angular)
  return what, close(got, ref, 4e-6 if sum_s else 0.0, what)


CASES = dict(resample=case_resample, harmonic=case_harmonic, noise=case_noise, fft_convolve=case_fft_convolve, reverb=case_reverb,
             loss=case_loss, oscillator_bank=case_oscillator_bank)

if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--seconds', type=float, default=120.0)
  ap.add_argument('--seed', type=int, default=1)
  ap.add_argument('--only', default='')
  args = ap.parse_args()
  names = [s for s in args.only.split(',') if s] or list(CASES)
  counts = {k: 0 for k in names}; worst = {k: 0.0 for k in names}; failures = []
  t_end, i = time.time() + args.seconds, 0
  while time.time() < t_end:
    name = names[i % len(names)]; seed = args.seed * 1000003 + i; i += 1
    try:
      r = CASES[name](np.random.default_rng(seed))
      if r is None:
        continue
      counts[name] += 1; worst[name] = max(worst[name], r[1])
    except Exception as e:
      failures.append({'case': name, 'seed': seed, 'error': repr(e)[:500]})
      print('FAIL', json.dumps(failures[-1]), traceback.format_exc()[-300:].replace('\n', ' | '), flush=True)
  print('SUMMARY', json.dumps({'cases_run': counts, 'worst_error_over_tolerance': {k: round(v, 3) for k, v in worst.items()},
                               'failures': len(failures), 'seed': args.seed}))
