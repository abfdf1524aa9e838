"""Run-to-run and schedule determinism of the default kernels, stressed in ONE process.

The round-3 loop (tools/loop_bit_equality.sh) re-ran two pytest cases per iteration: a python start and ~10 s of GPU box
per sample of the race it was looking for.  Here every case keeps its inputs and its first result in HBM and repeats the
launch `--iters` times, comparing BITS on the device (`(a != b).sum()`), with the other synth kernel running beside it on a
second stream as in bench.py (two persistent kernels competing for the CUs is what changes the timing from launch to
launch), and re-checks rows run alone / in a sub-batch against the rows of the full batch every few iterations (another
cut into chunks, other template instances: harmonic_table.hip's independence claim).  A mismatch is reported with where it
is - row, sample, frame, chunk-tick arithmetic is left to the reader of the log - and how large.

    python tools/stress_determinism.py [--lib tools/bin/libddsp_amd_x.so] [--iters 300] [--cases a,b,..] [--out log.jsonl]

Exit status 1 if anything differed.  Batch rows are independent and the op chain deterministic in the reference
(ddsp/core.py:912-962); this is the property test of that.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main(argv=None):
  """Returns the summary record; `launches_with_a_difference` == 0 is the pass criterion."""
  ap = argparse.ArgumentParser()
  ap.add_argument('--lib', default='')
  ap.add_argument('--iters', type=int, default=300)
  ap.add_argument('--cases', default='')
  ap.add_argument('--out', default='')
  ap.add_argument('--label', default='')
  args = ap.parse_args(argv)

  import numpy as np
  import torch
  from ddsp_amd import _lib
  if args.lib:
    if _lib._lib is not None and os.path.abspath(args.lib) != _lib.LIB_PATH:
      raise RuntimeError('another library is already loaded in this process')
    _lib.LIB_PATH = os.path.abspath(args.lib)
  import ddsp_amd as ddsp
  _lib.load()
  dev = 'cuda'
  T = ddsp.core.tf_float32

  def controls(b, f, k, f0c, jitter, seed, m=65):
    rng = np.random.default_rng(seed)
    amps = T(rng.standard_normal((b, f, 1)))
    hd = T(rng.standard_normal((b, f, k)))
    if isinstance(jitter, str) and jitter == 'vibrato':      # a note with vibrato: 6 Hz at 5.5 Hz over the clip, per-row phase
      t = np.arange(f)[None, :, None] / 250.0
      f0 = f0c + 6.0 * np.sin(2 * np.pi * 5.5 * t + rng.uniform(0, 6.28, (b, 1, 1)))
    else:
      f0 = f0c + jitter * rng.standard_normal((b, f, 1))
    mags = T(rng.standard_normal((b, f, m)))
    return amps, hd, T(f0), mags

  side = torch.cuda.Stream()
  log = []

  def report(case, it, what, a, b, hop):
    d = (a != b)
    n = int(d.sum().item())
    if n == 0:
      return 0
    idx = d.nonzero()[:12].tolist()
    diff = (a.double() - b.double()).abs()
    rec = {'case': case, 'iter': it, 'what': what, 'differing': n, 'of': a.numel(), 'max_abs': float(diff.max().item()),
           'first': [{'row': r, 'sample': s, 'frame': s // hop, 'a': float(a[r, s]), 'b': float(b[r, s])} for r, s in idx]}
    rows = d.any(dim=1).nonzero().flatten().tolist()
    rec['rows'] = rows[:16]
    if rows:
      cols = d[rows[0]].nonzero().flatten()
      rec['row0_span'] = [int(cols.min()), int(cols.max()), int(cols.numel())]
    log.append(rec)
    print('MISMATCH ' + json.dumps(rec), flush=True)
    return n

  def stress(case, fn, side_fn, hop, sub_checks):
    """fn() -> [B, N]; side_fn(): the other kernel, launched on the side stream before every fn()."""
    t0 = time.perf_counter()
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(args.iters):
      if side_fn is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          side_fn()
          if it % 3 == 0:
            side_fn()
      out = fn()
      bad += 1 if report(case, it, 'run-to-run', out, ref, hop) else 0
      if sub_checks and it % 4 == 0:
        for what, sl, run in sub_checks(it):
          bad += 1 if report(case, it, what, run(), ref[sl], hop) else 0
    torch.cuda.synchronize()
    rec = {'case': case, 'iters': args.iters, 'launches_with_a_difference': bad, 'seconds': round(time.perf_counter() - t0, 2)}
    log.append(rec)
    print('CASE ' + json.dumps(rec), flush=True)
    return bad

  rng = np.random.default_rng(99)
  want = set(c for c in args.cases.split(',') if c)
  total = 0

  def harm_case(name, b, f, k, n, sr, f0c, jitter, seed, method='window'):
    nonlocal total
    if want and name not in want:
      return
    amps, hd, f0, mags = controls(b, f, k, f0c, jitter, seed)
    harm = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)
    noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)

    def subs(it):
      r = int(rng.integers(0, b))
      out = [('row %d alone' % r, slice(r, r + 1), lambda: harm(amps[r:r + 1], hd[r:r + 1], f0[r:r + 1]))]
      if b >= 64:
        q = int(rng.integers(0, b // 32))
        s = slice(32 * q, 32 * q + 32)
        out.append(('rows %d..%d as a batch of 32' % (s.start, s.stop - 1), s, lambda: harm(amps[s], hd[s], f0[s])))
      elif b >= 16:
        h = int(rng.integers(1, b - 1))
        out.append(('rows %d.. as a batch of %d' % (h, b - h), slice(h, b), lambda: harm(amps[h:], hd[h:], f0[h:])))
      return out
    total += stress(name, lambda: harm(amps, hd, f0), lambda: noise(mags), n // f, subs)

  #             name                    B    F     K    N       sr     f0   jitter
  harm_case('harm_b128_f0_70',          128, 1000, 100, 64000,  16000, 70.0, 1.0, 21)
  harm_case('harm_b128_f0_200',         128, 1000, 100, 64000,  16000, 200.0, 1.0, 22)
  harm_case('harm_b128_f0_333',         128, 1000, 100, 64000,  16000, 333.0, 1.0, 23)
  harm_case('harm_b128_vibrato_220',    128, 1000, 100, 64000,  16000, 220.0, 'vibrato', 24)
  harm_case('harm_b32_f0_70',           32,  1000, 100, 64000,  16000, 70.0, 1.0, 5)
  harm_case('harm_b32_k60_f0_440',      32,  1000, 60,  64000,  16000, 440.0, 3.0, 6)
  harm_case('harm_b32_k128_hop128',     32,  500,  128, 64000,  16000, 55.0, 0.5, 7)
  harm_case('harm_config5_b32',         32,  2500, 200, 480000, 48000, 100.0, 1.0, 31, 'linear')
  harm_case('harm_config5_b8_f0_120',   8,   2500, 200, 480000, 48000, 120.0, 1.0, 32, 'linear')
  # which part of config 5 matters: the 129 .. 200-harmonic instances on frames of one tile; config 5's frames with 100 harmonics
  harm_case('harm_b32_k160_hop64',      32,  1000, 160, 64000,  16000, 45.0, 0.5, 33)
  harm_case('harm_b8_k136_hop192',      8,   2500, 136, 480000, 48000, 120.0, 1.0, 34, 'linear')
  harm_case('harm_b8_k100_hop192',      8,   2500, 100, 480000, 48000, 120.0, 1.0, 35, 'linear')
  harm_case('harm_b8_k128_hop192',      8,   2500, 128, 480000, 48000, 120.0, 1.0, 36, 'linear')
  harm_case('harm_b32_k128_hop64',      32,  1000, 128, 64000,  16000, 55.0, 0.5, 8)
  harm_case('harm_b32_k100_hop128',     32,  500,  100, 64000,  16000, 70.0, 0.5, 9)
  harm_case('harm_b32_k64_hop64',       32,  1000, 64,  64000,  16000, 110.0, 0.5, 10)
  harm_case('harm_b8_k200_hop192_window', 8, 2500, 200, 480000, 48000, 120.0, 1.0, 37, 'window')

  # the controls dict beside the audio (phase A's registers written out): [audio | harmonic_distribution | amplitudes] per row
  if not want or 'harm_config5_b8_controls' in want:
    b, f, k, n = 8, 2500, 200, 480000
    amps, hd, f0, mags = controls(b, f, k, 120.0, 1.0, 38)
    harm = ddsp.synths.Harmonic(n_samples=n, sample_rate=48000, amp_resample_method='linear')
    noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)

    def with_controls():
      o = harm(amps, hd, f0, return_outputs_dict=True)
      c = o['controls']
      return torch.cat([o['signal'], c['harmonic_distribution'].reshape(b, -1), c['amplitudes'].reshape(b, -1)], dim=1)
    total += stress('harm_config5_b8_controls', with_controls, lambda: noise(mags), 192, None)

  # FilteredNoise: supplied noise (the parity entry) and generated noise with the call counter pinned
  for name, b in (('noise_b128', 128), ('noise_b32', 32)):
    if want and name not in want:
      continue
    amps, hd, f0, mags = controls(b, 1000, 100, 70.0, 1.0, 40 + b)
    z = T(np.random.default_rng(41).uniform(-1, 1, (b, 64000)))
    fn = ddsp.synths.FilteredNoise(window_size=0, seed=7)
    harm = ddsp.synths.Harmonic()

    def gen():
      fn._calls = 0
      return fn(mags)

    def subs_n(it, b=b, fn=fn, mags=mags, z=z):
      r = int(rng.integers(0, b))
      return [('row %d alone, supplied noise' % r, slice(r, r + 1), lambda: fn(mags[r:r + 1], noise=z[r:r + 1]))]
    total += stress(name + '_supplied', lambda: fn(mags, noise=z), lambda: harm(amps, hd, f0), 64, subs_n)
    total += stress(name + '_generated', gen, lambda: harm(amps, hd, f0), 64, None)

  # round 4's other kernels: the general FilteredNoise path (100 bands: one launch, taps designed per tile; 256 bands: two
  # launches), and the two backward passes (gradients of a fixed upstream gradient, bits compared like the audio)
  for name, b, m in (('noise_general_m100_b32', 32, 100), ('noise_general_m256_b8', 8, 256)):
    if want and name not in want:
      continue
    amps, hd, f0, mags = controls(b, 1000, 100, 70.0, 1.0, 60 + m, m=m)
    z = T(np.random.default_rng(61).uniform(-1, 1, (b, 64000)))
    fn = ddsp.synths.FilteredNoise(window_size=0, seed=9)
    harm = ddsp.synths.Harmonic()

    def gen(fn=fn, mags=mags):
      fn._calls = 0
      return fn(mags)

    def subs_g(it, b=b, fn=fn, mags=mags, z=z):
      r = int(rng.integers(0, b))
      return [('row %d alone, supplied noise' % r, slice(r, r + 1), lambda: fn(mags[r:r + 1], noise=z[r:r + 1]))]
    total += stress(name + '_supplied', lambda fn=fn, mags=mags, z=z: fn(mags, noise=z), lambda: harm(amps, hd, f0), 64, subs_g)
    total += stress(name + '_generated', gen, lambda: harm(amps, hd, f0), 64, None)

  if not want or 'backward_b32' in want:
    b = 32
    amps, hd, f0, mags = controls(b, 1000, 100, 200.0, 1.0, 71)
    g = T(np.random.default_rng(72).standard_normal((b, 64000)))
    harm = ddsp.synths.Harmonic()
    noise = ddsp.synths.FilteredNoise(window_size=0, seed=4)

    def grads():
      a_ = amps.detach().requires_grad_(True)
      h_ = hd.detach().requires_grad_(True)
      m_ = mags.detach().requires_grad_(True)
      noise._calls = 0
      (harm.call_add(a_, h_, f0, noise(m_))).backward(g)
      return torch.cat([a_.grad.reshape(b, -1), h_.grad.reshape(b, -1), m_.grad.reshape(b, -1)], dim=1)
    total += stress('backward_b32', grads, None, 100, None)

  # Harmonic with processors.Add fused in (ddsp_harmonic_add_f32) against the two calls
  if not want or 'fused_add_b128' in want:
    b = 128
    amps, hd, f0, mags = controls(b, 1000, 100, 200.0, 1.0, 51)
    other = T(np.random.default_rng(52).standard_normal((b, 64000)))
    harm = ddsp.synths.Harmonic()
    noise = ddsp.synths.FilteredNoise(window_size=0)
    two_calls = (harm(amps, hd, f0) + other)
    torch.cuda.synchronize()

    def subs_a(it):
      return [('fused Add vs Harmonic then +', slice(0, b), lambda: two_calls)]
    total += stress('fused_add_b128', lambda: harm.call_add(amps, hd, f0, other), lambda: noise(mags), 64, subs_a)

  # round 5: effects.Reverb on persistent transform blocks (a block loops over transforms: the LDS array is reused behind a
  # barrier) - one IR for the batch (row pairs) and an IR per row; its backward pass (the two correlations, the shared IR's
  # gradient collected in a fixed order); rows alone against the rows of the batch
  for name, b, bir in (('reverb_b128_one_ir', 128, 1), ('reverb_b32_ir_per_row', 32, 32), ('reverb_b7_one_ir_odd', 7, 1)):
    if want and name not in want:
      continue
    r5 = np.random.default_rng(80 + b)
    audio = T(r5.standard_normal((b, 64000)))
    ir = T(r5.standard_normal((bir, 48000)) * np.exp(-np.arange(48000) / 9600.0))
    amps, hd, f0, mags = controls(32, 1000, 100, 70.0, 1.0, 81)
    rev = ddsp.effects.Reverb(add_dry=True)
    harm = ddsp.synths.Harmonic()

    def subs_r(it, b=b, bir=bir, audio=audio, ir=ir, rev=rev):
      r = int(rng.integers(0, b))
      return [('row %d alone' % r, slice(r, r + 1), lambda: rev(audio[r:r + 1], ir[r:r + 1] if bir > 1 else ir))]
    # (a row alone is compared where the batch runs the same arithmetic: with ONE impulse response two rows share a complex
    #  transform, a row alone does not - equal to rounding, tests/test_gpu_parity.py, not to the bit)
    total += stress(name, lambda audio=audio, ir=ir, rev=rev: rev(audio, ir), lambda: harm(amps, hd, f0), 4096,
                    subs_r if bir > 1 else None)
  if not want or 'reverb_backward_b16_one_ir' in want:
    b = 16
    r5 = np.random.default_rng(85)
    audio = T(r5.standard_normal((b, 64000)))
    ir = T(r5.standard_normal((1, 48000)) * np.exp(-np.arange(48000) / 9600.0))
    g = T(r5.standard_normal((b, 64000)))
    rev = ddsp.effects.Reverb(add_dry=True)

    def rgrads():
      a_ = audio.detach().requires_grad_(True)
      i_ = ir.detach().requires_grad_(True)
      rev(a_, i_).backward(g)
      return torch.cat([a_.grad, i_.grad.reshape(1, -1).expand(b, -1)[:, :48000]], dim=1)
    total += stress('reverb_backward_b16_one_ir', rgrads, None, 4096, None)
  # the backward pass through the chain of materialised envelopes (gathers, no atomics), and the stand-alone oscillator bank
  # with more than 64 sinusoids (its per-wavefront sums in a fixed order since round 5)
  if not want or 'harmonic_materialised_backward' in want:
    b, f, k, n = 4, 50, 100, 3300
    amps, hd, f0, mags = controls(b, f, k, 90.0, 2.0, 91)
    g = T(np.random.default_rng(92).standard_normal((b, n)))
    harm = ddsp.synths.Harmonic(n_samples=n, amp_resample_method='cubic')

    def mgrads():
      a_ = amps.detach().requires_grad_(True)
      h_ = hd.detach().requires_grad_(True)
      y = harm(a_, h_, f0)
      y.backward(g)
      return torch.cat([y.detach(), a_.grad.reshape(b, -1), h_.grad.reshape(b, -1)], dim=1)
    total += stress('harmonic_materialised_backward', mgrads, None, 66, None)

  summary = {'cases_run': sum(1 for r in log if 'iters' in r), 'label': args.label, 'lib': _lib.LIB_PATH, 'iters': args.iters, 'launches_with_a_difference': total,
             'device': torch.cuda.get_device_name(0)}
  print('SUMMARY ' + json.dumps(summary), flush=True)
  if args.out:
    with open(args.out, 'a') as f:
      for rec in log + [summary]:
        f.write(json.dumps(rec) + '\n')
  return summary


if __name__ == '__main__':
  sys.exit(1 if main()['launches_with_a_difference'] else 0)
