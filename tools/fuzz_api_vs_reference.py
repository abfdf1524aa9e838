"""The DROP-IN BOUNDARY, differentially: the Python mirror (ddsp_amd) against the reference's own classes and functions on the same
random arguments - valid and invalid (TEST INFRASTRUCTURE; build container only: needs /root/reference; the kernels run on the CPU
emulation of tests/hip_emu, the reference on the numpy stand-in for its TF ops, tests/golden/tf_numpy_shim.py).

For every draw: call the reference, call the mirror with the same arguments; then
  * the reference raised ValueError  ->  the mirror must raise ValueError too ("same ValueErrors raised in Python");
  * the reference returned           ->  the mirror must return the same shapes, the same dict keys, and values that agree within
                                         the parity contract's tolerance against the reference's fp32 chain.
What this checks is argument handling - ranks, broadcasting, defaults, optional arguments, error conditions - over many more
combinations than the goldens hold; numerics are the business of tools/fuzz_parity.py (GPU) and tests/.

    python tools/fuzz_api_vs_reference.py [--seconds 120] [--seed 1] [--only harmonic,...]
"""
import argparse, json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
import tf_numpy_shim
tf_numpy_shim.install('/root/reference')
import ddsp as R                                                  # noqa: E402  the reference (core, synths, effects, ...)
from ddsp import core as Rcore, synths as Rsynths, effects as Reffects, processors as Rproc, losses as Rlosses   # noqa: E402

os.environ.setdefault('DDSP_EMU_CUS', '4')
import ddsp_amd as M                                              # noqa: E402
from ddsp_amd import _lib, core as Mcore                          # noqa: E402
from tests.hip_emu import emu_simt                                # noqa: E402
_emu = emu_simt.load()
_lib.load = lambda: _emu
Mcore._device = lambda: torch.device('cpu')
Mcore._stream = lambda: None
Mcore._ws_bytes_cache.clear()


def npy(x):
  if isinstance(x, torch.Tensor):
    return x.detach().cpu().numpy()
  return np.asarray(x)


def run(fn):
  try:
    return 'ok', fn()
  except ValueError as e:
    return 'ValueError', str(e)[:160]
  except NotImplementedError as e:
    return 'NotImplementedError', str(e)[:160]


def run_any(fn):
  try:
    return 'ok', fn()
  except ValueError as e:
    return 'ValueError', str(e)
  except Exception as e:                                        # noqa: BLE001  (whatever numpy raises inside the reference)
    return type(e).__name__, str(e)


def same(ref, got, tol, what, path=''):
  """Structures (dicts, tuples) and arrays: same keys, same shapes, values within tol * max(1, |ref|)."""
  if isinstance(ref, dict):
    assert isinstance(got, dict) and sorted(ref) == sorted(got), (what, path, 'keys', sorted(ref), sorted(got) if isinstance(got, dict) else type(got))
    for k in ref:
      same(ref[k], got[k], tol, what, path + '/' + k)
    return
  if isinstance(ref, (tuple, list)):
    assert len(ref) == len(got), (what, path, 'length')
    for i, (r, g) in enumerate(zip(ref, got)):
      same(r, g, tol, what, path + '[%d]' % i)
    return
  r, g = npy(ref), npy(got)
  assert tuple(r.shape) == tuple(g.shape), (what, path, 'shape', r.shape, g.shape)
  if r.size:
    err = float(np.abs(r.astype(np.float64) - g.astype(np.float64)).max())
    lim = tol * max(1.0, float(np.abs(r).max()))
    assert err <= lim, (what, path, 'value', err, lim)


def compare(what, ref_fn, got_fn, tol):
  rs, rv = run_any(ref_fn)
  if rs not in ('ok', 'ValueError'):
    # the reference fails inside python / numpy with something else (ZeroDivisionError for upsample_with_windows on one frame
    # without an endpoint: `n_timesteps % 0`): not a documented behaviour - the mirror is free (it raises a ValueError there)
    return 'skipped (the reference does not define this: %s)' % rs
  gs, gv = run(got_fn)
  if rs == 'ValueError':
    assert gs == 'ValueError', (what, 'the reference raises ValueError(%s), the mirror: %s %s' % (rv, gs, str(gv)[:120]))
    return 'both raise'
  assert rs == 'ok', (what, 'reference', rs, rv)
  assert gs == 'ok', (what, 'the reference returns, the mirror raises', gs, gv)
  same(rv, gv, tol, what)
  return 'both return'


def maybe(rng, p=0.5):
  return bool(rng.random() < p)


# ---- cases -------------------------------------------------------------------------------------------------------------------------
def case_harmonic(rng):
  b, f, k = int(rng.integers(1, 3)), int(rng.integers(1, 12)), int(rng.choice([1, 5, 16, 40]))
  hop = int(rng.choice([8, 16, 50, 64]))
  sr = int(rng.choice([16000, 48000]))
  method = str(rng.choice(['window', 'window', 'linear', 'nearest', 'cubic', 'bogus']))
  n = f * hop if maybe(rng, 0.7) else int(rng.integers(1, f * hop + 5))
  kw = dict(n_samples=n, sample_rate=sr, amp_resample_method=method, normalize_below_nyquist=maybe(rng, 0.8),
            use_angular_cumsum=maybe(rng, 0.3))
  scale = maybe(rng, 0.8)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = rng.uniform(60.0, 1500.0, (b, f, 1)).astype(np.float32)
  if not scale:
    amps, hd = np.abs(amps) + 0.1, np.abs(hd) + 0.01
  bad = str(rng.choice(['none'] * 6 + ['frames', 'batch', 'rank']))
  if bad == 'frames' and f > 1:
    f0 = f0[:, :-1]
  elif bad == 'batch' and b > 1:
    hd = hd[:1]
  elif bad == 'rank':
    amps = amps[..., 0]
  as_dict = maybe(rng, 0.4)
  what = dict(case='harmonic', b=b, f=f, k=k, scale=scale, bad=bad, as_dict=as_dict, **kw)
  ref = lambda: Rsynths.Harmonic(scale_fn=Rcore.exp_sigmoid if scale else None, **kw)(amps, hd, f0, return_outputs_dict=as_dict)
  got = lambda: M.synths.Harmonic(scale_fn=M.core.exp_sigmoid if scale else None, **kw)(amps, hd, f0, return_outputs_dict=as_dict)
  # (malformed shapes: the reference broadcasts or fails somewhere inside numpy / TF with whatever error that raises - only the
  #  documented ValueErrors are part of the contract, so a malformed draw is checked only when the reference raises ValueError)
  if bad != 'none':
    rs, _ = run_any(ref)
    if rs != 'ValueError':
      return what, 'skipped (the reference does not define this)'
  # (the reference's fp32 chain - sequential cumsum - against the closed-form phase: HARM_FAITHFUL_ATOL of the tests; 'nearest' /
  #  'cubic' envelopes step or overshoot between frames and carry a little more of TF's fp32 positions)
  return what, compare(what, ref, got, 2e-3 if method in ('window', 'linear') else 4e-3)


def case_filtered_noise(rng):
  b, f, m = int(rng.integers(1, 3)), int(rng.integers(1, 12)), int(rng.choice([3, 9, 33, 65]))
  fs = int(rng.choice([8, 16, 64, 100]))
  n = f * fs - int(rng.integers(0, fs)) if maybe(rng, 0.8) else int(rng.integers(1, 3 * f * fs))
  ws = int(rng.choice([0, 257, 5, 2 * (m - 1), 2 * (m - 1) + 9, max(2, m // 2)]))
  scale = maybe(rng, 0.8)
  bias = float(rng.choice([-5.0, 0.0]))
  mags = rng.standard_normal((b, f, m)).astype(np.float32)
  if not scale:
    mags = np.abs(mags)
  noise = rng.uniform(-1, 1, (b, max(n, 1))).astype(np.float32)
  what = dict(case='filtered_noise', b=b, f=f, m=m, n=n, ws=ws, scale=scale, bias=bias)
  rkw = dict(n_samples=n, window_size=ws, scale_fn=Rcore.exp_sigmoid if scale else None, initial_bias=bias)
  mkw = dict(n_samples=n, window_size=ws, scale_fn=M.core.exp_sigmoid if scale else None, initial_bias=bias)
  # the reference draws its noise inside get_signal (tf.random.uniform): controls through the class, the signal through the
  # function get_signal calls (synths.py:192-196) on the same supplied noise
  r1 = compare(what, lambda: Rsynths.FilteredNoise(**rkw).get_controls(mags), lambda: M.synths.FilteredNoise(**mkw).get_controls(mags), 3e-6)
  ctl = npy(Rsynths.FilteredNoise(**rkw).get_controls(mags)['magnitudes'])
  r2 = compare(what, lambda: Rcore.frequency_filter(noise, ctl, window_size=ws), lambda: M.core.frequency_filter(noise, ctl, window_size=ws), 2e-5)
  r3 = compare(what, lambda: Rcore.frequency_impulse_response(ctl, window_size=ws), lambda: M.core.frequency_impulse_response(ctl, window_size=ws), 3e-6)
  return what, '%s / %s / %s' % (r1, r2, r3)


def case_resample(rng):
  rank = int(rng.choice([1, 2, 3, 3, 3, 4]))
  f = int(rng.integers(1, 20))
  shape = {1: (f,), 2: (2, f), 3: (2, f, int(rng.integers(1, 5))), 4: (2, f, 3, 2)}[rank]
  method = str(rng.choice(['nearest', 'linear', 'cubic', 'window', 'quadratic']))
  add_endpoint = maybe(rng)
  n = int(rng.choice([f * int(rng.integers(1, 40)), int(rng.integers(1, f * 40 + 1)), max(1, f - 1), int(rng.integers(1, f + 1))]))   # (down too:
  x = rng.standard_normal(shape).astype(np.float32)                                                       #  core_test.py:268-290)
  what = dict(case='resample', shape=shape, n=n, method=method, add_endpoint=add_endpoint)
  return what, compare(what, lambda: Rcore.resample(x, n, method=method, add_endpoint=add_endpoint),
                       lambda: M.core.resample(x, n, method=method, add_endpoint=add_endpoint), 2e-5)


def case_upsample_with_windows(rng):
  rank = int(rng.choice([2, 3, 3, 3, 4]))
  f = int(rng.integers(1, 12))
  shape = {2: (2, f), 3: (2, f, 3), 4: (2, f, 3, 1)}[rank]
  add_endpoint = maybe(rng)
  n = int(rng.choice([f * int(rng.integers(1, 30)), max(f - 1, 1) * int(rng.integers(1, 30)), int(rng.integers(1, 200))]))
  x = rng.standard_normal(shape).astype(np.float32)
  what = dict(case='upsample_with_windows', shape=shape, n=n, add_endpoint=add_endpoint)
  return what, compare(what, lambda: Rcore.upsample_with_windows(x, n, add_endpoint), lambda: M.core.upsample_with_windows(x, n, add_endpoint), 2e-5)


def case_fft_convolve(rng):
  b, f = int(rng.integers(1, 3)), int(rng.integers(1, 10))
  fs = int(rng.choice([4, 16, 50]))
  n = f * fs - int(rng.integers(0, fs)) if maybe(rng, 0.8) else int(rng.integers(2, 3 * f * fs))
  l = int(rng.choice([2, 3, 17, 64]))
  padding = str(rng.choice(['same', 'same', 'valid', 'full']))
  dc = int(rng.choice([-1, -1, 0, 2]))
  ir_b = int(rng.choice([1, b, b + 1]))
  ir_rank3 = maybe(rng, 0.8)
  x = rng.standard_normal((b, max(n, 2))).astype(np.float32)
  h = rng.standard_normal((ir_b, f, l) if ir_rank3 else (ir_b, l)).astype(np.float32)
  what = dict(case='fft_convolve', b=b, f=f if ir_rank3 else 1, n=max(n, 2), l=l, padding=padding, dc=dc, ir_b=ir_b)
  return what, compare(what, lambda: Rcore.fft_convolve(x, h, padding=padding, delay_compensation=dc),
                       lambda: M.core.fft_convolve(x, h, padding=padding, delay_compensation=dc), 2e-5)


def case_reverb(rng):
  b, n, l = int(rng.integers(1, 4)), int(rng.integers(2, 600)), int(rng.integers(2, 400))
  add_dry = maybe(rng)
  ir_rank = int(rng.choice([2, 2, 3]))
  ir_b = int(rng.choice([b, b, 1])) if b > 1 else 1
  x = rng.standard_normal((b, n)).astype(np.float32)
  h = (rng.standard_normal((ir_b, l)) * np.exp(-np.arange(l) / (0.3 * l + 1))).astype(np.float32)
  h_in = h[:, :, None] if ir_rank == 3 else h
  as_dict = maybe(rng, 0.3)
  what = dict(case='reverb', b=b, n=n, l=l, add_dry=add_dry, ir_rank=ir_rank, ir_b=ir_b, as_dict=as_dict)
  rs, _ = run_any(lambda: Reffects.Reverb(add_dry=add_dry)(x, h_in))
  if rs not in ('ok', 'ValueError'):
    return what, 'skipped (the reference does not define this)'
  return what, compare(what, lambda: Reffects.Reverb(add_dry=add_dry)(x, h_in, return_outputs_dict=as_dict),
                       lambda: M.effects.Reverb(add_dry=add_dry)(x, h_in, return_outputs_dict=as_dict), 3e-5)


def case_small_core(rng):
  which = str(rng.choice(['exp_sigmoid', 'safe_divide', 'safe_log', 'get_harmonic_frequencies', 'remove_above_nyquist',
                          'normalize_harmonics', 'get_fft_size', 'crop', 'angular_cumsum', 'add', 'apply_window']))
  what = dict(case='small_core', fn=which)
  if which == 'apply_window':
    l0 = int(rng.integers(1, 300)); ws = int(rng.choice([0, 1, 2, 3, l0, l0 + 3, max(1, l0 // 2), int(rng.integers(1, 320))])); causal = maybe(rng)
    x = rng.standard_normal((2, 3, l0) if maybe(rng) else (4, l0)).astype(np.float32)
    what.update(l0=l0, ws=ws, causal=causal)
    return what, compare(what, lambda: Rcore.apply_window_to_impulse_response(x, ws, causal), lambda: M.core.apply_window_to_impulse_response(x, ws, causal), 2e-6)
  if which == 'exp_sigmoid':
    x = (10.0 * rng.standard_normal((2, 7, 3))).astype(np.float32)
    kw = dict(exponent=float(rng.choice([10.0, 2.0])), max_value=float(rng.choice([2.0, 1.0])), threshold=float(rng.choice([1e-7, 1e-3])))
    what.update(kw)
    return what, compare(what, lambda: Rcore.exp_sigmoid(x, **kw), lambda: M.core.exp_sigmoid(x, **kw), 2e-6)
  if which == 'safe_divide':
    a_ = rng.standard_normal((3, 5)).astype(np.float32); d = rng.standard_normal((3, 5)).astype(np.float32); d[0, :2] = 0.0
    return what, compare(what, lambda: Rcore.safe_divide(a_, d), lambda: M.core.safe_divide(a_, d), 1e-6)
  if which == 'safe_log':
    x = rng.standard_normal((4, 6)).astype(np.float32); x[0, 0] = 0.0
    return what, compare(what, lambda: Rcore.safe_log(x), lambda: M.core.safe_log(x), 2e-6)
  if which == 'get_harmonic_frequencies':
    f0 = rng.uniform(50, 500, (2, 9, 1)).astype(np.float32); k = int(rng.integers(1, 70))
    return what, compare(what, lambda: Rcore.get_harmonic_frequencies(f0, k), lambda: M.core.get_harmonic_frequencies(f0, k), 0.0)
  if which == 'remove_above_nyquist':
    fr = rng.uniform(0, 12000, (2, 30, 5)).astype(np.float32); am = rng.standard_normal((2, 30, 5)).astype(np.float32)
    sr = int(rng.choice([16000, 8000]))
    return what, compare(what, lambda: Rcore.remove_above_nyquist(fr, am, sr), lambda: M.core.remove_above_nyquist(fr, am, sr), 0.0)
  if which == 'normalize_harmonics':
    hd = np.abs(rng.standard_normal((2, 8, 12))).astype(np.float32); f0 = rng.uniform(100, 3000, (2, 8, 1)).astype(np.float32)
    with_f0 = maybe(rng, 0.7)
    what['with_f0'] = with_f0
    if with_f0:
      return what, compare(what, lambda: Rcore.normalize_harmonics(hd, f0, 16000), lambda: M.core.normalize_harmonics(hd, f0, 16000), 2e-6)
    return what, compare(what, lambda: Rcore.normalize_harmonics(hd), lambda: M.core.normalize_harmonics(hd), 2e-6)
  if which == 'get_fft_size':
    fs, l, p2 = int(rng.integers(1, 5000)), int(rng.integers(1, 70000)), maybe(rng)
    what.update(frame=fs, ir=l, power_of_2=p2)
    return what, compare(what, lambda: np.asarray(Rcore.get_fft_size(fs, l, p2)), lambda: np.asarray(M.core.get_fft_size(fs, l, p2)), 0.0)
  if which == 'crop':
    n, l = int(rng.integers(4, 200)), int(rng.integers(1, 60))
    total = n + l - 1
    x = rng.standard_normal((2, total)).astype(np.float32)
    padding = str(rng.choice(['same', 'valid', 'nope'])); dc = int(rng.choice([-1, 0, 3, l]))
    what.update(n=n, l=l, padding=padding, dc=dc)
    return what, compare(what, lambda: Rcore.crop_and_compensate_delay(x, n, l, padding, dc), lambda: M.core.crop_and_compensate_delay(x, n, l, padding, dc), 0.0)
  if which == 'angular_cumsum':
    n = int(rng.integers(1, 2500)); w = rng.uniform(0, 1.0, (2, n, 3)).astype(np.float32); cs = int(rng.choice([1000, 100, 7]))
    what.update(n=n, chunk=cs)
    # (phases: equal modulo a revolution - compared as points on the circle; the reference's fp32 scan against an fp64 one)
    circle = lambda ph: np.stack([np.cos(npy(ph).astype(np.float64)), np.sin(npy(ph).astype(np.float64))])
    return what, compare(what, lambda: circle(Rcore.angular_cumsum(w, cs)), lambda: circle(M.core.angular_cumsum(w, cs)), 2e-3)
  x, y = rng.standard_normal((2, 50)).astype(np.float32), rng.standard_normal((2, 50)).astype(np.float32)
  return what, compare(what, lambda: Rproc.Add()(x, y), lambda: M.processors.Add()(x, y), 0.0)


def case_spectral_loss(rng):
  b, n = int(rng.integers(1, 3)), int(rng.choice([64, 500, 1500]))
  sizes = tuple(int(s) for s in rng.permutation([1024, 512, 256, 128, 64, 768, 384, 192, 96, 48])[:int(rng.integers(1, 4))])
  kw = dict(fft_sizes=sizes, loss_type=str(rng.choice(['L1', 'L1', 'L2', 'COSINE', 'L3'])),
            mag_weight=float(rng.choice([1.0, 0.0])), logmag_weight=float(rng.choice([0.0, 1.0])),
            delta_time_weight=float(rng.choice([0.0, 0.0, 1.0])), delta_freq_weight=float(rng.choice([0.0, 0.0, 1.0])),
            cumsum_freq_weight=float(rng.choice([0.0, 0.0, 1.0])), loudness_weight=float(rng.choice([0.0, 0.0, 0.5])))
  t = (0.3 * rng.standard_normal((b, n))).astype(np.float32)
  x = (0.8 * t + 0.05 * rng.standard_normal((b, n))).astype(np.float32)
  what = dict(case='spectral_loss', b=b, n=n, **kw)
  def ref():
    v = npy(Rlosses.SpectralLoss(**kw)(t, x))
    return np.nan_to_num(v, nan=-1.0)                          # (one frame and a delta-time term: NaN in both)
  def got():
    return np.nan_to_num(npy(M.losses.SpectralLoss(**kw)(t, x)), nan=-1.0)
  return what, compare(what, ref, got, 5e-5)


def case_processors(rng):
  which = str(rng.choice(['mix', 'crop', 'fir_filter', 'group']))
  what = dict(case='processors', which=which)
  if which == 'mix':
    b, n, f = int(rng.integers(1, 3)), int(rng.integers(8, 300)), int(rng.integers(1, 8))
    # ([batch, n, 1] signals: the reference's docstring says "2-D or 3-D", its broadcast against the [batch, n, 1] mix level
    #  only works for 3-D - the mirror takes both)
    n2 = n if maybe(rng, 0.8) else n + 1
    x, y = rng.standard_normal((b, n, 1)).astype(np.float32), rng.standard_normal((b, n2, 1)).astype(np.float32)
    lvl = rng.standard_normal((b, f, 1)).astype(np.float32)
    what.update(b=b, n=n, n2=n2, f=f)
    return what, compare(what, lambda: Rproc.Mix()(x, y, lvl), lambda: M.processors.Mix()(x, y, lvl), 2e-6)
  if which == 'crop':
    b, n = int(rng.integers(1, 3)), int(rng.integers(8, 300))
    fs = int(rng.integers(1, 40)); loc = str(rng.choice(['front', 'center', 'back', 'middle']))
    x = rng.standard_normal((b, n)).astype(np.float32)
    what.update(b=b, n=n, frame_size=fs, crop_location=loc)
    return what, compare(what, lambda: Rproc.Crop(fs, loc)(x), lambda: M.processors.Crop(fs, loc)(x), 0.0)
  if which == 'fir_filter':
    b, f, m = int(rng.integers(1, 3)), int(rng.integers(1, 8)), int(rng.choice([5, 17, 65]))
    fs = int(rng.choice([16, 64])); n = f * fs
    ws = int(rng.choice([0, 257, 9]))
    x = rng.standard_normal((b, n)).astype(np.float32); mags = rng.standard_normal((b, f, m)).astype(np.float32)
    what.update(b=b, f=f, m=m, n=n, ws=ws)
    return what, compare(what, lambda: Reffects.FIRFilter(window_size=ws)(x, mags), lambda: M.effects.FIRFilter(window_size=ws)(x, mags), 2e-5)
  # a ProcessorGroup: two Harmonic synths (different pitch ranges) summed, then cropped - every node deterministic
  b, f, k, hop = int(rng.integers(1, 3)), int(rng.integers(2, 8)), int(rng.choice([4, 16])), 16
  n = f * hop
  feats = dict(a1=rng.standard_normal((b, f, 1)).astype(np.float32), h1=rng.standard_normal((b, f, k)).astype(np.float32),
               f1=rng.uniform(100, 400, (b, f, 1)).astype(np.float32), a2=rng.standard_normal((b, f, 1)).astype(np.float32),
               h2=rng.standard_normal((b, f, k)).astype(np.float32), f2=rng.uniform(400, 900, (b, f, 1)).astype(np.float32))
  def dag(P, S):
    return [(S.Harmonic(n_samples=n, name='low'), ['a1', 'h1', 'f1']), (S.Harmonic(n_samples=n, name='high'), ['a2', 'h2', 'f2']),
            (P.Add(name='add'), ['low/signal', 'high/signal']), (P.Crop(8, 'front', name='crop'), ['add/signal'])]
  as_dict = maybe(rng)
  what.update(b=b, f=f, k=k, as_dict=as_dict)
  def flat(out):
    if not as_dict:
      return out
    # processors.py:124-134: {'signal': the last node's signal, 'controls': the DAG's outputs (per node: controls and signal)}
    return {key: out['controls'][key]['signal'] for key in ('low', 'high', 'add', 'crop')} | {'signal': out['signal']}
  return what, compare(what, lambda: flat(Rproc.ProcessorGroup(dag=dag(Rproc, Rsynths))(feats, return_outputs_dict=as_dict)),
                       lambda: flat(M.processors.ProcessorGroup(dag=dag(M.processors, M.synths))(feats, return_outputs_dict=as_dict)), 2e-3)


def case_synthesis(rng):
  which = str(rng.choice(['harmonic_synthesis', 'oscillator_bank', 'harmonic_oscillator_bank', 'streaming']))
  what = dict(case='synthesis', which=which)
  b, f, k = int(rng.integers(1, 3)), int(rng.integers(2, 8)), int(rng.choice([1, 6, 20]))
  sr = int(rng.choice([16000, 48000]))
  if which == 'harmonic_synthesis':
    hop = int(rng.choice([16, 50])); method = str(rng.choice(['window', 'linear', 'nearest', 'cubic']))
    n = f * hop if method == 'window' or maybe(rng) else int(rng.integers(f, f * hop))
    fr = rng.uniform(80, 900, (b, f, 1)).astype(np.float32); am = rng.uniform(0.1, 1, (b, f, 1)).astype(np.float32)
    shifts = (0.01 * rng.standard_normal((b, f, k))).astype(np.float32) if maybe(rng, 0.4) else None
    hd = rng.uniform(0, 1, (b, f, k)).astype(np.float32) if maybe(rng, 0.7) else None
    kw = dict(harmonic_shifts=shifts, harmonic_distribution=hd, n_samples=n, sample_rate=sr, amp_resample_method=method,
              use_angular_cumsum=maybe(rng, 0.3))
    what.update(b=b, f=f, k=k, n=n, method=method, shifts=shifts is not None, distribution=hd is not None)
    return what, compare(what, lambda: Rcore.harmonic_synthesis(fr, am, **kw), lambda: M.core.harmonic_synthesis(fr, am, **kw), 4e-3)
  n = int(rng.integers(1, 600))
  if which == 'oscillator_bank':
    fr = rng.uniform(0, sr * 0.6, (b, n, k)).astype(np.float32); am = rng.standard_normal((b, n, k)).astype(np.float32)
    kw = dict(sample_rate=sr, sum_sinusoids=maybe(rng), use_angular_cumsum=maybe(rng, 0.3))
    what.update(b=b, n=n, k=k, **kw)
    return what, compare(what, lambda: Rcore.oscillator_bank(fr, am, **kw), lambda: M.core.oscillator_bank(fr, am, **kw), 2e-3)
  phase = rng.uniform(0, 6.0, (b, 1, 1)).astype(np.float32) if maybe(rng) else None
  def wrapped(out):                                           # (audio, final_phase): the phase modulo a revolution
    return out[0], np.mod(npy(out[1]) + 1e-4, 2 * np.pi)
  if which == 'harmonic_oscillator_bank':
    fr = rng.uniform(50, 500, (b, n, 1)).astype(np.float32); am = rng.uniform(0, 1, (b, n, k)).astype(np.float32)
    kw = dict(initial_phase=phase, sample_rate=sr, use_angular_cumsum=maybe(rng, 0.7))
    what.update(b=b, n=n, k=k, angular=kw['use_angular_cumsum'], phase=phase is not None)
    return what, compare(what, lambda: wrapped(Rcore.harmonic_oscillator_bank(fr, am, **kw)), lambda: wrapped(M.core.harmonic_oscillator_bank(fr, am, **kw)), 2e-3)
  method = str(rng.choice(['linear', 'window', 'nearest']))
  n = f * int(rng.integers(2, 60)) if method == 'window' else n + f
  fr = rng.uniform(80, 500, (b, f, 1)).astype(np.float32); am = rng.uniform(0.1, 1, (b, f, 1)).astype(np.float32)
  hd = rng.uniform(0, 1, (b, f, k)).astype(np.float32) if maybe(rng, 0.7) else None
  kw = dict(harmonic_distribution=hd, initial_phase=phase, n_samples=n, sample_rate=sr, amp_resample_method=method)
  what.update(b=b, f=f, k=k, n=n, method=method, distribution=hd is not None, phase=phase is not None)
  return what, compare(what, lambda: wrapped(Rcore.streaming_harmonic_synthesis(fr, am, **kw)), lambda: wrapped(M.core.streaming_harmonic_synthesis(fr, am, **kw)), 2e-3)


def case_spectral_ops(rng):
  from ddsp import spectral_ops as RS
  b, n = int(rng.integers(1, 3)), int(rng.integers(100, 5000))
  x = (0.3 * rng.standard_normal((b, n))).astype(np.float32)
  if maybe(rng):
    size = int(rng.choice([64, 256, 512, 2048, 192, 768, 100, 1000]))
    overlap = float(rng.choice([0.75, 0.5, 0.875]))
    pad_end = maybe(rng, 0.7)
    what = dict(case='spectral_ops', fn='compute_mag', b=b, n=n, size=size, overlap=overlap, pad_end=pad_end)
    if not pad_end and n < size:
      return what, 'skipped (no frame fits)'
    return what, compare(what, lambda: RS.compute_mag(x, size, overlap, pad_end), lambda: M.spectral_ops.compute_mag(x, size, overlap, pad_end), 3e-6)
  if maybe(rng, 0.4):
    size = int(rng.choice([64, 100, 256, 512, 2048, 192, 768, 1000]))
    overlap, pad_end = float(rng.choice([0.75, 0.5, 0.875])), maybe(rng, 0.7)
    what = dict(case='spectral_ops', fn='stft', b=b, n=n, size=size, overlap=overlap, pad_end=pad_end)
    if not pad_end and n < size:
      return what, 'skipped (no frame fits)'
    as_pairs = lambda z: np.stack([np.real(npy(z)), np.imag(npy(z))], -1)
    return what, compare(what, lambda: as_pairs(RS.stft(x, size, overlap, pad_end)), lambda: as_pairs(M.spectral_ops.stft(x, size, overlap, pad_end)), 3e-6)
  kw = dict(sample_rate=int(rng.choice([16000, 24000, 44100])), frame_rate=int(rng.choice([250, 100, 50])),
            n_fft=int(rng.choice([512, 1024, 2048])), range_db=float(rng.choice([80.0, 120.0])), ref_db=float(rng.choice([0.0, 20.7])),
            padding=str(rng.choice(['center', 'same', 'valid', 'bogus'])))
  what = dict(case='spectral_ops', fn='compute_loudness', b=b, n=n, **kw)
  if kw['padding'] == 'valid' and n < kw['n_fft']:
    return what, 'skipped (no frame fits)'
  return what, compare(what, lambda: RS.compute_loudness(x, **kw), lambda: M.spectral_ops.compute_loudness(x, **kw), 1e-4)     # (dB: 0.012 at -120)


def case_mean_difference(rng):
  from ddsp import losses as RL
  rank = int(rng.integers(1, 5))
  shape = tuple(int(rng.integers(1, 7)) for _ in range(rank))
  t = rng.standard_normal(shape).astype(np.float32)
  v = rng.standard_normal(shape).astype(np.float32)
  loss_type = str(rng.choice(['L1', 'L2', 'COSINE', 'l1', 'cosine', 'L3']))
  kind = str(rng.choice(['none', 'scalar', 'full', 'rows', 'lead']))
  if kind == 'none':
    w = None
  elif kind == 'scalar':
    w = float(rng.uniform(0.0, 2.0))
  else:
    wshape = {'full': shape, 'rows': shape[:-1] + (1,), 'lead': (shape[0],) + (1,) * (rank - 1)}[kind]
    if loss_type.upper() == 'COSINE':
      wshape = wshape[:-1] + (1,)                  # (cosine_distance's weights go against [..., 1])
    w = rng.standard_normal(wshape).astype(np.float32)
    if maybe(rng, 0.3):
      w.flat[0] = 0.0
  what = dict(case='mean_difference', shape=shape, loss_type=loss_type, weights=kind)
  return what, compare(what, lambda: RL.mean_difference(t, v, loss_type, w), lambda: M.losses.mean_difference(t, v, loss_type, w), 3e-6)


CASES = dict(harmonic=case_harmonic, filtered_noise=case_filtered_noise, resample=case_resample,
             upsample_with_windows=case_upsample_with_windows, fft_convolve=case_fft_convolve, reverb=case_reverb,
             small_core=case_small_core, spectral_loss=case_spectral_loss, processors=case_processors, synthesis=case_synthesis,
             spectral_ops=case_spectral_ops, mean_difference=case_mean_difference)

if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--seconds', type=float, default=120.0)
  ap.add_argument('--seed', type=int, default=1)
  ap.add_argument('--only', default='')
  args = ap.parse_args()
  names = [s for s in args.only.split(',') if s] or list(CASES)
  counts = {k: {} for k in names}; failures = []
  t_end, i = time.time() + args.seconds, 0
  while time.time() < t_end:
    name = names[i % len(names)]; seed = args.seed * 1000003 + i; i += 1
    try:
      what, outcome = CASES[name](np.random.default_rng(seed))
      key = 'skipped' if outcome.startswith('skipped') else outcome
      counts[name][key] = counts[name].get(key, 0) + 1
    except Exception as e:                                      # noqa: BLE001
      failures.append({'case': name, 'seed': seed, 'error': repr(e)[:700]})
      print('FAIL', json.dumps(failures[-1]), flush=True)
  print('SUMMARY', json.dumps({'outcomes': counts, 'failures': len(failures), 'seed': args.seed}))
