"""GPU parity at the shapes the contract is quoted on: BASELINE.json's north-star shape
(batch 128 x 4 s @ 16 kHz, 100 harmonics, 65 bands; ddsp/training/gin/models/ae.gin:15,31-33,59-68), configs[4] at
full length (48 kHz, 200 harmonics, 10 s, frame size 192; gin/models/vst/vst_48k.gin:16-17,102), configs[3]'s per-GPU
node (ProcessorGroup[Harmonic, FilteredNoise, Add, Reverb(48 000 taps)] at batch 128; gin/models/solo_instrument.gin:26-40)
and every row of configs[1] (batch 32).  What the smaller parity tests cannot see: the multi-tick schedules of the
persistent kernels (16.5 chunks per block at batch 128 against 4 at batch 32), the buffer rotation over many ticks,
the fp64 phase prefix over 2500 frames.

Tolerances are the ones of tests/test_gpu_parity.py (HARM_TABLE_ATOL for the wavetable kernel, HARM_TRUTH_ATOL for
the direct sum, noise_tol for the FIR); every comparison is logged to $DDSP_PARITY_LOG.
"""
import numpy as np
import pytest
import torch

from conftest import parity_check
from oracle import ddsp_oracle as O
from test_gpu_parity import (DEV, HARM_TABLE_ATOL, HARM_TRUTH_ATOL, _harmonic_exact, assert_knife_edges_take_the_fp32_side,  # noqa: F401
                             canonical_inputs, ddsp, noise_tol, npy, reverb_tol)

pytestmark = pytest.mark.gpu


def _rows_equal_alone(run, args, full, rows):
  """Row r of the batched result is bit-equal to the same row run as a batch of one (batch rows are independent)."""
  for r in rows:
    one = npy(run(*[a[r:r + 1] for a in args]))
    np.testing.assert_array_equal(one, full[r:r + 1], err_msg='row %d' % r)


def test_north_star_shape_batch128_harmonic(ddsp):
  b = 128
  x = canonical_inputs(b, seed=21)
  harm = ddsp.synths.Harmonic()
  args = tuple(torch.as_tensor(x[k], device=DEV) for k in ('amplitudes', 'harmonic_distribution', 'f0_hz'))
  full = npy(harm(*args))
  assert full.shape == (b, 64000) and np.isfinite(full).all()
  # every row: bit-equal to the row run alone, and to the row inside a batch of 32 (configs[1]'s schedule)
  _rows_equal_alone(harm, args, full, range(b))
  for q in range(4):
    np.testing.assert_array_equal(npy(harm(*[a[32 * q:32 * q + 32] for a in args])), full[32 * q:32 * q + 32])
  # 8 random rows against exact arithmetic, 2 of them against the oracle's fp64 truth as well
  rng = np.random.default_rng(22)
  rows = sorted(rng.choice(b, 8, replace=False).tolist())
  scale = max(1.0, float(O.exp_sigmoid(x['amplitudes'].astype(np.float64), dtype=np.float64).max()))
  for i, r in enumerate(rows):
    sl = slice(r, r + 1)
    exact = _harmonic_exact(x['amplitudes'][sl], x['harmonic_distribution'][sl], x['f0_hz'][sl], 64000, 16000, 'window')
    parity_check(full[sl], exact, HARM_TABLE_ATOL * scale, 'batch 128 row %d vs exact arithmetic' % r)
    if i < 2:
      truth = O.harmonic(x['amplitudes'][sl], x['harmonic_distribution'][sl], x['f0_hz'][sl], dtype=np.float64)
      parity_check(full[sl], truth, HARM_TABLE_ATOL * scale, 'batch 128 row %d vs fp64 oracle' % r)


def test_north_star_shape_batch128_filtered_noise(ddsp):
  b = 128
  x = canonical_inputs(b, seed=23)
  mags = torch.as_tensor(x['magnitudes'], device=DEV)
  rng = np.random.default_rng(24)
  noise_np = rng.uniform(-1, 1, (b, 64000)).astype(np.float32)
  noise = torch.as_tensor(noise_np, device=DEV)
  fn = ddsp.synths.FilteredNoise(window_size=0)
  full = npy(fn(mags, noise=noise))
  assert full.shape == (b, 64000) and np.isfinite(full).all()
  _rows_equal_alone(lambda m, z: fn(m, noise=z), (mags, noise), full, range(b))
  for q in range(4):
    np.testing.assert_array_equal(npy(fn(mags[32 * q:32 * q + 32], noise=noise[32 * q:32 * q + 32])),
                                  full[32 * q:32 * q + 32])
  rows = sorted(rng.choice(b, 8, replace=False).tolist())
  for r in rows:
    ref = O.filtered_noise(x['magnitudes'][r:r + 1], noise_np[r:r + 1], 0, dtype=np.float64)
    parity_check(full[r:r + 1], ref, noise_tol(ref), 'batch 128 row %d vs fp64 oracle' % r)
  # generated noise (what the bench runs): rows against the generator's restatement pushed through the fp64 oracle
  gen = ddsp.synths.FilteredNoise(window_size=0, seed=7)
  y = npy(gen(mags))
  u = O.device_uniform_noise(b, 64000, 7, 0)
  for r in rows[:4]:
    ref = O.filtered_noise(x['magnitudes'][r:r + 1], u[r:r + 1], 0, dtype=np.float64)
    parity_check(y[r:r + 1], ref, noise_tol(ref), 'batch 128 row %d, generated noise' % r)


def test_config5_full_length_48k_200_harmonics(ddsp):
  """BASELINE configs[4] per clip: 10 s at 48 kHz, 2500 frames of 192 samples, 200 harmonics, 'linear' envelopes,
  angular cumsum (vst_48k.gin), Harmonic and FilteredNoise, against exact arithmetic / the fp64 oracle."""
  b, f, hop, k, sr = 2, 2500, 192, 200, 48000
  n = f * hop
  rng = np.random.default_rng(31)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = (100 + rng.standard_normal((b, f, 1))).astype(np.float32)
  mags = rng.standard_normal((b, f, 65)).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method='linear', use_angular_cumsum=True)
  ours = npy(synth(amps, hd, f0))
  assert ours.shape == (b, n)
  exact = _harmonic_exact(amps, hd, f0, n, sr, 'linear')
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  # the wavetable kernel (129 .. 200 harmonics since the end of round 3), held to its own tolerance
  parity_check(ours, exact, HARM_TABLE_ATOL * scale, 'config 5 full length vs exact arithmetic')
  np.testing.assert_array_equal(npy(synth(amps[1:], hd[1:], f0[1:])), ours[1:])
  noise_np = rng.uniform(-1, 1, (b, n)).astype(np.float32)
  fn = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)
  z = npy(fn(mags, noise=noise_np))
  ref = O.filtered_noise(mags, noise_np, 0, dtype=np.float64)
  parity_check(z, ref, noise_tol(ref), 'config 5 full length FilteredNoise vs fp64 oracle')


def test_config4_node_processor_group_reverb_batch128(ddsp):
  """BASELINE configs[3] per GPU: ProcessorGroup[Harmonic, FilteredNoise, Add, Reverb(48 000)] at batch 128
  (solo_instrument.gin:26-40); two rows against the oracle, every row of the dry mix against its parts."""
  b, n, l = 128, 64000, 48000
  x = canonical_inputs(b, seed=41)
  feats = {'amps': x['amplitudes'], 'harmonic_distribution': x['harmonic_distribution'], 'f0_hz': x['f0_hz'],
           'magnitudes': x['magnitudes']}
  harm = ddsp.synths.Harmonic(n_samples=n)
  noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=3)
  add = ddsp.processors.Add()
  rev = ddsp.effects.Reverb(trainable=True, reverb_length=l)
  rev.build(device=torch.device(DEV))
  rng = np.random.default_rng(42)
  rev._ir = ddsp.core.tf_float32(rng.standard_normal(l) * np.exp(-np.arange(l) / 2000.0) * 0.05)
  dag = [(harm, ['amps', 'harmonic_distribution', 'f0_hz']), (noise, ['magnitudes']),
         (add, ['filtered_noise/signal', 'harmonic/signal']), (rev, ['add/signal'])]
  group = ddsp.processors.ProcessorGroup(dag=dag)
  outs = group.get_controls(feats)
  h, z = npy(outs['harmonic']['signal']), npy(outs['filtered_noise']['signal'])
  dry = npy(outs['add']['signal'])
  np.testing.assert_array_equal(dry, h + z)
  got = npy(group.get_signal(outs))
  assert got.shape == (b, n) and np.isfinite(got).all()
  u = O.device_uniform_noise(b, n, 3, 0)
  scale = max(1.0, float(O.exp_sigmoid(x['amplitudes'].astype(np.float64), dtype=np.float64).max()))
  for r in (5, 101):
    sl = slice(r, r + 1)
    truth_h = O.harmonic(x['amplitudes'][sl], x['harmonic_distribution'][sl], x['f0_hz'][sl], n, dtype=np.float64)
    parity_check(h[sl], truth_h, HARM_TABLE_ATOL * scale, 'config 4 node, harmonic row %d' % r)
    truth_z = O.filtered_noise(x['magnitudes'][sl], u[sl], 0, dtype=np.float64)
    parity_check(z[sl], truth_z, noise_tol(truth_z), 'config 4 node, noise row %d' % r)
    ref = O.reverb(truth_h + truth_z, npy(rev._ir), add_dry=True, dtype=np.float64)
    parity_check(got[sl], ref, reverb_tol(ref) + HARM_TABLE_ATOL * scale * 4, 'config 4 node, output row %d' % r)
    # the Reverb alone, on the dry mix the GPU produced
    ref2 = O.reverb(dry[sl], npy(rev._ir), add_dry=True, dtype=np.float64)
    parity_check(got[sl], ref2, reverb_tol(ref2), 'config 4 node, reverb row %d' % r)


def test_config1_batch32_every_row(ddsp):
  """Every row of configs[1] (batch 32): bit-equal to the row run alone (both default kernels), four rows against the
  oracle (tests/test_gpu_parity.py::test_full_size_properties_batch32 compares one)."""
  b = 32
  x = canonical_inputs(b, seed=5)
  harm, fn = ddsp.synths.Harmonic(), ddsp.synths.FilteredNoise(window_size=0)
  args = tuple(torch.as_tensor(x[k], device=DEV) for k in ('amplitudes', 'harmonic_distribution', 'f0_hz'))
  full = npy(harm(*args))
  _rows_equal_alone(harm, args, full, range(b))
  scale = max(1.0, float(O.exp_sigmoid(x['amplitudes'].astype(np.float64), dtype=np.float64).max()))
  for r in (0, 13, 22, 31):
    sl = slice(r, r + 1)
    truth = O.harmonic(x['amplitudes'][sl], x['harmonic_distribution'][sl], x['f0_hz'][sl], dtype=np.float64)
    parity_check(full[sl], truth, HARM_TABLE_ATOL * scale, 'batch 32 row %d vs fp64 oracle' % r)
  rng = np.random.default_rng(6)
  noise_np = rng.uniform(-1, 1, (b, 64000)).astype(np.float32)
  mags, noise = torch.as_tensor(x['magnitudes'], device=DEV), torch.as_tensor(noise_np, device=DEV)
  y = npy(fn(mags, noise=noise))
  _rows_equal_alone(lambda m, z: fn(m, noise=z), (mags, noise), y, range(b))
  for r in (0, 13, 22, 31):
    ref = O.filtered_noise(x['magnitudes'][r:r + 1], noise_np[r:r + 1], 0, dtype=np.float64)
    parity_check(y[r:r + 1], ref, noise_tol(ref), 'batch 32 row %d vs fp64 oracle' % r)


@pytest.mark.parametrize('f0_centre', [200.0, 333.0])
def test_north_star_shape_batch128_harmonic_crossing_regimes(ddsp, f0_centre):
  """The north-star shape where the Nyquist-crossing path runs in most frames: f0 = 200 +- 1 Hz is
  SURVEY 8(d)'s second regime (processors_test.py:40; harmonic 40 sits on 8 kHz), 333 +- 1 Hz puts harmonic 24 there and the
  table reads on their worst bank-conflict resonance - on the 16.5-chunk schedule of batch 128.  Every row bit-equal to the row
  run alone and inside a batch of 32; six rows against exact arithmetic outside the knife-edge samples (audio-rate mask in TF's
  fp32 op order, ddsp/core.py:942-944)."""
  b = 128
  x = canonical_inputs(b, seed=int(f0_centre))
  rng = np.random.default_rng(int(f0_centre) + 1)
  f0 = (f0_centre + rng.standard_normal((b, 1000, 1))).astype(np.float32)
  harm = ddsp.synths.Harmonic()
  args = tuple(torch.as_tensor(a, device=DEV) for a in (x['amplitudes'], x['harmonic_distribution'], f0))
  full = npy(harm(*args))
  assert full.shape == (b, 64000) and np.isfinite(full).all()
  _rows_equal_alone(harm, args, full, range(b))
  for q in range(4):
    np.testing.assert_array_equal(npy(harm(*[a[32 * q:32 * q + 32] for a in args])), full[32 * q:32 * q + 32])
  scale = max(1.0, float(O.exp_sigmoid(x['amplitudes'].astype(np.float64), dtype=np.float64).max()))
  for r in sorted(rng.choice(b, 6, replace=False).tolist()):
    sl = slice(r, r + 1)
    exact, knife, exact32 = _harmonic_exact(x['amplitudes'][sl], x['harmonic_distribution'][sl], f0[sl], 64000, 16000, 'window',
                                            with_knife_edges='fp32 mask')
    assert knife.mean() <= 2e-2
    err = float(np.abs(full[sl] - exact)[~knife].max())
    parity_check(np.where(knife, exact, full[sl]), exact, HARM_TABLE_ATOL * scale,
                 'batch 128, f0 = %g +- 1 Hz, row %d vs exact arithmetic (%.2e)' % (f0_centre, r, err))
    # ... and the knife-edge samples against the same sum with the mask decided in the reference's fp32 op order (core.py:942-944):
    # no sample is left unchecked
    assert_knife_edges_take_the_fp32_side(full[sl], exact32, knife, HARM_TABLE_ATOL * scale, ('batch 128', f0_centre, r))


def test_config5_batch32_every_row(ddsp):
  """BASELINE configs[4] per GPU as bench.py runs it: batch 32 x 10 s at 48 kHz, 2500 frames of 192
  samples, 200 harmonics (gin/models/vst/vst_48k.gin:16-17,102) - the 129 .. 200-harmonic instances on ~312 frames per block,
  ten chunks with the fp64 phase prefix carried across row boundaries.  Every row bit-equal to the row run alone; four rows
  against exact arithmetic."""
  b, f, hop, k, sr = 32, 2500, 192, 200, 48000
  n = f * hop
  rng = np.random.default_rng(51)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = (100 + rng.standard_normal((b, f, 1))).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method='linear', use_angular_cumsum=True)
  args = tuple(torch.as_tensor(a, device=DEV) for a in (amps, hd, f0))
  full_t = synth(*args)
  full = npy(full_t)
  assert full.shape == (b, n) and np.isfinite(full).all()
  for r in range(b):                                   # compared on the device: 61 MB per row pair otherwise
    one = synth(*[a[r:r + 1] for a in args])
    assert bool(torch.equal(one, full_t[r:r + 1])), 'row %d run alone differs from the row in its batch' % r
  assert bool(torch.equal(synth(*[a[8:24] for a in args]), full_t[8:24]))
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  for r in (0, 9, 22, 31):
    sl = slice(r, r + 1)
    exact = _harmonic_exact(amps[sl], hd[sl], f0[sl], n, sr, 'linear')
    parity_check(full[sl], exact, HARM_TABLE_ATOL * scale, 'config 5 at batch 32, row %d vs exact arithmetic' % r)


def test_config3_spectral_loss_batch128(ddsp):
  """BASELINE configs[2] at ITS batch (the loss tests of test_gpu_parity.py stop at batch 32 and check two
  clips): losses.SpectralLoss (fft sizes 2048 .. 64, L1, mag + logmag: ddsp/losses.py:131-243, gin/models/ae.gin:36-41) on the
  synths' DAG output at batch 128.  The loss is a mean over [batch, frames, bins] per scale, so it is additive over rows: the
  batch's value is the mean of its sub-batches' and of its rows' values; six rows' values against the fp64 oracle; value AND
  gradient w.r.t. the audio (what a training step runs, one kernel) against the analytic fp64 gradient on two rows."""
  b, f, k, n = 128, 1000, 100, 64000
  rng = np.random.default_rng(31)
  harm = ddsp.synths.Harmonic(n_samples=n)
  fnoise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=3)

  def dag(seed, f0c):
    r = np.random.default_rng(seed)
    z = fnoise(r.standard_normal((b, f, 65)) + 2.0)
    return harm.call_add(r.standard_normal((b, f, 1)), r.standard_normal((b, f, k)), f0c + r.standard_normal((b, f, 1)), z)
  with torch.no_grad():
    audio, target = dag(32, 200.0), dag(33, 210.0)
  loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
  got = float(loss(target, audio))
  assert np.isfinite(got) and got > 0
  # the same call twice: the same bits (fp64 partials, fixed-order finish)
  assert float(loss(target, audio)) == got
  # additive over rows
  quarters = [float(loss(target[32 * q:32 * q + 32].contiguous(), audio[32 * q:32 * q + 32].contiguous())) for q in range(4)]
  np.testing.assert_allclose(got, np.mean(quarters), rtol=1e-6)
  rows = sorted(rng.choice(b, 6, replace=False).tolist())
  row_vals = []
  for r in rows:
    v = float(loss(target[r:r + 1].contiguous(), audio[r:r + 1].contiguous()))
    ref = float(O.spectral_loss(npy(target[r:r + 1]), npy(audio[r:r + 1]), logmag_weight=1.0, dtype=np.float64))
    np.testing.assert_allclose(v, ref, rtol=5e-5, err_msg='row %d' % r)
    row_vals.append(v)
  # each scale on its own adds up to the total, at this batch
  parts = sum(float(ddsp.losses.SpectralLoss(fft_sizes=(s,), logmag_weight=1.0)(target, audio)) for s in (2048, 1024, 512, 256, 128, 64))
  np.testing.assert_allclose(got, parts, rtol=1e-6)
  # value and gradient at batch 128: the value is the forward's, rows' gradients are their own (x 1 / batch: the mean).
  # (On BROADBAND signals, as the smaller gradient tests: the log-magnitude term's gradient is 1 / |bin|, and between the partials
  # of a tone the bins are 1e-3 .. 1e-4 of the peaks - where the fp32 transform's own rounding, 1e-7 of the PEAK, and the 1-ulp
  # v_sqrt / v_log are a per cent of the gradient.  Measured on the synths' output: the value right to 5e-5, the mag term's
  # gradient to 3e-3 of the tolerance, the logmag term's 0.7 % of the samples up to 30 tolerances off on the MI355X (8 under
  # the emulation's exact transcendentals).  That is the conditioning of the function in fp32, the reference's included; what
  # this test is for is the batch-128 schedule: 48 384 blocks, every output sample through an atomic.)
  audio = ddsp.core.tf_float32(0.3 * rng.standard_normal((b, n)))
  target = ddsp.core.tf_float32(0.8 * npy(audio) + 0.05 * rng.standard_normal((b, n)))
  got = float(loss(target, audio))
  ta = audio.clone().requires_grad_(True)
  val = loss(target, ta)
  val.backward()
  np.testing.assert_allclose(float(val.detach()), got, rtol=1e-6)
  g = npy(ta.grad)
  assert g.shape == (b, n) and np.isfinite(g).all()
  for r in rows[:2]:
    ref = O.spectral_loss_backward(npy(target[r:r + 1]), npy(audio[r:r + 1]), (2048, 1024, 512, 256, 128, 64), 1.0, 1.0) / b
    atol = 1e-9 + 2e-4 * np.abs(ref).max()
    err = np.abs(g[r:r + 1] - ref)
    # d|x|/dx is a sign: a bin whose two magnitudes agree to fp32 rounding may take the other sign.  Every bin of every size
    # weighs the same in the gradient (weight / (frames * bins) is the same for the six sizes), a sample's gradient is the sum
    # of ~5500 such terms (~75 single terms in magnitude), so ONE flipped bin is up to ~1 % of the largest gradient = ~45
    # tolerances, spread over its frame - 64 .. 2048 samples, up to 3 % of a row.  With ~4 M bins per row a handful flip in
    # any fp32 evaluation.  What is asserted is therefore the bulk (the median error far under the tolerance, nineteen samples
    # in twenty inside it) and a cap on the flips (60 tolerances, the error's energy 5e-3 of the gradient's): a fault of the
    # batch-128 schedule - a lost atomic, a wrong row - is an error of the size of the gradient itself.
    # (Measured on the MI355X over rows of several batches: 0.15 - 0.31 % of the samples outside, the largest 6 - 25
    # tolerances, energy ratio up to 1.1e-3.)
    rel_l2 = float(np.sqrt((err ** 2).sum() / (ref ** 2).sum()))
    assert (np.median(err) <= 0.25 * atol and (err > atol).mean() <= 0.05 and err.max() <= 60 * atol and rel_l2 <= 5e-3), (
        r, float(np.median(err)), float((err > atol).mean()), float(err.max()), atol, rel_l2)
