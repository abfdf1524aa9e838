"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors.  Tolerances (fp32 path, stated per SURVEY.md H1 / DESIGN.md "Parity contract"):

  HARM_TRUTH_ATOL  |ours - fp64 truth| <= 6e-5 * max(1, sum_k a_k)    (both phase modes; ~5x what the direct sum
                   measures on the MI355X, 1.3e-5; the wavetable kernel measures 4.7e-6 and is held to 2.5e-5 in
                   test_harm_table_*; every comparison is logged to $DDSP_PARITY_LOG when that is set)
  HARM_FAITHFUL    |ours - fp32 TF-faithful oracle| <= 2e-3 on clips <= 2k samples, where
                   the sequential fp32 cumsum has not yet drifted
  NOISE            |ours - fp64 oracle| <= 2e-6 + 1e-5 * max|ref|
  integer / RNG work (generated noise): bit exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, parity_check
from oracle import ddsp_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda'      # tests/test_simt_emulated.py re-runs a subset of these tests on host memory with DEV = 'cpu'

HARM_TRUTH_ATOL = 6e-5
HARM_TABLE_ATOL = 2.5e-5
HARM_FAITHFUL_ATOL = 2e-3


def harm_truth_check(ours, truth, scale=1.0, atol=None):
  parity_check(ours, truth, (HARM_TRUTH_ATOL if atol is None else atol) * scale)


@pytest.fixture(scope='module')
def ddsp():
  if not torch.cuda.is_available():
    pytest.skip('gpu tests need a GPU (run with -m gpu on an MI355X)')
  from ddsp_amd import build
  build.build()
  import ddsp_amd
  from ddsp_amd import _lib
  _lib.load()
  return ddsp_amd


@pytest.fixture(params=['auto', 'direct'])
def harm_kernel(request, ddsp):
  """Runs a test under both Harmonic kernels: 'auto' (matrix-core wavetables where they apply) and
  'direct' (every harmonic at every sample)."""
  old = ddsp.synths.Harmonic.kernel
  ddsp.synths.Harmonic.kernel = request.param
  yield request.param
  ddsp.synths.Harmonic.kernel = old


@pytest.fixture(params=['auto', 'vector'])
def noise_kernel(request, ddsp):
  """Runs a test under both FilteredNoise kernels: 'auto' (IR design and FIR on the matrix cores where the shape
  allows) and 'vector' (the FIR on the vector ALUs)."""
  old = ddsp.synths.FilteredNoise.kernel
  ddsp.synths.FilteredNoise.kernel = request.param
  yield request.param
  ddsp.synths.FilteredNoise.kernel = old


def npy(t):
  return t.detach().cpu().numpy()


def noise_tol(ref):
  return 2e-6 + 1e-5 * np.abs(ref).max()


HARMONIC_CASES = ['harmonic_window_cumsum', 'harmonic_window_angular',
                  'harmonic_linear_cumsum', 'harmonic_k100_live',
                  'harmonic_hop192_angular', 'harmonic_noscale_nonorm']
NOISE_CASES = ['noise_m65_w257', 'noise_m65_w0', 'noise_m33_w17', 'noise_m17_w16_even',
               'noise_ragged']


def make_harmonic(ddsp, g):
  return ddsp.synths.Harmonic(
      n_samples=int(g['n_samples']), sample_rate=int(g['sample_rate']),
      scale_fn=ddsp.core.exp_sigmoid if int(g['scale']) else None,
      normalize_below_nyquist=bool(g['normalize']), amp_resample_method=str(g['amp_method']),
      use_angular_cumsum=bool(g['angular']))


# ---- golden vectors (reference source files on the TF stand-in) ---------------------------
@pytest.mark.parametrize('name', HARMONIC_CASES)
def test_harmonic_golden(ddsp, harm_kernel, name):
  g = load_golden(name)
  synth = make_harmonic(ddsp, g)
  args = (g['amplitudes'], g['harmonic_distribution'], g['f0_hz'])
  out = synth(*args, return_outputs_dict=True)
  np.testing.assert_allclose(npy(out['controls']['amplitudes']), g['ctl_amplitudes'],
                             rtol=2e-5, atol=1e-9)
  np.testing.assert_allclose(npy(out['controls']['harmonic_distribution']),
                             g['ctl_harmonic_distribution'], rtol=2e-5, atol=1e-9)
  np.testing.assert_array_equal(npy(out['controls']['f0_hz']), g['f0_hz'])
  sig = npy(out['signal'])
  assert sig.shape == g['signal'].shape and sig.dtype == np.float32
  # vs the fp32 reference-source result (short clips: sequential cumsum has not drifted)
  assert np.abs(sig - g['signal']).max() <= HARM_FAITHFUL_ATOL
  # vs fp64 truth
  scale_fn = O.exp_sigmoid if int(g['scale']) else None
  truth = O.harmonic(g['amplitudes'], g['harmonic_distribution'], g['f0_hz'],
                     int(g['n_samples']), int(g['sample_rate']), scale_fn,
                     bool(g['normalize']), str(g['amp_method']), dtype=np.float64)
  amp_sum = max(1.0, float(np.abs(g['ctl_amplitudes']).max()))
  harm_truth_check(sig, truth, amp_sum)
  # unfused path (get_controls then get_signal; always the direct sum) gives the same audio as the fused call:
  # to round-off under 'direct', within the contract when the fused call ran on the wavetable kernel
  c = synth.get_controls(*args)
  sig2 = npy(synth.get_signal(**c))
  np.testing.assert_allclose(sig2, sig, rtol=0, atol=1e-6 if harm_kernel == 'direct' else 1e-4 * amp_sum)
  # with or without the controls dict: the same kernel, the same audio
  np.testing.assert_array_equal(npy(synth(*args)), sig)


@pytest.mark.parametrize('name', NOISE_CASES)
def test_filtered_noise_golden(ddsp, noise_kernel, name):
  g = load_golden(name)
  ws = int(g['window_size'])
  synth = ddsp.synths.FilteredNoise(n_samples=int(g['n_samples']), window_size=ws,
                                    scale_fn=ddsp.core.exp_sigmoid if int(g['scale']) else None)
  c = synth.get_controls(g['magnitudes'])
  np.testing.assert_allclose(npy(c['magnitudes']), g['ctl_magnitudes'], rtol=2e-5, atol=1e-9)
  ir = npy(ddsp.core.frequency_impulse_response(c['magnitudes'], window_size=ws))
  assert ir.shape == g['impulse_response'].shape
  assert np.abs(ir - g['impulse_response']).max() <= 2e-7 + 1e-5 * np.abs(g['impulse_response']).max()
  out = synth(g['magnitudes'], noise=g['noise'], return_outputs_dict=True)
  sig = npy(out['signal'])
  assert sig.shape == g['signal'].shape
  assert np.abs(sig - g['signal']).max() <= noise_tol(g['signal'])
  np.testing.assert_allclose(npy(out['controls']['magnitudes']), g['ctl_magnitudes'],
                             rtol=2e-5, atol=1e-9)
  # unfused: get_signal on controls, and core.frequency_filter (effects.FIRFilter maths)
  sig2 = npy(synth.get_signal(c['magnitudes'], noise=g['noise']))
  np.testing.assert_allclose(sig2, sig, rtol=0, atol=1e-7)
  sig3 = npy(ddsp.core.frequency_filter(g['noise'], c['magnitudes'], window_size=ws))
  np.testing.assert_allclose(sig3, sig, rtol=0, atol=1e-7)


def test_prepare_makes_the_constant_tables_ahead_of_the_first_launch(ddsp):
  """ddsp_prepare (the tables' first-use hipMalloc / hipMemcpy may not fall inside a graph capture): callable for any
  shape, idempotent, and the launches after it give what they give without it."""
  for k, m, ws in [(100, 65, 0), (128, 100, 0), (200, 256, 257), (300, 2, 0), (0, 0, 0), (60, 5000, 0)]:
    ddsp.core.prepare(k, m, ws)
    ddsp.core.prepare(k, m, ws)
  rng = np.random.default_rng(0)
  mags = rng.standard_normal((1, 10, 100)).astype(np.float32)
  noise = rng.uniform(-1, 1, (1, 640)).astype(np.float32)
  ours = npy(ddsp.synths.FilteredNoise(n_samples=640, window_size=0)(mags, noise=noise))
  ref = O.filtered_noise(mags, noise, 0, dtype=np.float64)
  assert np.abs(ours - ref).max() <= noise_tol(ref)


def test_synth_step_replays_from_a_captured_hip_graph(ddsp):
  """The C ABI neither allocates nor synchronises once ddsp_prepare has run, so a caller may capture its launches into a HIP
  graph and replay them (the way a serving loop amortises its launches): Harmonic + FilteredNoise + Add and the SpectralLoss
  captured with torch.cuda.CUDAGraph (plumbing: streams and capture are torch's), replayed on NEW inputs written into the
  captured buffers, against the eager calls - the same bits (supplied noise: the generated stream's call counter is a launch
  argument, a captured launch would replay one counter)."""
  if not torch.cuda.is_available():
    pytest.skip('needs the GPU')
  b, f, k, m, n = 4, 100, 100, 65, 6400
  ddsp.core.prepare(k, m, 0)
  rng = np.random.default_rng(123)
  T = ddsp.core.tf_float32

  def fresh():
    return (T(rng.standard_normal((b, f, 1))), T(rng.standard_normal((b, f, k))), T(200.0 + rng.standard_normal((b, f, 1))),
            T(rng.standard_normal((b, f, m))), T(rng.uniform(-1, 1, (b, n))), T(rng.standard_normal((b, n))))
  amps, hd, f0, mags, noise, target = fresh()
  harm = ddsp.synths.Harmonic(n_samples=n)
  fnoise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)
  loss = ddsp.losses.SpectralLoss(fft_sizes=(512, 256, 64), logmag_weight=1.0)

  def step():
    audio = harm.call_add(amps, hd, f0, fnoise(mags, noise=noise))
    return audio, loss(target, audio)
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):                     # warm-up on the capture stream: workspaces are per stream, made here
    for _ in range(3):
      step()
  torch.cuda.current_stream().wait_stream(side)
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, stream=side):
    g_audio, g_loss = step()
  for _ in range(3):
    new = fresh()
    for dst, src in zip((amps, hd, f0, mags, noise, target), new):
      dst.copy_(src)
    graph.replay()
    torch.cuda.synchronize()
    e_audio, e_loss = step()
    np.testing.assert_array_equal(npy(g_audio), npy(e_audio))
    assert float(g_loss) == float(e_loss)


def test_add_golden(ddsp):
  g = load_golden('add')
  np.testing.assert_array_equal(npy(ddsp.processors.Add()(g['signal_one'], g['signal_two'])),
                                g['signal'])


# ---- canonical shape (ae.gin: F=1000, K=100, M=65, N=64000, 16 kHz) at small batch ---------
def canonical_inputs(batch, seed=0, f0_center=70.0, n_frames=1000, k=100, m=65):
  rng = np.random.default_rng(seed)
  return dict(
      amplitudes=rng.standard_normal((batch, n_frames, 1)).astype(np.float32),
      harmonic_distribution=rng.standard_normal((batch, n_frames, k)).astype(np.float32),
      f0_hz=(f0_center + rng.standard_normal((batch, n_frames, 1))).astype(np.float32),
      magnitudes=rng.standard_normal((batch, n_frames, m)).astype(np.float32))


@pytest.mark.parametrize('f0_center', [70.0, 200.0])
def test_harmonic_canonical_vs_truth_and_faithful(ddsp, harm_kernel, f0_center):
  x = canonical_inputs(2, seed=1, f0_center=f0_center)
  args = (x['amplitudes'], x['harmonic_distribution'], x['f0_hz'])
  ours = npy(ddsp.synths.Harmonic()(*args))
  truth = O.harmonic(*args, dtype=np.float64)
  faithful_seq = O.harmonic(*args)                                  # tf.cumsum fp32 path
  faithful_ang = O.harmonic(*args, use_angular_cumsum=True)
  err_ours = np.abs(ours - truth).max()
  err_seq = np.abs(faithful_seq - truth).max()
  err_ang = np.abs(faithful_ang - truth).max()
  print('canonical f0~%g: |ours-truth| %.2e  |tf.cumsum fp32-truth| %.2e  |angular fp32-truth| %.2e'
        % (f0_center, err_ours, err_seq, err_ang))
  harm_truth_check(ours, truth, 2.0)                 # amplitudes are <= 2 (exp_sigmoid max)
  assert err_ours <= err_seq and err_ours <= err_ang   # closer to truth than TF's own fp32 paths
  # default path: direct parity on the prefix where fp32 sequential cumsum has not drifted
  assert np.abs(ours[:, :2000] - faithful_seq[:, :2000]).max() <= HARM_FAITHFUL_ATOL
  # angular path: direct parity over the full clip, tolerance 5e-2 * sum_k a_k
  ours_ang = npy(ddsp.synths.Harmonic(use_angular_cumsum=True)(*args))
  assert np.abs(ours_ang - faithful_ang).max() <= 5e-2 * 2.0
  np.testing.assert_array_equal(ours_ang, ours)


def test_filtered_noise_canonical_vs_oracle(ddsp, noise_kernel):
  x = canonical_inputs(2, seed=2)
  noise = np.random.default_rng(3).uniform(-1, 1, (2, 64000)).astype(np.float32)
  for ws in (0, 257):                      # ae.gin uses 0; the class default is 257 (same IR)
    ours = npy(ddsp.synths.FilteredNoise(window_size=ws)(x['magnitudes'], noise=noise))
    ref = O.filtered_noise(x['magnitudes'], noise, ws, dtype=np.float64)
    assert np.abs(ours - ref).max() <= noise_tol(ref)


# ---- generated noise: integer work, bit exact ------------------------------------------------
def test_generated_noise_is_bit_exact_and_fused_path_matches_injection(ddsp, noise_kernel):
  b, n = 3, 6400
  dev_noise = npy(ddsp.core.uniform_noise(b, n, seed=1234, batch_offset=5))
  np.testing.assert_array_equal(dev_noise, O.device_uniform_noise(b, n, 1234, 5))
  mags = np.random.default_rng(4).standard_normal((b, 100, 65)).astype(np.float32)
  assert ddsp.synths.FilteredNoise().noise_bits == 23       # the reference's resolution is the default
  # (2048-level noise is carried in ONE fp16 plane whether generated or supplied: same bits; 23-bit samples: within the parity
  # tolerance, test_generated_noise_contract_both_resolutions)
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=99, noise_bits=11)
  gen = npy(synth(mags))                                    # call counter 0 -> key (99, 0)
  inj = npy(synth(mags, noise=O.device_uniform_noise(b, n, 99, 0, noise_bits=11)))
  np.testing.assert_array_equal(gen, inj)
  gen2 = npy(synth(mags))                                   # stateful like tf.random: differs
  assert np.abs(gen2 - gen).max() > 0
  assert abs(gen.mean()) < 1e-3 and gen.std() > 0


def noise_statistics(x):
  """What 'uniform on (-1, 1), white' means in numbers, for one long row: mean, variance, the largest autocorrelation at
  lags 1 .. 128 (the FIR has 128 taps: correlations inside its reach are what would colour the output) and the
  Kolmogorov-Smirnov distance to U(-1, 1)."""
  x = x.astype(np.float64)
  n = x.size
  xc = x - x.mean()
  spec = np.fft.rfft(xc, 2 * n)
  ac = np.fft.irfft(spec * np.conj(spec))[:129] / (xc.var() * n)
  xs = np.sort(x)
  cdf = (xs + 1.0) / 2.0
  ks = max(np.abs(cdf - np.arange(1, n + 1) / n).max(), np.abs(cdf - np.arange(0, n) / n).max())
  return dict(mean=x.mean(), var=x.var(), max_ac=np.abs(ac[1:]).max(), ks=ks)


@pytest.mark.parametrize('noise_bits', [11, 23])
def test_generated_noise_contract_both_resolutions(ddsp, noise_kernel, noise_bits):
  """The generated-noise contract of include/ddsp_amd.h at both settings of FilteredNoise(noise_bits=): the stream is
  bit-exact against the oracle's restatement, the synth run on it equals the synth handed the same samples (bit for bit
  where the kernel carries them the same way: 11-bit samples ARE their fp16 hi part; 23-bit samples made on chip and
  supplied ones go through the same hi / lo planes, but supplied noise is normalised by its row's power of two first -
  a no-op in exact arithmetic - so that comparison has the parity tolerance), the backward pass regenerates the same samples,
  and the samples are what the reference's tf.random.uniform(-1, 1) (ddsp/synths.py:192-193) promises statistically."""
  b, n = 2, 6400
  dev_noise = npy(ddsp.core.uniform_noise(b, n, seed=1234, batch_offset=5, noise_bits=noise_bits))
  ref_noise = O.device_uniform_noise(b, n, 1234, 5, noise_bits=noise_bits)
  np.testing.assert_array_equal(dev_noise, ref_noise)
  levels = np.unique(dev_noise).size
  assert levels <= 2048 if noise_bits == 11 else levels > 6000        # 12 800 samples: 23-bit samples hardly ever repeat
  rng = np.random.default_rng(4)
  mags = rng.standard_normal((b, 100, 65)).astype(np.float32)
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=99, noise_bits=noise_bits)
  tm = ddsp.core.tf_float32(mags).requires_grad_(True)
  gen_t = synth(tm)                                                   # call counter 0 -> key (99, 0)
  g = rng.standard_normal((b, n)).astype(np.float32)
  gen_t.backward(ddsp.core.tf_float32(g))
  gen = npy(gen_t)
  noise = O.device_uniform_noise(b, n, 99, 0, noise_bits=noise_bits)
  inj = npy(synth(mags, noise=noise))
  ref = O.filtered_noise(mags, noise, 0, dtype=np.float64)
  assert np.abs(gen - ref).max() <= noise_tol(ref)
  if noise_bits == 11:
    np.testing.assert_array_equal(gen, inj)
  else:
    assert np.abs(gen - inj).max() <= noise_tol(ref)
  gref = O.filtered_noise_backward(mags, noise, g, 0, O.exp_sigmoid)
  np.testing.assert_allclose(npy(tm.grad), gref, rtol=0, atol=1e-6 + 2e-5 * np.abs(gref).max())
  # the other resolution is another stream
  other = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=99, noise_bits=34 - noise_bits)
  assert np.abs(npy(other(mags)) - gen).max() > 1e-3
  # statistics on 10^6 samples of one row (sigma of the mean 5.8e-4, of the variance 3e-4, of an autocorrelation 1e-3;
  # KS at the 0.1 % level 1.95e-3 - the 2048-level staircase itself is 2.4e-4 away from the uniform law)
  st = noise_statistics(npy(ddsp.core.uniform_noise(1, 1000000, seed=2024, noise_bits=noise_bits))[0])
  assert abs(st['mean']) < 3e-3 and abs(st['var'] - 1.0 / 3.0) < 1.5e-3, st
  assert st['max_ac'] < 6e-3 and st['ks'] < 2.5e-3, st
  with pytest.raises(ValueError):
    ddsp.synths.FilteredNoise(noise_bits=16)


def test_supplied_audio_and_filters_at_any_scale(ddsp):
  """core.fft_convolve, frequency_filter, frequency_impulse_response and FilteredNoise with supplied noise
  run on the matrix cores with fp16 hi / lo operands; the reference (ddsp/core.py:1382-1473, tf.signal's FFTs) is fp32 and
  scale invariant.  int16-range audio (x 32768) overflowed fp16 to NaN, an impulse response of 1e-9 lost 8 %.  Every operand
  the caller supplies is normalised by a power of two per tile now: out(s x, t h) / (s t) matches the fp64 result at the
  tolerance of s = t = 1."""
  rng = np.random.default_rng(5)
  b, f, l, n = 2, 25, 97, 3200
  audio = rng.standard_normal((b, n)).astype(np.float32)
  ir = (rng.standard_normal((b, f, l)) / np.sqrt(l)).astype(np.float32)
  ref = O.fft_convolve(audio, ir, dtype=np.float64)
  for sa, sh in [(32768.0, 1.0), (1.0, 1e-7), (1.0, 1e-9), (1e-9, 1.0), (1e5, 1e4), (3e-12, 2e-11)]:
    got = npy(ddsp.core.fft_convolve((audio * np.float32(sa)).astype(np.float32), (ir * np.float32(sh)).astype(np.float32)))
    ref_s = O.fft_convolve((audio * np.float32(sa)).astype(np.float32), (ir * np.float32(sh)).astype(np.float32), dtype=np.float64)
    assert np.isfinite(got).all(), (sa, sh)
    np.testing.assert_allclose(got / (sa * sh), ref_s / (sa * sh), rtol=0, atol=2e-6 + 1e-5 * np.abs(ref).max(), err_msg=str((sa, sh)))
  # one filter for the whole batch, 128 taps on frames of 64 (the canonical filter's geometry through fft_convolve)
  ir1 = (rng.standard_normal((1, 50, 128)) / 11.0).astype(np.float32)
  ref1 = O.fft_convolve(audio, ir1, dtype=np.float64)
  for sa, sh in [(32768.0, 1e-6), (1e-8, 1e3)]:
    got = npy(ddsp.core.fft_convolve((audio * np.float32(sa)).astype(np.float32), (ir1 * np.float32(sh)).astype(np.float32)))
    np.testing.assert_allclose(got / (sa * sh), ref1, rtol=0, atol=2e-6 + 2e-5 * np.abs(ref1).max(), err_msg=str((sa, sh)))
  # magnitudes of any size: frequency_impulse_response (raw magnitudes, no exp_sigmoid) and the synth with scale_fn=None
  for m, ws in [(65, 0), (100, 0), (33, 17)]:
    mags = np.abs(rng.standard_normal((b, 20, m))).astype(np.float32)
    ir_ref = O.frequency_impulse_response(mags, ws, dtype=np.float64)
    for sm in (1e-8, 1.0, 3e5):
      got = npy(ddsp.core.frequency_impulse_response((mags * np.float32(sm)).astype(np.float32), window_size=ws))
      np.testing.assert_allclose(got / sm, ir_ref, rtol=0, atol=2e-7 + 1e-5 * np.abs(ir_ref).max(), err_msg=str((m, ws, sm)))
  mags = np.abs(rng.standard_normal((b, 50, 65))).astype(np.float32)
  noise = rng.uniform(-1, 1, (b, n)).astype(np.float32)
  ref_fn = O.filtered_noise(mags, noise, 0, None, dtype=np.float64)
  for kernel in ('auto', 'vector'):
    for sm, sx in [(1.0, 1.0), (1e-7, 32768.0), (2e4, 1e-9)]:
      synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, scale_fn=None)
      synth.kernel = kernel
      got = npy(synth((mags * np.float32(sm)).astype(np.float32), noise=(noise * np.float32(sx)).astype(np.float32)))
      np.testing.assert_allclose(got / (sm * sx), ref_fn, rtol=0, atol=2e-6 + 1e-5 * np.abs(ref_fn).max(), err_msg=str((kernel, sm, sx)))
  # supplied noise at any scale through the canonical kernel (exp_sigmoid magnitudes)
  mags = rng.standard_normal((b, 50, 65)).astype(np.float32)
  ref_c = O.filtered_noise(mags, noise, 0, dtype=np.float64)
  for sx in (32768.0, 1e-9):
    got = npy(ddsp.synths.FilteredNoise(n_samples=n, window_size=0)(mags, noise=(noise * np.float32(sx)).astype(np.float32)))
    np.testing.assert_allclose(got / sx, ref_c, rtol=0, atol=noise_tol(ref_c), err_msg=str(sx))


# ---- edge cases --------------------------------------------------------------------------------
@pytest.mark.parametrize('k,n_frames,hop,sr', [
    (1, 8, 64, 16000), (64, 8, 64, 16000), (65, 8, 128, 16000), (300, 6, 64, 48000),
    (20, 7, 100, 16000), (40, 5, 192, 48000), (30, 3, 320, 16000), (10, 1, 256, 16000),
    (100, 33, 64, 16000)])
@pytest.mark.parametrize('method', ['window', 'linear'])
def test_harmonic_edge_shapes(ddsp, harm_kernel, k, n_frames, hop, sr, method):
  rng = np.random.default_rng(k + hop)
  b, n = 2, n_frames * hop
  amps = rng.standard_normal((b, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((b, n_frames, k)).astype(np.float32)
  f0 = rng.uniform(60, 0.6 * sr / max(k, 2) + 80, (b, n_frames, 1)).astype(np.float32)
  ours = npy(ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)(
      amps, hd, f0))
  truth = O.harmonic(amps, hd, f0, n, sr, amp_resample_method=method, dtype=np.float64)
  harm_truth_check(ours, truth, 2.0)


def test_harmonic_silent_above_nyquist_and_exact_boundary(ddsp, harm_kernel):      # core_test.py:484-503
  for sr in (4000, 16000, 44100):
    for ratio in (1.0, 1.1, 1.5, 2.0):              # 1.0: f == sr/2 exactly is masked (>=)
      f0 = np.full((2, 10, 1), ratio * sr / 2.0, np.float32)
      out = npy(ddsp.synths.Harmonic(n_samples=640, sample_rate=sr, scale_fn=None)(
          np.ones((2, 10, 1), np.float32), np.ones((2, 10, 3), np.float32), f0))
      assert np.all(out == 0.0)
  # f0 = 80 Hz: harmonic 100 sits exactly on Nyquist -> removed; 99 harmonics remain
  k = 100
  amps, hd = np.ones((1, 10, 1), np.float32), np.ones((1, 10, k), np.float32)
  f0 = np.full((1, 10, 1), 80.0, np.float32)
  ours = npy(ddsp.synths.Harmonic(n_samples=640, scale_fn=None)(amps, hd, f0))
  truth = O.harmonic(amps, hd, f0, 640, scale_fn=None, dtype=np.float64)
  harm_truth_check(ours, truth)


def test_harmonic_nyquist_crossing_between_frames(ddsp, harm_kernel):
  """Audio-rate mask on the interpolated frequency (core.py:942-944)."""
  n_frames, hop, k = 12, 64, 8
  f0 = np.linspace(700.0, 1400.0, n_frames, dtype=np.float32)[None, :, None]   # k*f0 crosses 8 kHz
  amps, hd = np.ones((1, n_frames, 1), np.float32), np.full((1, n_frames, k), 1.0 / k, np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n_frames * hop, scale_fn=None,
                               normalize_below_nyquist=False)
  ours = npy(synth(amps, hd, f0))
  truth = O.harmonic(amps, hd, f0, n_frames * hop, scale_fn=None, normalize_below_nyquist=False,
                     dtype=np.float64)
  faithful = O.harmonic(amps, hd, f0, n_frames * hop, scale_fn=None,
                        normalize_below_nyquist=False)
  harm_truth_check(ours, truth)
  assert np.abs(ours - faithful).max() <= HARM_FAITHFUL_ATOL


@pytest.mark.parametrize('m,ws,n_frames,n', [(3, 0, 4, 64), (9, 0, 7, 100), (65, 0, 1000, 64000), (65, 0, 3, 192), (65, 0, 1, 300),
                                             (65, 257, 10, 640), (129, 65, 5, 1000),
                                             (1025, 257, 1, 4096), (33, 17, 16, 4096)])
def test_filtered_noise_edge_shapes(ddsp, noise_kernel, m, ws, n_frames, n):
  rng = np.random.default_rng(m + n)
  mags = rng.standard_normal((2, n_frames, m)).astype(np.float32)
  noise = rng.uniform(-1, 1, (2, n)).astype(np.float32)
  ours = npy(ddsp.synths.FilteredNoise(n_samples=n, window_size=ws)(mags, noise=noise))
  ref = O.filtered_noise(mags, noise, ws, dtype=np.float64)
  assert ours.shape == ref.shape
  assert np.abs(ours - ref).max() <= noise_tol(ref)


def test_delay_compensation_identity_filter(ddsp):                   # core_test.py:759-785
  n = 4 * 1024
  audio = np.sin(np.linspace(0, 200.0, n))[None, :].astype(np.float32)
  for gain in (1.0, 0.1):
    mags = gain * np.ones([1, 1025], np.float32)
    out = npy(ddsp.core.frequency_filter(audio, mags, window_size=257))
    assert np.abs(out - gain * audio).mean() <= 1e-3


def test_fft_convolve_broadcast_ir_and_errors(ddsp):
  rng = np.random.default_rng(7)
  audio = rng.standard_normal((3, 500)).astype(np.float32)
  ir = rng.standard_normal((1, 5, 31)).astype(np.float32)
  ours = npy(ddsp.core.fft_convolve(audio, ir))
  ref = O.fft_convolve(audio, ir, dtype=np.float64)
  assert np.abs(ours - ref).max() <= 1e-5 * np.abs(ref).max()
  ours0 = npy(ddsp.core.fft_convolve(audio, ir, delay_compensation=0))
  ref0 = O.fft_convolve(audio, ir, delay_compensation=0, dtype=np.float64)
  assert np.abs(ours0 - ref0).max() <= 1e-5 * np.abs(ref0).max()
  with pytest.raises(ValueError, match='Batch size'):
    ddsp.core.fft_convolve(audio, np.ones((2, 5, 31), np.float32))
  with pytest.raises(ValueError, match='do not match'):
    ddsp.core.fft_convolve(np.ones((1, 100), np.float32), np.ones((1, 30, 5), np.float32))
  with pytest.raises(ValueError, match='Padding'):
    ddsp.core.fft_convolve(audio, ir, padding='bogus')


def test_exp_sigmoid_matches_oracle(ddsp):
  x = np.linspace(-30, 30, 4001).astype(np.float32)
  ours = npy(ddsp.core.exp_sigmoid(x))
  np.testing.assert_allclose(ours, O.exp_sigmoid(x, dtype=np.float64), rtol=2e-5, atol=1e-12)


def test_reference_shape_tests(ddsp):                       # synths_test.py:23-50
  h = ddsp.synths.Harmonic(n_samples=64000, sample_rate=16000, scale_fn=None,
                           normalize_below_nyquist=True)
  batch = 3
  out = h(np.zeros((batch, 16000, 1), np.float32) + 1.0,
          np.zeros((batch, 16000, 16), np.float32) + 1.0,
          np.zeros((batch, 16000, 1), np.float32) + 440.0)
  assert tuple(out.shape) == (3, 64000)
  z = ddsp.synths.FilteredNoise(n_samples=16000)(np.zeros((batch, 1000, 100), np.float32) + 3.0)
  assert tuple(z.shape) == (3, 16000)


# ---- full-size properties (BASELINE configs; the oracle is too slow there) ---------------------
def test_full_size_properties_batch32(ddsp, harm_kernel):
  b = 32
  x = canonical_inputs(b, seed=5)
  harm, fn = ddsp.synths.Harmonic(), ddsp.synths.FilteredNoise(window_size=0)
  args = (x['amplitudes'], x['harmonic_distribution'], x['f0_hz'])
  full = npy(harm(*args))
  assert full.shape == (b, 64000) and np.isfinite(full).all()
  # batch rows are independent: any sub-batch reproduces its rows bit for bit
  half = npy(harm(*[a[16:] for a in args]))
  np.testing.assert_array_equal(half, full[16:])
  one = npy(harm(*[a[7:8] for a in args]))
  np.testing.assert_array_equal(one, full[7:8])
  # spot rows against fp64 truth
  truth = O.harmonic(*[a[7:8] for a in args], dtype=np.float64)
  harm_truth_check(one, truth, 2.0)
  # |audio| <= sum_k a_k = amplitude (distribution is normalised)
  amp = O.exp_sigmoid(x['amplitudes'], dtype=np.float64).max()
  assert np.abs(full).max() <= amp * (1 + 1e-5)

  # FIR: linear in the noise, rows independent, time-shift structure via impulse response
  rng = np.random.default_rng(6)
  n1 = rng.uniform(-1, 1, (b, 64000)).astype(np.float32)
  n2 = rng.uniform(-1, 1, (b, 64000)).astype(np.float32)
  y1, y2 = npy(fn(x['magnitudes'], noise=n1)), npy(fn(x['magnitudes'], noise=n2))
  y12 = npy(fn(x['magnitudes'], noise=n1 + n2))
  assert np.abs(y12 - (y1 + y2)).max() <= 1e-5 * np.abs(y12).max() + 1e-7
  np.testing.assert_array_equal(npy(fn(x['magnitudes'][3:5], noise=n1[3:5])), y1[3:5])
  ref = O.filtered_noise(x['magnitudes'][3:4], n1[3:4], 0, dtype=np.float64)
  assert np.abs(y1[3:4] - ref).max() <= noise_tol(ref)
  # an impulse at sample t0 reads out frame(t0)'s impulse response, delayed by start=62
  imp = np.zeros((b, 64000), np.float32)
  t0 = 64 * 500 + 13
  imp[:, t0] = 1.0
  yi = npy(fn(x['magnitudes'], noise=imp))
  c = fn.get_controls(x['magnitudes'])
  ir = npy(ddsp.core.frequency_impulse_response(c['magnitudes'], 0))[:, 500, :]
  np.testing.assert_allclose(yi[:, t0 - 62:t0 - 62 + 128], ir, rtol=0, atol=1e-7)
  assert np.all(yi[:, :t0 - 62] == 0) and np.all(yi[:, t0 - 62 + 128:] == 0)


# ---- next rows of SURVEY 8(f): ProcessorGroup plumbing over the HIP processors ------------------
def test_processor_group_harmonic_noise_add(ddsp):           # gin/models/ae.gin:49-56 DAG
  n_frames, n = 100, 6400
  x = canonical_inputs(2, seed=9, n_frames=n_frames)
  features = {'amps': x['amplitudes'], 'harmonic_distribution': x['harmonic_distribution'],
              'f0_hz': x['f0_hz'], 'magnitudes': x['magnitudes']}
  harmonic = ddsp.synths.Harmonic(n_samples=n, name='harmonic')
  noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, name='filtered_noise', seed=3)
  add = ddsp.processors.Add(name='add')
  dag = [(harmonic, ['amps', 'harmonic_distribution', 'f0_hz']),
         (noise, ['magnitudes']),
         (add, ['filtered_noise/signal', 'harmonic/signal'])]
  group = ddsp.processors.ProcessorGroup(dag=dag, name='processor_group')
  out = group(features, return_outputs_dict=True)
  c = out['controls']
  for key in ['inputs', 'harmonic', 'filtered_noise', 'add', 'out']:
    assert key in c
  h, z = npy(c['harmonic']['signal']), npy(c['filtered_noise']['signal'])
  np.testing.assert_array_equal(npy(out['signal']), h + z)
  truth = O.harmonic(x['amplitudes'], x['harmonic_distribution'], x['f0_hz'], n, dtype=np.float64)
  harm_truth_check(h, truth, 2.0)
  zref = O.filtered_noise(x['magnitudes'], O.device_uniform_noise(2, n, 3, 0), 0, dtype=np.float64)
  assert np.abs(z - zref).max() <= noise_tol(zref)
  np.testing.assert_allclose(npy(c['harmonic']['controls']['harmonic_distribution']).sum(-1), 1.0,
                             rtol=1e-5)


def test_harmonic_48k_200_harmonics(ddsp, harm_kernel):                   # BASELINE config 5 shape, one short clip
  n_frames, hop, k, sr = 250, 192, 200, 48000
  rng = np.random.default_rng(11)
  amps = rng.standard_normal((2, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((2, n_frames, k)).astype(np.float32)
  f0 = (100 + rng.standard_normal((2, n_frames, 1))).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n_frames * hop, sample_rate=sr, amp_resample_method='linear',
                               use_angular_cumsum=True)
  ours = npy(synth(amps, hd, f0))
  truth = O.harmonic(amps, hd, f0, n_frames * hop, sr, amp_resample_method='linear', dtype=np.float64)
  harm_truth_check(ours, truth, 2.0)


def test_standalone_resample_and_normalize(ddsp):            # core_test.py:153-267, golden 'resample'
  g = load_golden('resample')
  np.testing.assert_allclose(npy(ddsp.core.resample(g['x'], 576, method='window')), g['window_576'],
                             rtol=0, atol=2e-6)
  for n in (576, 1728, 900):
    np.testing.assert_allclose(npy(ddsp.core.resample(g['x'], n, method='linear')), g['linear_%d' % n],
                               rtol=0, atol=1e-6)
  x1 = np.array([0.0, 1.0, 0.5, -0.3, 0.8], np.float32)
  for method in ('linear', 'window'):                        # core_test.py:242-267 sub-sampling
    y = npy(ddsp.core.resample(x1, 16000, method=method))
    np.testing.assert_allclose(y[np.arange(5) * 3200], x1, atol=1e-3)
  with pytest.raises(ValueError, match='is invalid'):
    ddsp.core.resample(g['x'], 576, method='bogus')
  with pytest.raises(ValueError, match='divisible'):
    ddsp.core.upsample_with_windows(np.ones((1, 10, 1), np.float32), 105)
  with pytest.raises(ValueError, match='3 dimensions'):
    ddsp.core.upsample_with_windows(np.ones((2, 10), np.float32), 100)
  rng = np.random.default_rng(12)
  hd = np.abs(rng.standard_normal((2, 9, 20))).astype(np.float32)
  f0 = rng.uniform(300, 900, (2, 9, 1)).astype(np.float32)
  np.testing.assert_allclose(npy(ddsp.core.normalize_harmonics(hd, f0, 16000)),
                             O.normalize_harmonics(hd, f0, 16000), rtol=2e-6, atol=1e-9)
  np.testing.assert_allclose(npy(ddsp.core.normalize_harmonics(hd)), O.normalize_harmonics(hd),
                             rtol=2e-6, atol=1e-9)
  assert ddsp.core.get_fft_size(64, 128) == 256


def test_standalone_oscillator_bank(ddsp):                    # core_test.py:460-503
  rng = np.random.default_rng(13)
  b, n, k, sr = 2, 3000, 70, 16000
  freq = np.cumsum(rng.standard_normal((b, n, k)), axis=1).astype(np.float32) * 2 + \
      rng.uniform(100, 9000, (b, 1, k)).astype(np.float32)
  amp = rng.uniform(0, 1, (b, n, k)).astype(np.float32)
  truth = O.oscillator_bank(freq.astype(np.float64), amp.astype(np.float64), sr)
  ours = npy(ddsp.core.oscillator_bank(freq, amp, sr))
  assert ours.shape == (b, n) and np.abs(ours - truth).max() <= 1e-4 * k
  each = npy(ddsp.core.oscillator_bank(freq, amp, sr, sum_sinusoids=False))
  truth_each = O.oscillator_bank(freq.astype(np.float64), amp.astype(np.float64), sr, sum_sinusoids=False)
  assert each.shape == (b, n, k) and np.abs(each - truth_each).max() <= 1e-4
  for srate in (4000, 16000, 44100):                         # silent at and above Nyquist
    for ratio in (1.0, 1.1, 2.0):
      f = np.full((2, 1000, 3), ratio * srate / 2.0, np.float32)
      assert np.all(npy(ddsp.core.oscillator_bank(f, np.ones_like(f), srate)) == 0.0)


@pytest.mark.parametrize('n_frames,n', [(40, 2560 - 17), (62, 3968), (63, 4032), (125, 8000 - 63), (1, 64)])
def test_filtered_noise_fused_tile_edges(ddsp, noise_kernel, n_frames, n):
  """Canonical M=65 / frame 64 shapes around the fused kernel's 62-frame tile boundary, ragged N."""
  rng = np.random.default_rng(n)
  mags = rng.standard_normal((3, n_frames, 65)).astype(np.float32)
  noise = rng.uniform(-1, 1, (3, n)).astype(np.float32)
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=257)
  out = synth(mags, noise=noise, return_outputs_dict=True)
  ref = O.filtered_noise(mags, noise, 257, dtype=np.float64)
  assert np.abs(npy(out['signal']) - ref).max() <= noise_tol(ref)
  np.testing.assert_allclose(npy(out['controls']['magnitudes']),
                             O.filtered_noise_get_controls(mags)['magnitudes'], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('batch,n_frames', [(1, 8), (1, 9), (3, 17), (2, 130), (5, 1000)])
def test_harmonic_fused_unit_edges(ddsp, harm_kernel, batch, n_frames):
  """Frame counts around the fused kernel's 8-frame units (partial last unit, halo at F-1)."""
  rng = np.random.default_rng(n_frames)
  k, hop = 100, 64
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  f0 = (70 + 30 * rng.standard_normal((batch, n_frames, 1))).astype(np.float32)
  n = n_frames * hop
  synth = ddsp.synths.Harmonic(n_samples=n)
  out = synth(amps, hd, f0, return_outputs_dict=True)
  if n_frames <= 130:
    truth = O.harmonic(amps, hd, f0, n, dtype=np.float64)
    c = O.harmonic_get_controls(amps, hd, f0)
  else:                                      # keep the oracle cheap: check two rows
    truth = O.harmonic(amps[:2], hd[:2], f0[:2], n, dtype=np.float64)
    c = O.harmonic_get_controls(amps[:2], hd[:2], f0[:2])
  nb = truth.shape[0]
  harm_truth_check(npy(out['signal'])[:nb], truth, 2.0)
  np.testing.assert_allclose(npy(out['controls']['harmonic_distribution'])[:nb],
                             c['harmonic_distribution'], rtol=2e-5, atol=1e-9)
  np.testing.assert_allclose(npy(out['controls']['amplitudes'])[:nb], c['amplitudes'], rtol=2e-5)
  # repeated launches reuse the self-resetting work counters; without the controls dict the call may
  # take the other kernel ('auto'): same audio within the tolerance, bit-identical run to run
  again = npy(synth(amps, hd, f0))
  harm_truth_check(again[:nb], truth, 2.0)
  np.testing.assert_array_equal(again, npy(synth(amps, hd, f0)))


# ---- harm_table_kernel: chunks of 15 frames, K / tap-count variants, the numpy model of the method ------
@pytest.mark.parametrize('batch,n_frames,k,hop,f0_center,f0_spread', [
    (1, 1, 100, 64, 70.0, 1.0), (2, 15, 100, 64, 70.0, 1.0), (2, 16, 100, 64, 200.0, 30.0),
    (3, 31, 100, 64, 70.0, 20.0), (1, 46, 60, 64, 130.0, 5.0), (2, 30, 4, 64, 1500.0, 800.0),
    (1, 20, 64, 128, 100.0, 10.0), (2, 17, 68, 64, 110.0, 3.0), (1, 33, 104, 192, 70.0, 2.0),
    (2, 19, 128, 64, 60.0, 1.0), (1, 250, 100, 64, 400.0, 300.0), (40, 7, 100, 64, 70.0, 1.0)])
@pytest.mark.parametrize('method', ['window', 'linear'])
def test_harmonic_table_kernel_vs_truth_model_and_direct(ddsp, batch, n_frames, k, hop, f0_center, f0_spread, method):
  from wavetable_model import harmonic_table_model
  rng = np.random.default_rng(n_frames * 1000 + k)
  n, sr = n_frames * hop, 16000
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  f0 = np.abs(f0_center + f0_spread * rng.standard_normal((batch, n_frames, 1))).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)
  assert synth.kernel == 'auto'
  ours = npy(synth(amps, hd, f0))
  synth.kernel = 'direct'
  direct = npy(synth(amps, hd, f0))
  nb = min(batch, 3)
  truth = O.harmonic(amps[:nb], hd[:nb], f0[:nb], n, sr, amp_resample_method=method, dtype=np.float64)
  err, err_direct = np.abs(ours[:nb] - truth).max(), np.abs(direct[:nb] - truth).max()
  print('table %.2e direct %.2e' % (err, err_direct))
  harm_truth_check(ours[:nb], truth, 2.0, atol=HARM_TABLE_ATOL)      # the window's aliasing is <= 6.5e-6 per harmonic
  assert np.abs(ours - direct).max() <= HARM_TRUTH_ATOL * 2.0
  model = harmonic_table_model(amps[:nb], hd[:nb], f0[:nb], n, sr, W=6 if k <= 100 else 8,
                               amp_linear=(method == 'linear'))
  assert np.abs(ours[:nb] - model).max() <= 4e-6 * 2.0     # same method, fp32 summation order apart


@pytest.mark.parametrize('batch,n_frames,k', [(2, 31, 100), (1, 15, 128), (3, 47, 60), (1, 1000, 100)])
def test_harmonic_table_kernel_controls_dict(ddsp, batch, n_frames, k):
  """return_outputs_dict=True (how dags.py:171-173 calls every processor) on the wavetable kernel: the controls are
  written by phase A, chunk by chunk (15 frames), without the halo rows; same audio as without the dict."""
  rng = np.random.default_rng(n_frames + k)
  hop, sr = 64, 16000
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  f0 = np.abs(150.0 + 60.0 * rng.standard_normal((batch, n_frames, 1))).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n_frames * hop, sample_rate=sr)
  out = synth(amps, hd, f0, return_outputs_dict=True)
  c = O.harmonic_get_controls(amps, hd, f0, sr)
  np.testing.assert_allclose(npy(out['controls']['harmonic_distribution']), c['harmonic_distribution'], rtol=2e-5, atol=1e-9)
  np.testing.assert_allclose(npy(out['controls']['amplitudes']), c['amplitudes'], rtol=2e-5)
  np.testing.assert_array_equal(npy(out['controls']['f0_hz']), f0)
  np.testing.assert_array_equal(npy(out['signal']), npy(synth(amps, hd, f0)))
  # with autograd on, one launch still yields audio and controls, and the gradient flows through the audio
  a_t = torch.tensor(amps, device=DEV, requires_grad=True)
  h_t = torch.tensor(hd, device=DEV, requires_grad=True)
  out_g = synth(a_t, h_t, f0, return_outputs_dict=True)
  np.testing.assert_array_equal(npy(out_g['signal']), npy(out['signal']))
  np.testing.assert_array_equal(npy(out_g['controls']['harmonic_distribution']), npy(out['controls']['harmonic_distribution']))
  assert not out_g['controls']['amplitudes'].requires_grad
  out_g['signal'].square().sum().backward()
  assert a_t.grad is not None and h_t.grad is not None and float(h_t.grad.abs().max()) > 0


def test_harmonic_table_kernel_edge_frequencies(ddsp):
  """f0 = 0, f0 above Nyquist, a harmonic exactly on Nyquist, a sweep through sr/2 inside frames."""
  n_frames, hop, k, sr = 40, 64, 100, 16000
  rng = np.random.default_rng(7)
  amps = rng.standard_normal((4, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((4, n_frames, k)).astype(np.float32)
  f0 = np.zeros((4, n_frames, 1), np.float32)
  f0[1] = 9000.0
  f0[2] = 80.0                                                      # harmonic 100 == 8000 Hz: masked (>=)
  f0[3, :, 0] = np.linspace(60.0, 4100.0, n_frames)                 # every harmonic crosses somewhere
  synth = ddsp.synths.Harmonic(n_samples=n_frames * hop, sample_rate=sr)
  ours = npy(synth(amps, hd, f0))
  truth = O.harmonic(amps, hd, f0, n_frames * hop, sr, dtype=np.float64)
  assert np.abs(ours[0]).max() <= 1e-7 and np.all(ours[1] == 0.0)     # f0 = 0: the taps cancel to round-off
  assert np.abs(ours - truth).max() <= 2e-5 * 2.0


@pytest.mark.parametrize('angular', [False, True])
@pytest.mark.parametrize('method', ['window', 'linear'])
def test_tf_op_order_kernel_matches_faithful_oracle_full_length(ddsp, angular, method):
  """The validation kernel follows the reference's fp32 op order (sequential tf.cumsum /
  angular_cumsum), so it tracks the fp32-faithful oracle over a whole 4 s clip, where the default
  path has drifted by O(1) from exact arithmetic."""
  x = canonical_inputs(1, seed=21, f0_center=200.0)
  c = O.harmonic_get_controls(x['amplitudes'], x['harmonic_distribution'], x['f0_hz'])
  faithful = O.harmonic_get_signal(c['amplitudes'], c['harmonic_distribution'], c['f0_hz'],
                                   amp_resample_method=method, use_angular_cumsum=angular)
  ours = npy(ddsp.core.harmonic_synthesis(c['f0_hz'], c['amplitudes'],
                                          harmonic_distribution=c['harmonic_distribution'],
                                          amp_resample_method=method, use_angular_cumsum=angular,
                                          tf_op_order=True))
  err = np.abs(ours - faithful)
  print('tf-order kernel vs faithful fp32 oracle (angular=%s, %s): max %.2e' % (angular, method, err.max()))
  # sin of fp32 phases up to 2e5 rad: 1 ulp of the phase is 1.6e-2 rad, so one differently rounded
  # add anywhere in the chain shows up as ~1e-2 on a harmonic; typical error is far smaller
  assert err.max() <= 5e-2 and np.sqrt((err**2).mean()) <= 2e-3
  if not angular:
    truth = O.harmonic_get_signal(c['amplitudes'], c['harmonic_distribution'], c['f0_hz'],
                                  amp_resample_method=method, dtype=np.float64)
    assert np.abs(faithful - truth).max() > 10 * err.max()    # the drift this kernel reproduces


@pytest.mark.parametrize('fs,n_frames,ragged', [(128, 40, 0), (192, 50, 5), (320, 30, 0), (80, 70, 3), (960, 9, 0)])
def test_filtered_noise_fused_other_frame_sizes(ddsp, noise_kernel, fs, n_frames, ragged):
  """M=65 with frame sizes other than 64 (48 kHz / VST configs: hop 192, 320, 960) on the fused kernel."""
  n = fs * n_frames - ragged
  rng = np.random.default_rng(fs)
  mags = rng.standard_normal((2, n_frames, 65)).astype(np.float32)
  noise = rng.uniform(-1, 1, (2, n)).astype(np.float32)
  out = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)(mags, noise=noise, return_outputs_dict=True)
  ref = O.filtered_noise(mags, noise, 0, dtype=np.float64)
  assert np.abs(npy(out['signal']) - ref).max() <= noise_tol(ref)
  np.testing.assert_allclose(npy(out['controls']['magnitudes']),
                             O.filtered_noise_get_controls(mags)['magnitudes'], rtol=2e-5, atol=1e-9)


# ---- effects.Reverb / long single-frame fft_convolve -------------
# REVERB  |ours - fp64 direct convolution| <= 1e-5 * max|ref| + 1e-6  (fp32 FFTs of 8192 points)
REVERB_CASES = ['reverb_b2_dry', 'reverb_b2_wet_rank3', 'reverb_trainable']


def reverb_tol(ref):
  return 1e-6 + 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize('name', REVERB_CASES)
def test_reverb_golden(ddsp, name):
  g = load_golden(name)
  rev = ddsp.effects.Reverb(trainable=bool(g['trainable']), reverb_length=g['ir'].shape[-1] if g['ir'].ndim == 1 else 48000,
                            add_dry=bool(g['add_dry']))
  if int(g['trainable']):
    rev._ir = ddsp.core.tf_float32(g['ir'])      # the variable's value the fixture was made with
    rev.built = True
    out = rev(g['audio'])
  else:
    out = rev(g['audio'], g['ir'])
  assert tuple(out.shape) == g['signal'].shape
  np.testing.assert_allclose(npy(out), g['signal'], rtol=0, atol=reverb_tol(g['signal']))


@pytest.mark.parametrize('batch,n,l,ir_batch,add_dry', [
    (2, 64000, 48000, 2, True),        # solo_instrument.gin: 4 s @ 16 kHz, 3 s reverb
    (3, 64000, 48000, 1, False),       # a trainable Reverb's single IR
    (1, 4096, 4096, 1, True),          # exactly one block / one partition
    (2, 4097, 4097, 2, False),         # one sample past the block / partition edge
    (2, 1000, 65536, 2, True),         # 16 partitions (the register window's limit), much longer than the audio
    (1, 20001, 5, 1, False),           # a tiny IR, ragged length (scalar load path)
    (2, 40000, 30000, 2, True)])       # 8 partitions, 10 spectra per row: the multiply-add pass's straight-line passes of eight
def test_reverb_vs_fp64_convolution(ddsp, batch, n, l, ir_batch, add_dry):
  import scipy.signal
  rng = np.random.default_rng(n + l)
  audio = rng.standard_normal((batch, n)).astype(np.float32)
  ir = (rng.standard_normal((ir_batch, l)) * np.exp(-np.arange(l) / (0.2 * l))).astype(np.float32)
  out = npy(ddsp.effects.Reverb(add_dry=add_dry)(audio, ir))
  h = ir.astype(np.float64).copy()
  h[:, 0] = 0.0                                                     # _mask_dry_ir
  ref = np.stack([scipy.signal.fftconvolve(audio[b].astype(np.float64), h[b % ir_batch])[:n]
                  for b in range(batch)])
  if add_dry:
    ref = ref + audio
  np.testing.assert_allclose(out, ref, rtol=0, atol=reverb_tol(ref))
  # and the fp32 restatement of the reference's own algorithm (one big FFT)
  np.testing.assert_allclose(out, O.reverb(audio, ir, add_dry=add_dry), rtol=0,
                             atol=4 * reverb_tol(ref))


@pytest.mark.parametrize('batch,n,l', [
    (2, 20000, 13000),        # 4 partitions of 4096 taps: a ring of four
    (5, 9000, 20000),         # 5 partitions (ring of eight), an odd batch: the last row pair holds one row
    (4, 30001, 48000),        # 12 partitions - the default reverb_length -, a length that is not a multiple of four
    (3, 5000, 65536),         # 16 partitions: the longest response of the windowed kernels (one row pair per block)
    (2, 64000, 12289),        # one tap past three partitions
    (3, 40000, 30000)])       # 8 partitions, 10 spectra: a straight-line pass of eight, then the tested form for two
def test_reverb_one_impulse_response_for_the_batch_row_pairs(ddsp, batch, n, l):
  """Round 5: one impulse response for the whole batch - the trainable Reverb of the shipped configurations
  (ddsp/effects.py:62-80, gin/models/solo_instrument.gin:26-40) - runs with two ROWS per complex transform
  (csrc/reverb.hip, "row pairs").  Against the fp64 convolution (ddsp/core.py:1382-1473 with one frame,
  delay_compensation = 0, the dry tap masked: effects.py:50-60,113-117); with and without the dry signal; odd batches (the
  last pair holds one row); the same bits from a second call and for a row pair run on its own; and the index-reversed forms
  the backward pass uses (core.fft_convolve_long)."""
  import scipy.signal
  rng = np.random.default_rng(n + l)
  audio = rng.standard_normal((batch, n)).astype(np.float32)
  ir = (rng.standard_normal((1, l)) * np.exp(-np.arange(l) / (0.2 * l))).astype(np.float32)
  h = ir[0].astype(np.float64).copy()
  h[0] = 0.0                                                        # _mask_dry_ir
  wet_ref = np.stack([scipy.signal.fftconvolve(audio[b].astype(np.float64), h)[:n] for b in range(batch)])
  for add_dry in (False, True):
    out = npy(ddsp.effects.Reverb(add_dry=add_dry)(audio, ir))
    ref = wet_ref + (audio if add_dry else 0.0)
    np.testing.assert_allclose(out, ref, rtol=0, atol=reverb_tol(ref))
  rev = ddsp.effects.Reverb(add_dry=False)
  out = npy(rev(audio, ir))
  np.testing.assert_array_equal(npy(rev(audio, ir)), out)
  np.testing.assert_array_equal(npy(rev(audio[:2], ir)), out[:2])   # (rows 0, 1: the first row pair alone)
  # the trainable form holds the response itself
  tr = ddsp.effects.Reverb(trainable=True, reverb_length=l, add_dry=False)
  tr.build(device=ddsp.core.tf_float32(audio).device)
  tr._ir = ddsp.core.tf_float32(ir[0])
  np.testing.assert_array_equal(npy(tr(audio)), out)
  # dL/d audio of the backward pass: reverse(conv(reverse(g), h))[0 : N] - and a delayed, shorter output window
  g = rng.standard_normal((batch, n)).astype(np.float32)
  got = npy(ddsp.core.fft_convolve_long(g, ir, delay=0, mask_tap0=True, reverse_audio=True, reverse_out=True))
  ref = np.stack([scipy.signal.fftconvolve(g[b, ::-1].astype(np.float64), h)[:n][::-1] for b in range(batch)])
  np.testing.assert_allclose(got, ref, rtol=0, atol=reverb_tol(ref))
  delay, n_out = 777, max(1, n // 3)
  got = npy(ddsp.core.fft_convolve_long(audio, ir, delay=delay, n_out=n_out))
  full = np.stack([scipy.signal.fftconvolve(audio[b].astype(np.float64), ir[0].astype(np.float64)) for b in range(batch)])
  np.testing.assert_allclose(got, full[:, delay:delay + n_out], rtol=0, atol=reverb_tol(full))


@pytest.mark.parametrize('batch,n,l,ir_batch,backward', [
    (2, 3000, 72000, 1, False),      # gin/models/vst/vst_48k.gin:88: a 72 000-tap reverb - 18 partitions, one IR for the batch (row pairs)
    (2, 5000, 70000, 2, False),      # ... an IR per row (two blocks of a row per spectrum: W_m for odd m)
    (3, 70000, 600, 1, True),        # dL/d ir of a clip of 70 000 samples: the AUDIO is the 18-partition operand of the correlation
])
def test_reverb_beyond_sixteen_partitions(ddsp, batch, n, l, ir_batch, backward):
  """Impulse responses of more than 65 536 taps (effects.FilteredNoiseReverb of vst_48k.gin has 72 000) and the backward pass of
  clips longer than that raised NotImplementedError until the end of round 5: the multiply-add pass kept its partitions in a
  register window of sixteen.  rv_mac_desc_kernel forms W_j in place from j = last downwards, any number of partitions."""
  import scipy.signal
  rng = np.random.default_rng(l)
  x = rng.standard_normal((batch, n)).astype(np.float32)
  h = (rng.standard_normal((ir_batch, l)) * np.exp(-np.arange(l) / (0.3 * l))).astype(np.float32)
  hm = h.astype(np.float64).copy(); hm[:, 0] = 0.0
  ref = np.stack([scipy.signal.fftconvolve(x[i].astype(np.float64), hm[i % ir_batch])[:n] for i in range(batch)]) + x
  rev = ddsp.effects.Reverb(add_dry=True)
  if not backward:
    np.testing.assert_allclose(npy(rev(x, h)), ref, rtol=0, atol=reverb_tol(ref))
    return
  g = rng.standard_normal((batch, n)).astype(np.float32)
  tx = ddsp.core.tf_float32(x).requires_grad_(True)
  th = ddsp.core.tf_float32(h).requires_grad_(True)
  out = rev(tx, th)
  out.backward(ddsp.core.tf_float32(g))
  np.testing.assert_allclose(npy(out), ref, rtol=0, atol=reverb_tol(ref))
  dx = np.stack([scipy.signal.fftconvolve(g[i].astype(np.float64)[::-1], hm[i % ir_batch])[:n][::-1] for i in range(batch)]) + g
  dh = np.stack([scipy.signal.fftconvolve(g[i].astype(np.float64), x[i].astype(np.float64)[::-1])[n - 1:n - 1 + l] for i in range(batch)])
  dh[:, 0] = 0.0
  if ir_batch == 1:
    dh = dh.sum(0, keepdims=True)
  np.testing.assert_allclose(npy(tx.grad), dx, rtol=0, atol=reverb_tol(dx))
  np.testing.assert_allclose(npy(th.grad).reshape(dh.shape), dh, rtol=0, atol=1e-6 + 1e-5 * max(np.abs(dh).max(), np.sqrt(n)))


def test_reverb_properties_full_size_batch32(ddsp):
  rng = np.random.default_rng(77)
  b, n, l = 32, 64000, 48000
  x1 = ddsp.core.tf_float32(rng.standard_normal((b, n)))
  x2 = ddsp.core.tf_float32(rng.standard_normal((b, n)))
  ir = ddsp.core.tf_float32(rng.standard_normal((b, l)) * np.exp(-np.arange(l) / 9000.0))
  wet = ddsp.effects.Reverb(add_dry=False)
  # linearity
  lhs = wet(0.5 * x1 - 2.0 * x2, ir)
  rhs = 0.5 * wet(x1, ir) - 2.0 * wet(x2, ir)
  scale = float(rhs.abs().max())
  assert float((lhs - rhs).abs().max()) <= 2e-5 * scale
  # a unit tap at k delays by k samples; tap 0 is masked; add_dry adds the input back
  delta = torch.zeros((1, l), device=DEV)
  delta[0, 0], delta[0, 4097] = 5.0, 1.0
  y = wet(x1, delta)
  assert float(y[:, :4097].abs().max()) <= 1e-5
  assert float((y[:, 4097:] - x1[:, :n - 4097]).abs().max()) <= 2e-5 * float(x1.abs().max())
  y_dry = ddsp.effects.Reverb(add_dry=True)(x1, delta)
  assert float((y_dry - y - x1).abs().max()) <= 1e-6
  # one IR tiled over the batch == the same IR given per row (since round 5 two sets of kernels: one response for the batch runs
  # reverb_fused.hip, a response per row reverb.hip's three launches - equal to rounding, not to the bit)
  one = wet(x1, ir[:1])
  rows = wet(x1, ir[:1].repeat(b, 1))
  assert float((one - rows).abs().max()) <= 2e-5 * float(rows.abs().max())


def test_fft_convolve_single_long_frame_routes_to_fft_path(ddsp):    # core.py:1428-1430, 1338-1379
  rng = np.random.default_rng(5)
  audio = rng.standard_normal((2, 3000)).astype(np.float32)
  ir = (rng.standard_normal((2, 2000)) * np.exp(-np.arange(2000) / 300.0)).astype(np.float32)
  for delay in (-1, 0, 17):
    out = npy(ddsp.core.fft_convolve(audio, ir, padding='same', delay_compensation=delay))
    ref = O.fft_convolve(audio, ir, padding='same', delay_compensation=delay, dtype=np.float64)
    np.testing.assert_allclose(out, ref, rtol=0, atol=reverb_tol(ref))


def test_reverb_in_processor_group(ddsp):                            # gin/models/solo_instrument.gin:26-40
  rng = np.random.default_rng(9)
  b, f, k, m, n, l = 2, 250, 60, 65, 16000, 12000
  feats = {'amps': rng.standard_normal((b, f, 1)).astype(np.float32),
           'harmonic_distribution': rng.standard_normal((b, f, k)).astype(np.float32),
           'f0_hz': (200 + rng.standard_normal((b, f, 1))).astype(np.float32),
           'magnitudes': rng.standard_normal((b, f, m)).astype(np.float32)}
  harm = ddsp.synths.Harmonic(n_samples=n)
  noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=3)
  add = ddsp.processors.Add()
  rev = ddsp.effects.Reverb(trainable=True, reverb_length=l)
  rev.build(device=torch.device(DEV))
  rev._ir = ddsp.core.tf_float32(rng.standard_normal(l) * np.exp(-np.arange(l) / 2000.0) * 0.05)
  dag = [(harm, ['amps', 'harmonic_distribution', 'f0_hz']), (noise, ['magnitudes']),
         (add, ['filtered_noise/signal', 'harmonic/signal']), (rev, ['add/signal'])]
  group = ddsp.processors.ProcessorGroup(dag=dag)
  outs = group.get_controls(feats)
  dry = npy(outs['add']['signal'])
  got = npy(group.get_signal(outs))
  ref = O.reverb(dry, npy(rev._ir), add_dry=True, dtype=np.float64)
  np.testing.assert_allclose(got, ref, rtol=0, atol=reverb_tol(ref))


# ---- losses.SpectralLoss forward ------------------------------------
# SPECTRAL  |ours - fp64 oracle| <= 2e-5 * |ref|   (sums of ~1e6 fp32 magnitudes, fp64 accumulation)
def test_spectral_loss_golden(ddsp):
  g = load_golden('spectral_loss')
  t, a = g['target_audio'], g['audio']
  for kwargs, key in [({}, 'loss_default'), (dict(mag_weight=1.0, logmag_weight=1.0), 'loss_ae_gin'),
                      (dict(fft_sizes=(512, 64), logmag_weight=0.5), 'loss_two_scales')]:
    out = ddsp.losses.SpectralLoss(**kwargs)(t, a)
    assert out.dim() == 0
    np.testing.assert_allclose(float(out), float(np.ravel(g[key])[0]), rtol=2e-5)


def test_spectral_loss_every_term_golden_and_gradient(ddsp):
  """The general form of SpectralLoss (delta / cumsum terms, 'L2', 'COSINE', the weights mask; losses.py:102-128,
  199-236) against the reference-source golden values, the fp64 oracle, and - the gradient - a directional finite
  difference of the fp64 oracle."""
  g = load_golden('spectral_loss_terms')
  t, a = g['target_audio'], g['audio']
  sizes = tuple(int(v) for v in g['fft_sizes'])
  kw = {k: float(g[k]) for k in ('mag_weight', 'delta_time_weight', 'delta_freq_weight', 'cumsum_freq_weight',
                                 'logmag_weight')}
  rng = np.random.default_rng(5)
  direction = rng.standard_normal(a.shape)
  direction /= np.abs(direction).max()
  for key, loss_type, w in (('l1', 'L1', None), ('l2', 'L2', None), ('cosine', 'COSINE', None),
                            ('l1_weighted', 'L1', g['weights']), ('l2_weighted', 'L2', g['weights']),
                            ('cosine_weighted', 'COSINE', g['weights'])):
    loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, loss_type=loss_type, **kw)
    out = loss(t, a, weights=w)
    assert out.dim() == 0
    np.testing.assert_allclose(float(out), float(np.ravel(g[key])[0]), rtol=5e-5, err_msg=key)
    ref64 = float(O.spectral_loss(t, a, sizes, loss_type, dtype=np.float64, weights=w, **kw))
    np.testing.assert_allclose(float(out), ref64, rtol=5e-5, err_msg=key)
    assert float(loss(t, a, weights=w)) == float(out)                    # deterministic
    # gradient: <dL/d audio, direction> against a central difference of the fp64 oracle
    ta = torch.tensor(a, device=DEV, requires_grad=True)
    loss(torch.tensor(t, device=DEV), ta, weights=w).backward()
    got = float((npy(ta.grad).astype(np.float64) * direction).sum())
    eps = 1e-5
    a64 = a.astype(np.float64)
    up = O.spectral_loss(t, a64 + eps * direction, sizes, loss_type, dtype=np.float64, weights=w, **kw)
    dn = O.spectral_loss(t, a64 - eps * direction, sizes, loss_type, dtype=np.float64, weights=w, **kw)
    want = float(up - dn) / (2 * eps)
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * abs(ref64), err_msg=key + ' gradient')
  # single terms at other sizes
  for kwargs, key in ((dict(fft_sizes=(256,), loss_type='L2', mag_weight=0.0, delta_time_weight=1.0), 'l2_delta_time_only'),
                      (dict(fft_sizes=(2048, 64), mag_weight=0.0, cumsum_freq_weight=1.0), 'l1_cumsum_only')):
    np.testing.assert_allclose(float(ddsp.losses.SpectralLoss(**kwargs)(t, a)), float(np.ravel(g[key])[0]), rtol=5e-5)
  # the same answer from the general path and the fused one where both apply ('L1' mag + logmag with unit weights)
  fused = float(ddsp.losses.SpectralLoss(fft_sizes=sizes, mag_weight=1.0, logmag_weight=0.75)(t, a))
  general = float(ddsp.losses.SpectralLoss(fft_sizes=sizes, mag_weight=1.0, logmag_weight=0.75)(t, a, weights=1.0))
  np.testing.assert_allclose(general, fused, rtol=1e-5)
  with pytest.raises(ValueError, match='broadcast'):
    ddsp.losses.SpectralLoss(fft_sizes=(64,))(t, a, weights=np.ones((2, 7, 1), np.float32))
  with pytest.raises(ValueError, match='Loss type'):
    ddsp.losses.SpectralLoss(loss_type='L3')(t, a)


@pytest.mark.parametrize('batch,n,sizes,kw', [
    (2, 1000, (48, 96), dict(mag_weight=1.0, logmag_weight=1.0)),
    (1, 9000, (1536, 3072, 192), dict(mag_weight=1.0, logmag_weight=0.5)),
    (2, 13000, (6144, 384, 256), dict(mag_weight=1.0, logmag_weight=1.0)),                 # the largest: one frame per block; mixed with 2^k
    # round 6: 8192-point transforms on the fused kernels (stft_l1_big_kernel: one signal of one frame at a time) - alone, mixed
    # with the one-grid sizes, on a clip shorter than the frame, and in the loss's general form (plain kernels)
    (2, 20000, (8192, 6144), dict(mag_weight=1.0, logmag_weight=1.0)),
    (1, 30000, (8192, 512, 6144, 96), dict(mag_weight=0.5, logmag_weight=1.0)),
    (3, 5001, (6144, 8192, 64), dict(mag_weight=1.0, logmag_weight=0.25)),
    (1, 20000, (8192,), dict(loss_type='L2', mag_weight=1.0, delta_time_weight=1.0, logmag_weight=1.0)),
    (2, 3000, (768, 192), dict(loss_type='L2', mag_weight=1.0, delta_time_weight=1.0, delta_freq_weight=1.0, cumsum_freq_weight=1.0)),
    (2, 3000, (384,), dict(loss_type='COSINE', mag_weight=1.0, logmag_weight=1.0)),
    # round 6 (VERDICT r5 "missing" #4): ANY even frame size - 100 and 1000 (round numbers), 102 (a hop of int(25.5) = 25 that does
    # not divide the frame), 34 and 8190 (the ends), mixed with powers of two
    (2, 3000, (100, 1000, 64), dict(mag_weight=1.0, logmag_weight=1.0)),
    (3, 2500, (102, 34), dict(mag_weight=1.0, logmag_weight=0.5)),
    (1, 20000, (8190, 5000, 2048), dict(mag_weight=1.0, logmag_weight=1.0)),
    (2, 4000, (250, 1022), dict(loss_type='L2', mag_weight=1.0, delta_time_weight=1.0, delta_freq_weight=1.0, cumsum_freq_weight=1.0, logmag_weight=1.0)),
])
def test_spectral_loss_with_frames_of_three_times_a_power_of_two(ddsp, batch, n, sizes, kw):
  """gin/models/vst/vst_48k.gin:56: fft_sizes = [6144, 3072, 1536, 768, 384, 192].  spectral_ops.stft (spectral_ops.py:34-47) hands
  tf.signal.stft fft_length=None: frames of F samples every F / 4 under a Hann window of F points, zero-padded to the enclosing
  power of two - 2 F / 3 + 1 bins.  stft_tq_mag_kernel / stft_tq_cot_bwd_kernel (the loss's general form); value against the
  fp64 oracle, the 'L1' gradient against its analytic one (distribution check: sign flips, see the 2^k test below)."""
  rng = np.random.default_rng(n)
  t = (0.3 * rng.standard_normal((batch, n))).astype(np.float32)
  a = (0.8 * t + 0.05 * rng.standard_normal((batch, n))).astype(np.float32)
  loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, **kw)
  ta = ddsp.core.tf_float32(a).requires_grad_(True)
  val = loss(t, ta)
  val.backward()
  ref = float(O.spectral_loss(t, a, sizes, dtype=np.float64, **kw))
  np.testing.assert_allclose(float(val.detach()), ref, rtol=5e-5)
  np.testing.assert_allclose(float(loss(t, a)), ref, rtol=5e-5)
  assert bool(torch.isfinite(ta.grad).all()) and float(ta.grad.abs().max()) > 0
  if kw.get('loss_type', 'L1') == 'L1':
    gref = O.spectral_loss_backward(t, a, sizes, kw['mag_weight'], kw['logmag_weight'])
    err = np.abs(npy(ta.grad) - gref)
    atol = 1e-9 + 2e-4 * np.abs(gref).max()
    assert np.median(err) <= 0.2 * atol and np.quantile(err, 0.9) <= atol, (float(np.median(err)), float(np.quantile(err, 0.9)), atol)
  for bad in (101, 20, 8192 + 2):                            # odd; below 34 without being a power of two; beyond an 8192-point transform
    with pytest.raises(ValueError, match='fft_sizes'):
      ddsp.losses.SpectralLoss(fft_sizes=(bad,))(t, a)


def test_spectral_loss_loudness_term_golden_and_gradient(ddsp):
  """losses.py:238-242: loudness_weight * mean_difference(compute_loudness(target), compute_loudness(audio)) with
  spectral_ops.compute_loudness (spectral_ops.py:253-324: centred frames of 2048 samples every 64, A-weighted mean power in dB
  per frame, 80 dB of range) - built at the end of round 5 (the reference's own SpectralLossTest sets loudness_weight = 1).
  Values against the golden vectors (the reference's source, librosa's two functions restated) and the fp64 oracle; the gradient
  against the oracle's analytic one, which tests/test_oracle.py checks against central differences."""
  g = load_golden('spectral_loss_loudness')
  t, a = g['target_audio'], g['audio']
  all_terms = dict(mag_weight=1.0, delta_time_weight=0.5, delta_freq_weight=0.25, cumsum_freq_weight=0.125, logmag_weight=0.75)
  for key, kwargs in (('l1_only', dict(mag_weight=0.0, loudness_weight=1.0)),
                      ('l2_only', dict(loss_type='L2', mag_weight=0.0, loudness_weight=1.0)),
                      ('cosine_only', dict(loss_type='COSINE', mag_weight=0.0, loudness_weight=1.0)),
                      ('l1_all', dict(loudness_weight=0.5, **all_terms))):
    got = float(ddsp.losses.SpectralLoss(**kwargs)(t, a))
    np.testing.assert_allclose(got, float(np.ravel(g[key])[0]), rtol=5e-5, err_msg=key)
    np.testing.assert_allclose(got, float(O.spectral_loss(t, a, dtype=np.float64, **kwargs)), rtol=5e-5, err_msg=key)
  np.testing.assert_allclose(O.compute_loudness(a, dtype=np.float32), g['loudness_audio'], rtol=0, atol=2e-4)
  # gradient of the loudness term alone ('L1'): d/d audio of mean |L_t - L_a|
  loss = ddsp.losses.SpectralLoss(mag_weight=0.0, loudness_weight=1.0)
  ta = ddsp.core.tf_float32(a).requires_grad_(True)
  (2.0 * loss(t, ta)).backward()
  lt, la = O.compute_loudness(t, dtype=np.float64), O.compute_loudness(a, dtype=np.float64)
  ref = 2.0 * O.compute_loudness_backward(a, -np.sign(lt - la) / lt.size)
  err = np.abs(npy(ta.grad) - ref)
  atol = 1e-9 + 2e-4 * np.abs(ref).max()
  assert np.quantile(err, 0.99) <= atol and err.max() <= 20 * atol, (float(np.quantile(err, 0.99)), float(err.max()), atol)
  # the reference's own test (losses_test.py: SpectralLossTest): every term on, two identical signals
  every = dict(mag_weight=1.0, delta_time_weight=1.0, delta_freq_weight=1.0, cumsum_freq_weight=1.0, logmag_weight=1.0, loudness_weight=1.0)
  ones = np.ones((3, 8000), np.float32)
  v = ddsp.losses.SpectralLoss(**every)(ones, ones)
  assert list(v.shape) == [] and float(v) == 0.0


def test_spectral_loss_degenerate_arguments_follow_the_reference(ddsp):
  """What tools/fuzz_api_vs_reference.py found the mirror doing differently from the reference's own code:
  a loss type that does not exist raises when losses.mean_difference is CALLED (losses.py:102-128) - a loss whose every weight
  is zero never calls it and returns 0; a delta-time term over ONE frame is tf.reduce_mean of no elements, NaN ('L1' / 'L2'),
  or cosine_distance's safe mean, 0 ('COSINE')."""
  rng = np.random.default_rng(5)
  t = (0.3 * rng.standard_normal((2, 64))).astype(np.float32)
  a = (0.8 * t + 0.05 * rng.standard_normal((2, 64))).astype(np.float32)
  assert float(ddsp.losses.SpectralLoss(loss_type='L3', mag_weight=0.0)(t, a)) == 0.0
  with pytest.raises(ValueError, match='Loss type'):
    ddsp.losses.SpectralLoss(loss_type='L3', mag_weight=0.0, delta_freq_weight=1.0)(t, a)
  one_frame = dict(fft_sizes=(256, 64), mag_weight=1.0, delta_time_weight=1.0)          # 64 samples: one frame of size 256
  assert np.isnan(float(ddsp.losses.SpectralLoss(loss_type='L1', **one_frame)(t, a)))
  assert np.isnan(float(ddsp.losses.SpectralLoss(loss_type='L2', **one_frame)(t, a)))
  cos = float(ddsp.losses.SpectralLoss(loss_type='COSINE', **one_frame)(t, a))
  ref = float(O.spectral_loss(t, a, (64,), loss_type='COSINE', mag_weight=1.0, delta_time_weight=1.0, dtype=np.float64)) + \
      float(O.spectral_loss(t, a, (256,), loss_type='COSINE', mag_weight=1.0, dtype=np.float64))
  np.testing.assert_allclose(cos, ref, rtol=5e-5)


# (9, 5000), (17, 1030): 45 and 34 units of (row, 1024 samples) - not multiples of the eight XCDs the block order deals them to (sl_where)
@pytest.mark.parametrize('batch,n', [(2, 64000), (3, 12345), (1, 100), (9, 5000), (17, 1030)])
def test_spectral_loss_vs_fp64_oracle(ddsp, batch, n):
  rng = np.random.default_rng(n)
  t = (0.3 * rng.standard_normal((batch, n))).astype(np.float32)
  a = (t * rng.uniform(0.5, 1.5) + 0.02 * rng.standard_normal((batch, n))).astype(np.float32)
  a[0, n // 2:] = 0.0                                            # a silent stretch: exact zeros -> safe_log eps
  loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
  ref = float(O.spectral_loss(t, a, logmag_weight=1.0, dtype=np.float64))
  np.testing.assert_allclose(float(loss(t, a)), ref, rtol=2e-5)
  np.testing.assert_allclose(float(loss(t[..., None], a[..., None])), ref, rtol=2e-5)   # [B, N, 1] audio
  assert float(loss(t, t)) == 0.0
  d = loss.get_losses_dict(t, a)
  assert list(d) == ['spectral_loss']
  # run-to-run determinism (fixed-order fp64 reduction)
  assert float(loss(t, a)) == float(loss(t, a))


def test_spectral_loss_on_the_synth_output_batch32(ddsp):          # ae.gin:36-41 on config-3 shapes
  rng = np.random.default_rng(3)
  b, f, k, n = 32, 1000, 100, 64000
  harm = ddsp.synths.Harmonic(n_samples=n)
  audio = harm(rng.standard_normal((b, f, 1)), rng.standard_normal((b, f, k)), 200 + rng.standard_normal((b, f, 1)))
  target = harm(rng.standard_normal((b, f, 1)), rng.standard_normal((b, f, k)), 210 + rng.standard_normal((b, f, 1)))
  loss = ddsp.losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)
  got = float(loss(target, audio))
  # the sum over sizes is additive: each scale on its own adds up to the total
  parts = sum(float(ddsp.losses.SpectralLoss(fft_sizes=(s,), logmag_weight=1.0)(target, audio))
              for s in (2048, 1024, 512, 256, 128, 64))
  np.testing.assert_allclose(got, parts, rtol=1e-6)
  # symmetric in its arguments (L1), and matches the oracle on the first two clips' worth of data
  np.testing.assert_allclose(float(loss(audio, target)), got, rtol=1e-6)
  ref2 = float(O.spectral_loss(npy(target[:2]), npy(audio[:2]), logmag_weight=1.0, dtype=np.float64))
  np.testing.assert_allclose(float(loss(target[:2].contiguous(), audio[:2].contiguous())), ref2, rtol=2e-5)


# ---- core.streaming_harmonic_synthesis --------------------------------
STREAMING_CASES = ['streaming_2frames_linear', 'streaming_2frames_window', 'streaming_nyquist_crossing',
                   'streaming_no_distribution', 'streaming_cubic', 'streaming_nearest_ragged', 'streaming_linear_ragged']


@pytest.mark.parametrize('name', STREAMING_CASES)
def test_streaming_synthesis_golden(ddsp, name):
  g = load_golden(name)
  hd = g.get('harmonic_distribution')
  audio, final_phase = ddsp.core.streaming_harmonic_synthesis(
      g['f0_hz'], g['amplitudes'], hd, g['initial_phase'], int(g['n_samples']), int(g['sample_rate']),
      str(g['amp_method']))
  assert tuple(audio.shape) == g['audio'].shape and tuple(final_phase.shape) == g['final_phase'].shape
  a64, p64 = O.streaming_harmonic_synthesis(
      g['f0_hz'], g['amplitudes'], hd, g['initial_phase'], int(g['n_samples']), int(g['sample_rate']),
      str(g['amp_method']), dtype=np.float64)
  amp_sum = float(np.abs(g['amplitudes']).max()) * (1.0 if hd is not None else 1.0)
  harm_truth_check(npy(audio), a64, max(1.0, amp_sum))
  np.testing.assert_allclose(npy(audio), g['audio'], rtol=0, atol=HARM_FAITHFUL_ATOL)
  # the carried phase: equal modulo 2 pi to fp64 truth (the reference wraps before adding initial_phase)
  d = (npy(final_phase) - p64 + np.pi) % (2 * np.pi) - np.pi
  assert np.abs(d).max() < 2e-5


def test_harmonic_oscillator_bank_audio_rate_golden(ddsp):         # core.py:966-1025
  g = load_golden('harmonic_oscillator_bank')
  sr = int(g['sample_rate'])
  for mode, angular, phase0 in (('angular', True, g['initial_phase']), ('cumsum', False, None)):
    audio, final_phase = ddsp.core.harmonic_oscillator_bank(g['frequency'], g['amplitude_envelopes'], phase0, sr,
                                                            use_angular_cumsum=angular)
    assert tuple(audio.shape) == g['audio_' + mode].shape and tuple(final_phase.shape) == (2, 1, 1)
    a64, p64 = O.harmonic_oscillator_bank(g['frequency'].astype(np.float64), g['amplitude_envelopes'].astype(np.float64),
                                          None if phase0 is None else phase0.astype(np.float64), sr,
                                          use_angular_cumsum=angular)
    harm_truth_check(npy(audio), a64, float(g['amplitude_envelopes'].sum(-1).max()))
    # against the reference's own fp32 arithmetic: within the drift of its sequential / chunked phase sums
    np.testing.assert_allclose(npy(audio), g['audio_' + mode], rtol=0, atol=HARM_FAITHFUL_ATOL * 12)
    if angular:                                  # equal modulo 2 pi (the reference wraps chunk by chunk)
      d = (npy(final_phase) - p64 + np.pi) % (2 * np.pi) - np.pi
      assert np.abs(d).max() < 2e-5
    else:                                        # the plain sum: hundreds of radians, fp32 resolution
      np.testing.assert_allclose(npy(final_phase), p64, rtol=2e-7, atol=0)
  with pytest.raises(ValueError, match='frequency'):
    ddsp.core.harmonic_oscillator_bank(g['frequency'][:, :10], g['amplitude_envelopes'])


def test_streaming_chunks_are_phase_continuous(ddsp):               # inference.py:446-472 call pattern
  rng = np.random.default_rng(8)
  hop, k, sr, n_hops = 64, 60, 16000, 40
  f0 = (220.0 + 30.0 * np.sin(np.arange(n_hops + 1) / 5.0)).astype(np.float32)
  amps = rng.uniform(0.5, 1.0, n_hops + 1).astype(np.float32)
  hd = rng.uniform(0.0, 1.0, (n_hops + 1, k)).astype(np.float32)
  phase = np.zeros((1, 1, 1), np.float32)
  phase64 = np.zeros((1, 1, 1))
  chunks, ref_chunks = [], []
  for i in range(n_hops):
    args = (f0[None, i:i + 2, None], amps[None, i:i + 2, None], hd[None, i:i + 2, :])
    a, phase = ddsp.core.streaming_harmonic_synthesis(*args, initial_phase=phase, n_samples=hop,
                                                      sample_rate=sr)
    r, phase64 = O.streaming_harmonic_synthesis(*args, initial_phase=phase64, n_samples=hop,
                                                sample_rate=sr, dtype=np.float64)
    chunks.append(npy(a)[0]); ref_chunks.append(r[0])
  got, ref = np.concatenate(chunks), np.concatenate(ref_chunks)
  # 40 calls with the phase carried through fp32: the drift stays far below audibility
  assert np.abs(got - ref).max() < 2e-3
  assert np.abs(got).max() > 0.2


@pytest.mark.parametrize('method', ['linear', 'cubic'])
def test_streaming_frame_rate_nyquist_mask_takes_the_reference_fp32_side(ddsp, method):
  """tools/fuzz_parity.py streaming:53047898 (round 6): f0 = fl32(296.2963) Hz, harmonic 27 at 7999.99997 Hz in exact arithmetic
  and at fl32(27 f0) = 8000.0 in the reference's fp32 product (core.get_harmonic_frequencies, then remove_above_nyquist's
  f >= sr / 2: core.py:869-903): the reference drops it from the frame's distribution, and so must streaming_harmonic_synthesis -
  the fp64 oracle keeps it and is 1e-2 away, the fp32-faithful oracle (the reference's op order) agrees to 3e-4."""
  f32 = np.float32
  f0 = np.array([296.2963, 305.857], dtype=f32).reshape(1, 2, 1)
  assert f32(27) * f0[0, 0, 0] == f32(8000.0) and 27 * float(f0[0, 0, 0]) < 8000.0          # the premise
  rng = np.random.default_rng(53047898)
  amps = rng.uniform(0.1, 1.0, (1, 2, 1)).astype(f32)
  hd = rng.uniform(0.0, 1.0, (1, 2, 100)).astype(f32)
  got, _ = ddsp.core.streaming_harmonic_synthesis(f0, amps, hd, n_samples=64, sample_rate=16000, amp_resample_method=method)
  ref32, _ = O.streaming_harmonic_synthesis(f0, amps, hd, n_samples=64, sample_rate=16000, amp_resample_method=method,
                                            dtype=np.float32)
  ref64, _ = O.streaming_harmonic_synthesis(f0, amps, hd, n_samples=64, sample_rate=16000, amp_resample_method=method,
                                            dtype=np.float64)
  assert np.abs(npy(got) - ref32).max() <= 3e-4
  assert np.abs(ref64 - ref32).max() >= 6e-4             # ... which is not what exact arithmetic gives: the case is what it claims


@pytest.mark.parametrize('method,n', [('linear', 320), ('window', 320), ('nearest', 333), ('cubic', 250)])
def test_streaming_synthesis_with_per_harmonic_amplitudes_and_no_distribution(ddsp, method, n):
  """core.streaming_harmonic_synthesis(frequencies, amplitudes [batch, n_frames, n_harmonics]) without a distribution: the
  reference takes the amplitudes as the harmonic amplitudes themselves (`harmonic_amplitudes = amplitudes`, core.py:1150-1151 -
  its docstring says [.., 1], its code takes any last axis).  This entry raised until tools/fuzz_parity.py drew the case."""
  rng = np.random.default_rng(n)
  b, f, k, sr = 2, 5, 20, 16000
  f0 = rng.uniform(80.0, 500.0, (b, f, 1)).astype(np.float32)
  amps = rng.uniform(0.0, 1.0, (b, f, k)).astype(np.float32)
  phase = rng.uniform(0.0, 6.0, (b, 1, 1)).astype(np.float32)
  got, final = ddsp.core.streaming_harmonic_synthesis(f0, amps, None, initial_phase=phase, n_samples=n, sample_rate=sr,
                                                      amp_resample_method=method)
  ref, ref_final = O.streaming_harmonic_synthesis(f0, amps, None, initial_phase=phase, n_samples=n, sample_rate=sr,
                                                  amp_resample_method=method, dtype=np.float64)
  assert tuple(got.shape) == (b, n) and tuple(final.shape) == (b, 1, 1)
  np.testing.assert_allclose(npy(got), ref, rtol=0, atol=3e-4 * max(1.0, np.abs(ref).max()))
  d = np.abs(((npy(final).astype(np.float64) - ref_final + np.pi) % (2 * np.pi)) - np.pi)
  assert d.max() <= 2e-4
  with pytest.raises(ValueError, match='frequencies must have shape'):
    ddsp.core.streaming_harmonic_synthesis(f0[:, :3], amps, None, n_samples=n, sample_rate=sr, amp_resample_method=method)


# ---- backward pass of Harmonic ----------------------------------------
# GRAD  |ours - fp64 analytic gradient| <= 2e-4 * max|ref| + 1e-6   (fp32 sines, fp32 accumulation
#       over one frame; the analytic oracle is itself checked against finite differences on CPU)
def grad_tol(ref):
  return 1e-6 + 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize('batch,n_frames,k,hop,sr,f_lo,f_hi,method,scale,normalize', [
    (2, 25, 12, 64, 16000, 300.0, 900.0, 'window', True, True),       # harmonics crossing Nyquist mid-frame
    (2, 20, 100, 64, 16000, 69.0, 71.0, 'window', True, True),        # all 100 harmonics live
    (1, 10, 40, 192, 48000, 200.0, 700.0, 'linear', True, True),      # hop 192, linear envelopes
    (2, 16, 20, 64, 16000, 350.0, 600.0, 'window', False, False),     # scale_fn=None, no Nyquist normalisation
    (1, 9, 200, 100, 16000, 30.0, 45.0, 'linear', True, True),        # K=200 (two wavefronts), hop 100
    (2, 12, 30, 50, 16000, 150.0, 400.0, 'window', True, True)])      # K not a multiple of 4, hop 50 (generic forward path)
def test_harmonic_backward_vs_analytic_oracle(ddsp, batch, n_frames, k, hop, sr, f_lo, f_hi, method, scale,
                                              normalize):
  rng = np.random.default_rng(batch * 1000 + k)
  n = n_frames * hop
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  if not scale:
    amps, hd = np.abs(amps) + 0.1, np.abs(hd) + 0.05
  f0 = rng.uniform(f_lo, f_hi, (batch, n_frames, 1)).astype(np.float32)
  g = rng.standard_normal((batch, n)).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, scale_fn=ddsp.core.exp_sigmoid if scale else None,
                               normalize_below_nyquist=normalize, amp_resample_method=method)
  ta = ddsp.core.tf_float32(amps).requires_grad_(True)
  th = ddsp.core.tf_float32(hd).requires_grad_(True)
  audio = synth(ta, th, f0)
  assert audio.requires_grad
  (audio * ddsp.core.tf_float32(g)).sum().backward()
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid if scale else None, normalize, method)
  np.testing.assert_allclose(npy(ta.grad), ga, rtol=0, atol=grad_tol(ga))
  np.testing.assert_allclose(npy(th.grad), gh, rtol=0, atol=grad_tol(gh))
  # the forward value is the same with and without recording
  np.testing.assert_array_equal(npy(audio), npy(synth(amps, hd, f0)))


@pytest.mark.parametrize('batch,n_frames,k,hop,sr,f0s', [
    (3, 40, 100, 64, 16000, (69.0, 71.0)),          # the canonical regime: all harmonics live, under one revolution per frame
    (2, 40, 100, 64, 16000, (180.0, 420.0)),        # harmonics crossing Nyquist in most frames, up to two revolutions per tile
    (2, 30, 128, 64, 16000, (900.0, 2500.0)),       # ten and more revolutions per tile, few live harmonics (eight taps)
    (2, 24, 60, 192, 48000, (100.0, 130.0)),        # three tiles per frame
    (2, 30, 100, 100, 16000, (20.0, 40.0)),         # frames of 100 samples; f0 below sr / 512 in most frames: the plain sum inside the kernel
    (1, 20, 37, 50, 16000, (0.0, 300.0)),           # f0 = 0 and a jump across many harmonics: more than eight crossing harmonics
    (2, 20, 200, 192, 48000, (100.0, 130.0)),       # 129 .. 200 harmonics (BASELINE configs[4]'s shape): ten taps, sixteen wavefronts
    (2, 30, 160, 64, 16000, (45.0, 60.0)),          # ... on frames of 64
])
def test_harmonic_backward_on_the_wavetable_adjoint(ddsp, batch, n_frames, k, hop, sr, f0s):
  """harm_bwd_table_kernel (the adjoint of the wavetable synthesis: spreading + one matrix product) against the fp64 analytic
  gradient, in every regime that takes a different way through it; the same call twice gives the same bits."""
  rng = np.random.default_rng(k * 31 + hop)
  n = n_frames * hop
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  f0 = rng.uniform(f0s[0], f0s[1], (batch, n_frames, 1)).astype(np.float32)
  if f0s[0] == 0.0:
    f0[:, ::5] = 0.0
    f0[:, 3::7] = 3000.0
  g = rng.standard_normal((batch, n)).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  grads = []
  for _ in range(2):
    ta = ddsp.core.tf_float32(amps).requires_grad_(True)
    th = ddsp.core.tf_float32(hd).requires_grad_(True)
    synth(ta, th, f0).backward(ddsp.core.tf_float32(g))
    grads.append((npy(ta.grad), npy(th.grad)))
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid, True, 'window')
  np.testing.assert_allclose(grads[0][0], ga, rtol=0, atol=grad_tol(ga))
  np.testing.assert_allclose(grads[0][1], gh, rtol=0, atol=grad_tol(gh))
  np.testing.assert_array_equal(grads[0][0], grads[1][0])
  np.testing.assert_array_equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize('k,hop,sr', [(300, 64, 48000), (257, 100, 16000), (512, 64, 48000), (513, 64, 48000), (1000, 64, 48000), (2048, 50, 48000)])
def test_harmonic_more_than_256_harmonics_forward_and_backward(ddsp, k, hop, sr):
  """257 .. 512 harmonics (300 live ones need 48 kHz and an f0 below 80 Hz): the forward runs the plain closed-form kernels, the
  backward - whose closed-form kernels stop at 256 - the chain of materialised envelopes and its adjoint (it raised
  DDSP_ERR_UNSUPPORTED until a probe at the end of round 5).  Round 6 (VERDICT r5 "missing" #3: the reference has no cap): up to 2048
  harmonics - the controls kernels hold a row as up to 32 values per lane; beyond that is refused by name."""
  rng = np.random.default_rng(k)
  b, f = 1, 5
  n = f * hop
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = rng.uniform(30.0, 45.0, (b, f, 1)).astype(np.float32) if k <= 512 else rng.uniform(9.0, 20.0, (b, f, 1)).astype(np.float32)
  g = rng.standard_normal((b, n)).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  truth = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=sr, dtype=np.float64)
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  assert np.abs(npy(synth(amps, hd, f0)) - truth).max() <= HARM_TRUTH_ATOL * scale
  ta = ddsp.core.tf_float32(amps).requires_grad_(True)
  th = ddsp.core.tf_float32(hd).requires_grad_(True)
  out = synth(ta, th, f0)
  out.backward(ddsp.core.tf_float32(g))
  assert np.abs(npy(out.detach()) - truth).max() <= 4 * HARM_TRUTH_ATOL * scale
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid, True, 'window')
  slack = 1.0 if hop == 64 else 3.0
  np.testing.assert_allclose(npy(ta.grad), ga, rtol=0, atol=slack * grad_tol(ga))
  np.testing.assert_allclose(npy(th.grad), gh, rtol=0, atol=slack * grad_tol(gh))
  with pytest.raises(NotImplementedError, match='2048 harmonics'):
    ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)(amps, np.zeros((b, f, 2049), np.float32), f0)


@pytest.mark.parametrize('hop', [20, 40, 100, 200])
def test_harmonic_backward_ragged_frames_and_steep_f0_drops(ddsp, hop):
  """frame sizes that are not multiples of 64 with f0 falling steeply inside a frame.  The lanes past a
  frame's last sample carried the frequency ramp EXTRAPOLATED past the frame - backwards when f0 falls -, harm_bwd_table_kernel
  took the tile's revolution count from lane 63 and skipped live lanes' turns: at hop 20 more than half of the gradient values
  were wrong by O(1).  The advisor's reproduction: f0 alternating 200-390 Hz and 35-45 Hz."""
  rng = np.random.default_rng(hop)
  batch, n_frames, k, sr = 1, 96, 20, 16000
  n = n_frames * hop
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  f0 = rng.uniform(200.0, 390.0, (batch, n_frames, 1)).astype(np.float32)
  f0[:, 1::2] = rng.uniform(35.0, 45.0, (batch, n_frames // 2, 1)).astype(np.float32)
  f0[:, 40:48] = rng.uniform(200.0, 390.0, (batch, 8, 1)).astype(np.float32)     # and a stretch without drops
  g = rng.standard_normal((batch, n)).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  ta = ddsp.core.tf_float32(amps).requires_grad_(True)
  th = ddsp.core.tf_float32(hd).requires_grad_(True)
  synth(ta, th, f0).backward(ddsp.core.tf_float32(g))
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid, True, 'window')
  # (frames of 200 samples with f0 jumping by 300 Hz per frame: the oracle keeps TF's fp32 resize positions, which are up to
  # 1.5e-5 off r / hop for frame sizes that are not powers of two - DESIGN.md "known limits"; the plain-sum kernel is the same
  # 6e-4 away from it at the same element.  The bug this test is for was 0.2 - 2.3.)
  slack = 3.0 if hop == 200 else 1.0
  np.testing.assert_allclose(npy(ta.grad), ga, rtol=0, atol=slack * grad_tol(ga))
  np.testing.assert_allclose(npy(th.grad), gh, rtol=0, atol=slack * grad_tol(gh))


def test_harmonic_sloping_ramp_exactly_on_nyquist(ddsp):
  """tools/fuzz_parity.py harmonic_bwd:41016811 (round 6, the one failure of seeds 31 / 37 / 41 / 43): frames of 192 samples,
  harmonic 8 of f0 falling from fl32(1000.07117) to fl32(999.85767) Hz - 8 f0 falls by 1.7080078125 Hz, a third of which is
  representable, so at r = 64 the ramp is EXACTLY 8000 Hz in fp64 and in the kernels' (and the reference's) fp32 op order: masked.
  The oracle's backward keeps TF's fp32 resize POSITION (fl32(4096 fl32(1 / 192)) = 21.333334) in fp64 arithmetic and finds the
  harmonic 1e-6 Hz below Nyquist: one sample's worth of that harmonic, 1 % of dL/d amplitude of the frame.  The exact-arithmetic
  checker reports such samples (`hits`: exactly on Nyquist on a sloping ramp) beside the knife edges: the forward output agrees
  with exact arithmetic there (both mask the harmonic); comparisons against oracle.harmonic_backward carry no cotangent on them."""
  f32 = np.float32
  hop, sr, k = 192, 16000, 20
  f0 = np.array([1000.3, 1000.07117, 999.85767, 1000.0506, 999.9], dtype=f32).reshape(1, -1, 1)
  n_frames = f0.shape[1]
  n = n_frames * hop
  rng = np.random.default_rng(41016811)
  amps = rng.standard_normal((1, n_frames, 1)).astype(f32)
  hd = rng.standard_normal((1, n_frames, k)).astype(f32)
  exact, knife, exact32 = _harmonic_exact(amps, hd, f0, n, sr, 'window', with_knife_edges='fp32 mask')
  _, _, hits = _harmonic_exact(amps, hd, f0, n, sr, 'window', with_knife_edges='and exact hits')
  t = 1 * hop + 64
  top, bot = f32(f0[0, 1, 0] * f32(8)), f32(f0[0, 2, 0] * f32(8))
  assert float(top) + (float(bot) - float(top)) * (64 / 192) == 8000.0          # the premise: exactly on Nyquist in fp64
  assert hits[0, t] and not knife[0, t] and int(hits.sum()) == 1 and int(knife.sum()) <= 8
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  got = npy(synth(amps, hd, f0))
  assert np.abs(got - exact)[~knife].max() <= HARM_TABLE_ATOL
  assert_knife_edges_take_the_fp32_side(got, exact32, knife, HARM_TABLE_ATOL, 'sloping ramp on Nyquist')
  g = rng.standard_normal((1, n)).astype(f32)
  g[knife | hits] = 0.0
  ta = ddsp.core.tf_float32(amps).requires_grad_(True)
  th = ddsp.core.tf_float32(hd).requires_grad_(True)
  synth(ta, th, f0).backward(ddsp.core.tf_float32(g))
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid, True, 'window')
  np.testing.assert_allclose(npy(ta.grad), ga, rtol=0, atol=grad_tol(ga))
  np.testing.assert_allclose(npy(th.grad), gh, rtol=0, atol=grad_tol(gh))


@pytest.mark.parametrize('scale', [1e-10, 1e-6, 1.0, 3e4, 1e5])
def test_synth_backward_is_invariant_to_the_scale_of_grad_audio(ddsp, scale):
  """the matrix-core backward kernels split dL/d audio (Harmonic: its spread image G) into fp16 hi / lo
  pairs; without normalisation 3e4 overflowed to NaN, 1e-10 lost 7-9 %, smaller gradients flushed to zero.  tf.GradientTape's
  fp32 gradients are scale invariant (loss scaling by 2^16, sum-reduced losses): so are these now - grad(s g) / s equals
  grad(g) to fp32 rounding for every s, checked against the fp64 analytic gradient."""
  rng = np.random.default_rng(77)
  batch, n_frames, k, hop = 2, 32, 100, 64
  n = n_frames * hop
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, k)).astype(np.float32)
  f0 = rng.uniform(69.0, 71.0, (batch, n_frames, 1)).astype(np.float32)
  mags = (rng.standard_normal((batch, n_frames, 65)) + 4.0).astype(np.float32)
  g = rng.standard_normal((batch, n)).astype(np.float32)
  gs = (g.astype(np.float64) * scale).astype(np.float32)
  harm = ddsp.synths.Harmonic(n_samples=n)
  ta = ddsp.core.tf_float32(amps).requires_grad_(True)
  th = ddsp.core.tf_float32(hd).requires_grad_(True)
  harm(ta, th, f0).backward(ddsp.core.tf_float32(gs))
  ga, gh = O.harmonic_backward(amps, hd, f0, gs, n, 16000, O.exp_sigmoid, True, 'window')
  np.testing.assert_allclose(npy(ta.grad) / scale, ga / scale, rtol=0, atol=grad_tol(ga / scale))
  np.testing.assert_allclose(npy(th.grad) / scale, gh / scale, rtol=0, atol=grad_tol(gh / scale))
  for given_noise in (False, True):
    noise_synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=5)
    # supplied noise at an arbitrary scale of its own (the forward is linear in it)
    noise = (rng.uniform(-1, 1, (batch, n)) * 300.0).astype(np.float32) if given_noise else O.device_uniform_noise(batch, n, seed=5)
    tm = ddsp.core.tf_float32(mags).requires_grad_(True)
    noise_synth(tm, noise=noise if given_noise else None).backward(ddsp.core.tf_float32(gs))
    ref = O.filtered_noise_backward(mags, noise, gs, 0, O.exp_sigmoid)
    np.testing.assert_allclose(npy(tm.grad) / scale, ref / scale, rtol=0, atol=(1e-6 + 2e-5 * np.abs(ref).max()) / scale)


def test_harmonic_backward_full_size_properties(ddsp):
  rng = np.random.default_rng(12)
  b, f, k, n = 32, 1000, 100, 64000
  amps = ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True)
  hd = ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True)
  f0 = ddsp.core.tf_float32(200 + rng.standard_normal((b, f, 1)))
  synth = ddsp.synths.Harmonic(n_samples=n)
  g1 = ddsp.core.tf_float32(rng.standard_normal((b, n)))
  g2 = ddsp.core.tf_float32(rng.standard_normal((b, n)))
  def grads(g):
    amps.grad = hd.grad = None
    (synth(amps, hd, f0) * g).sum().backward()
    return amps.grad.clone(), hd.grad.clone()
  a1, h1 = grads(g1)
  a2, h2 = grads(g2)
  a3, h3 = grads(0.5 * g1 - 2.0 * g2)                                # the backward map is linear in grad_audio
  assert float((a3 - (0.5 * a1 - 2.0 * a2)).abs().max()) <= 1e-4 * float(a1.abs().max() + a2.abs().max())
  assert float((h3 - (0.5 * h1 - 2.0 * h2)).abs().max()) <= 1e-4 * float(h1.abs().max() + h2.abs().max())
  # harmonics at or above Nyquist (k * 200 Hz >= 8 kHz: k >= 40) get exactly zero gradient
  assert float(h1[:, :, 45:].abs().max()) == 0.0
  # directional derivative: <grad, d> ~ (L(x + eps d) - L(x - eps d)) / 2 eps, L = <audio, g1>
  d = ddsp.core.tf_float32(rng.standard_normal((b, f, 1)))
  with torch.no_grad():
    eps = 1e-2
    lp = float((synth(amps + eps * d, hd, f0) * g1).sum())
    lm = float((synth(amps - eps * d, hd, f0) * g1).sum())
  fd = (lp - lm) / (2 * eps)
  an = float((a1 * d).sum())
  assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0)
  # first two clips against the fp64 analytic oracle
  ga, gh = O.harmonic_backward(npy(amps[:2].detach()), npy(hd[:2].detach()), npy(f0[:2]), npy(g1[:2]), n, 16000)
  np.testing.assert_allclose(npy(a1[:2]), ga, rtol=0, atol=grad_tol(ga))
  np.testing.assert_allclose(npy(h1[:2]), gh, rtol=0, atol=grad_tol(gh))


@pytest.mark.parametrize('batch,n_frames,n,scale,given_noise', [
    (2, 25, 1600, True, True),          # hop 64, noise supplied
    (2, 40, 2560 - 17, True, False),    # ragged tail, noise regenerated on chip by the same Philox counters
    (1, 50, 50 * 192 - 5, True, False), # frame size 192
    (2, 30, 30 * 80, False, True),      # frame size 80 (not a multiple of 64), scale_fn=None
    (3, 130, 130 * 64, True, False),    # several blocks per row
    (2, 45, 45 * 128 - 60, False, True),   # frame size 128 (two pieces per frame), noise supplied, no scale function
    (1, 21, 21 * 256, True, False)])       # frame size 256: tiles of 16 frames
def test_filtered_noise_backward_vs_analytic_oracle(ddsp, noise_kernel, batch, n_frames, n, scale, given_noise):
  rng = np.random.default_rng(n_frames)
  mags = (rng.standard_normal((batch, n_frames, 65)) + (4.0 if scale else 0.0)).astype(np.float32)
  if not scale:
    mags = np.abs(mags)
  g = rng.standard_normal((batch, n)).astype(np.float32)
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, scale_fn=ddsp.core.exp_sigmoid if scale else None,
                                    seed=11)
  if given_noise:
    noise = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
  else:
    noise = O.device_uniform_noise(batch, n, seed=11)        # call counter 0: the first call of this synth
  tm = ddsp.core.tf_float32(mags).requires_grad_(True)
  audio = synth(tm, noise=noise if given_noise else None)
  (audio * ddsp.core.tf_float32(g)).sum().backward()
  ref = O.filtered_noise_backward(mags, noise, g, 0, O.exp_sigmoid if scale else None)
  np.testing.assert_allclose(npy(tm.grad), ref, rtol=0, atol=1e-6 + 2e-5 * np.abs(ref).max())
  np.testing.assert_allclose(npy(audio), O.filtered_noise(mags, noise, 0, O.exp_sigmoid if scale else None,
                                                          dtype=np.float64), rtol=0, atol=2e-6 + 1e-5)
  # the same gradient bit for bit from a second call (the noise regenerated from the same counters)
  synth2 = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, scale_fn=ddsp.core.exp_sigmoid if scale else None, seed=11)
  tm2 = ddsp.core.tf_float32(mags).requires_grad_(True)
  (synth2(tm2, noise=noise if given_noise else None) * ddsp.core.tf_float32(g)).sum().backward()
  np.testing.assert_array_equal(npy(tm2.grad), npy(tm.grad))


def test_synth_backward_through_add(ddsp):
  rng = np.random.default_rng(4)
  b, f, k, n = 2, 50, 60, 3200
  amps = ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True)
  hd = ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True)
  mags = ddsp.core.tf_float32(rng.standard_normal((b, f, 65))).requires_grad_(True)
  f0 = 200 + rng.standard_normal((b, f, 1))
  harm = ddsp.synths.Harmonic(n_samples=n)
  noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)
  y = harm(amps, hd, f0) + noise(mags)                       # torch's own add: plumbing
  target = ddsp.core.tf_float32(rng.standard_normal((b, n)))
  loss = ((y - target) ** 2).mean()
  loss.backward()
  for t in (amps, hd, mags):
    assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0
  # a gradient step along -grad lowers the loss
  with torch.no_grad():
    noise2 = ddsp.synths.FilteredNoise(n_samples=n, window_size=0)     # same seed / call counter as `noise` had
    y2 = harm(amps - 50.0 * amps.grad, hd - 50.0 * hd.grad, f0) + noise2(mags - 50.0 * mags.grad)
    assert float(((y2 - target) ** 2).mean()) < float(loss)


def test_processor_group_is_trainable(ddsp):                         # ae.gin DAG + trainers.py:162-171
  rng = np.random.default_rng(6)
  b, f, k, n = 2, 50, 60, 3200
  feats = {'amps': ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True),
           'harmonic_distribution': ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True),
           'f0_hz': ddsp.core.tf_float32(200 + rng.standard_normal((b, f, 1))),
           'magnitudes': ddsp.core.tf_float32(rng.standard_normal((b, f, 65))).requires_grad_(True)}
  dag = [(ddsp.synths.Harmonic(n_samples=n), ['amps', 'harmonic_distribution', 'f0_hz']),
         (ddsp.synths.FilteredNoise(n_samples=n, window_size=0), ['magnitudes']),
         (ddsp.processors.Add(), ['filtered_noise/signal', 'harmonic/signal'])]
  audio = ddsp.processors.ProcessorGroup(dag=dag)(feats)
  assert audio.requires_grad
  audio.pow(2).mean().backward()
  for key in ('amps', 'harmonic_distribution', 'magnitudes'):
    g = feats[key].grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
  assert feats['f0_hz'].grad is None


@pytest.mark.parametrize('batch,n,l,ir_batch,add_dry,ir_rank', [
    (2, 3000, 700, 2, True, 2), (3, 5000, 6000, 1, False, 1), (2, 64000, 48000, 2, True, 3)])
def test_reverb_backward_vs_analytic_oracle(ddsp, batch, n, l, ir_batch, add_dry, ir_rank):
  rng = np.random.default_rng(l)
  x = rng.standard_normal((batch, n)).astype(np.float32)
  h = (rng.standard_normal((ir_batch, l)) * np.exp(-np.arange(l) / (0.3 * l))).astype(np.float32)
  g = rng.standard_normal((batch, n)).astype(np.float32)
  h_in = h[0] if ir_rank == 1 else (h[:, :, None] if ir_rank == 3 else h)
  tx = ddsp.core.tf_float32(x).requires_grad_(True)
  th = ddsp.core.tf_float32(h_in).requires_grad_(True)
  rev = ddsp.effects.Reverb(add_dry=add_dry)
  out = rev(tx, th)
  (out * ddsp.core.tf_float32(g)).sum().backward()
  if n * l <= 4e7:
    dx, dh = O.reverb_backward(x, h, g, add_dry)
  else:                                                   # full size: fp64 FFT correlations (the direct form takes minutes)
    import scipy.signal
    hm = h.astype(np.float64).copy(); hm[:, 0] = 0.0
    dx = np.stack([scipy.signal.fftconvolve(g[b].astype(np.float64)[::-1], hm[b % ir_batch])[:n][::-1]
                   for b in range(batch)]) + (g if add_dry else 0.0)
    dh = np.stack([scipy.signal.fftconvolve(g[b].astype(np.float64), x[b].astype(np.float64)[::-1])[n - 1:n - 1 + l]
                   for b in range(batch)])
    dh[:, 0] = 0.0
    if ir_batch == 1:
      dh = dh.sum(0, keepdims=True)
  np.testing.assert_allclose(npy(tx.grad), dx, rtol=0, atol=reverb_tol(dx))
  assert tuple(th.grad.shape) == tuple(th.shape)
  np.testing.assert_allclose(npy(th.grad).reshape(dh.shape), dh, rtol=0, atol=reverb_tol(dh))
  np.testing.assert_array_equal(npy(out), npy(rev(x, h_in)))        # same forward value without recording


def test_synth_plus_trainable_reverb_trains(ddsp):                   # solo_instrument.gin:26-40 + trainers.py:162-171
  rng = np.random.default_rng(10)
  b, f, k, n, l = 2, 50, 60, 3200, 1500
  amps = ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True)
  hd = ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True)
  f0 = 200 + rng.standard_normal((b, f, 1))
  rev = ddsp.effects.Reverb(trainable=True, reverb_length=l)
  rev.build(device=torch.device(DEV))
  rev._ir = ddsp.core.tf_float32(0.02 * rng.standard_normal(l)).requires_grad_(True)
  dag = [(ddsp.synths.Harmonic(n_samples=n), ['amps', 'harmonic_distribution', 'f0_hz']),
         (rev, ['harmonic/signal'])]
  group = ddsp.processors.ProcessorGroup(dag=dag)
  target = ddsp.core.tf_float32(rng.standard_normal((b, n)) * 0.1)
  def loss_of():
    return ((group({'amps': amps, 'harmonic_distribution': hd, 'f0_hz': f0}) - target) ** 2).mean()
  loss = loss_of()
  loss.backward()
  for t in (amps, hd, rev._ir):
    assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0
  with torch.no_grad():
    for t in (amps, hd, rev._ir):
      t -= 20.0 * t.grad
  with torch.no_grad():
    assert float(loss_of()) < float(loss.detach())


# ---- SpectralLoss backward: |ours - fp64 analytic gradient| <= 2e-4 * max|grad| + 1e-9 ------------------

# ---- SpectralLoss on random short clips: the draw and the checks of tools/fuzz_parity.py's `loss` family, and the seeds its round-5
# campaigns ended on as regression cases -----------------------------------------------------------------------
LOSS_FUZZ_SIZES = [4096, 2048, 1024, 512, 256, 128, 64, 32, 16, 6144, 3072, 1536, 768, 384, 192, 96, 48]   # (3 * 2**k: vst_48k.gin's kind)


def draw_loss_case(rng):
  """One random (batch, clip length, subset of frame sizes, weights) with its two signals - the draw order is part of the
  replay contract of tools/fuzz_parity.py (--replay loss:<seed>)."""
  b = int(rng.integers(1, 10))
  n = int(rng.choice([17, 64, 100, 1023, 1024, 1025, 3000, 12345, 20000, int(rng.integers(16, 30000))]))
  sizes = tuple(int(s) for s in rng.permutation(LOSS_FUZZ_SIZES)[:int(rng.integers(1, 7))])
  mw, lw = float(rng.choice([1.0, 0.0, 0.5])), float(rng.choice([1.0, 0.0, 0.5]))
  if mw == 0.0 and lw == 0.0:
    mw = 1.0
  t = (0.3 * rng.standard_normal((b, n))).astype(np.float32)
  a = (0.8 * t + 0.05 * rng.standard_normal((b, n))).astype(np.float32)
  if n >= 64:
    a[0, n // 2: n // 2 + n // 8] = 0.0
  return dict(batch=b, n=n, sizes=sizes, mag_weight=mw, logmag_weight=lw, target=t, audio=a)


def loss_gradient_excess(grad, ref, env):
  """(tolerance, by how much every sample of `grad` lies outside the subdifferential [ref - env, ref + env]); samples under a frame
  with a bin at fp32's noise floor (env = inf) are not held to anything.  tests/test_oracle.py checks that this does not make the
  comparison vacuous: a frame dropped from the overlap-add, a stretch of halved samples or a missing scale are all caught."""
  atol = 1e-9 + 2e-4 * np.abs(ref).max()
  held = np.isfinite(env)
  return atol, np.where(held, np.maximum(np.abs(grad - ref) - 1.05 * np.where(held, env, 0.0), 0.0), 0.0)


def check_loss_case(ddsp, case):
  """Value and gradient of SpectralLoss against exact arithmetic; returns the worst error over its tolerance.

  Value: 5e-5 relative on clips of >= 4096 samples; on shorter ones - a few samples under the first points of a long
  window - 1e-4, or THREE TIMES the error of the reference's own fp32 arithmetic if that is larger (the faithful oracle runs
  tf.signal.hann_window's fp32 op order: 0.5 - 0.5 cos(x) is good to 2e-4 at sample 17 of 4096 points, in TensorFlow as anywhere).
  Gradient: against the SUBDIFFERENTIAL (oracle.spectral_loss_backward, fp32_envelope: bins whose |X_t| - |X_a| or |X_a| is below
  what fp32 knows a magnitude to are left out of the reference and their largest possible contribution is allowed for, sample by
  sample) - 999 samples in 1000 within 2e-4 of the largest gradient, none beyond 30 x that.  Rounds 4-5 checked the median and the
  80th percentile of the plain difference and ended their campaigns on "known" failures where one flipped bin covered a fifth of
  a short clip."""
  t, a, sizes, mw, lw, n = case['target'], case['audio'], case['sizes'], case['mag_weight'], case['logmag_weight'], case['n']
  loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, mag_weight=mw, logmag_weight=lw)
  ta = ddsp.core.tf_float32(a).requires_grad_(True)
  val = loss(t, ta)
  val.backward()
  ref_v = float(O.spectral_loss(t, a, sizes, mag_weight=mw, logmag_weight=lw, dtype=np.float64))
  vtol = 5e-5
  if n < 4096:
    ref_v32 = float(O.spectral_loss(t, a, sizes, mag_weight=mw, logmag_weight=lw, dtype=np.float32))
    vtol = max(1e-4, 3.0 * abs(ref_v32 - ref_v) / max(abs(ref_v), 1e-12))
  ev = abs(float(val.detach()) - ref_v) / max(abs(ref_v), 1e-12)
  fwd = float(loss(t, a))                                           # (the forward-only kernel)
  if ev > vtol or abs(fwd - ref_v) > vtol * abs(ref_v):
    # A spectral NULL: core.safe_log passes any positive magnitude, and the log of a bin that cancels to 2e-7 is the log of what
    # rounding left of it - in TensorFlow's fp32 as in any (fuzz seed loss:107037044, round 6: the Nyquist bin of one frame of 256
    # samples, exact 2.46e-7, the reference's own op order 4.75e-7, the MI355X 6e-9: 1.56e-4 of the loss).  Only then: the value is
    # held to the interval oracle.spectral_loss_value_bounds derives - every other bin to what fp32 knows its magnitudes to, the
    # null bins' terms to "not negative, and not 110 nats above exact".  tests/test_oracle.py: without a null the interval is the
    # value +- 2.5e-4 of it, and a null opens it upwards only.
    lo, hi, n_null = O.spectral_loss_value_bounds(t, a, sizes, mw, lw)
    assert n_null > 0, ('loss value', float(val.detach()), fwd, ref_v, vtol)
    for v in (float(val.detach()), fwd):
      assert lo - vtol * abs(ref_v) <= v <= hi + vtol * abs(ref_v), ('loss value with %d null bin(s)' % n_null, v, lo, hi, ref_v)
    ev = min(ev, vtol)
  if n < 256:
    return ev / vtol
  ref, env = O.spectral_loss_backward(t, a, sizes, mw, lw, fp32_envelope=5e-6)
  atol, err = loss_gradient_excess(npy(ta.grad), ref, env)
  # 999 samples in 1000 within the tolerance and none beyond 30 x (what fp32 rounding adds up to in the worst sample of 10^5 under
  # 1 / |X| amplification: measured up to 3 x, once 23 x, in 7000 random cases).  Anything structural - a frame missing from the
  # overlap-add, a wrong window, a wrong hop - moves at least a frame's worth of samples by the gradient's own magnitude, which is
  # 5000 x the tolerance.
  worst, bulk = float(err.max()), float(np.quantile(err, 0.999))
  assert bulk <= atol and worst <= 30.0 * atol, ('loss gradient', bulk, worst, atol, float(np.abs(ref).max()))
  # ... which the envelope must not hide: it stays below a twentieth of the gradient's magnitude on most samples of the rows without
  # the silent stretch (every row but the first)
  # (clips of at least four of their longest frames: one frame at the noise floor cannot cover such a row)
  if case['batch'] > 1 and n >= 4 * max(sizes):
    small = float((env[1:] <= 0.05 * np.abs(ref).max()).mean())
    assert small >= 0.5, ('the envelope hides the gradient', small)
  return max(ev / vtol, bulk / atol, worst / (30.0 * atol))


@pytest.mark.parametrize('seed', [37014845,                                  # round 5's last campaign: a gradient bulk at n = 1025
                                  31000384, 31003046, 31012330,             # seed 31: values of 17-sample clips under 4096 / 6144 points
                                  4007713, 4032057, 4014585, 4029689,        # seed 4: two gradient bulks, two more 17-sample clips
                                  47050735,                                  # round 6, seed 47: the mag term's phasor at a DC bin that cancels to 2.5e-8
                                  107037044])                                # round 6, seed 107: the logmag VALUE with a Nyquist bin that cancels to 2.5e-7
def test_spectral_loss_seeds_the_round_5_campaigns_ended_on(ddsp, seed):
  case = draw_loss_case(np.random.default_rng(seed))
  assert check_loss_case(ddsp, case) <= 1.0

@pytest.mark.parametrize('batch,n,sizes', [(2, 3000, (2048, 1024, 512, 256, 128, 64)), (1, 777, (64, 16)),
                                           (3, 20000, (4096, 512)),
                                           (2, 1500, (32, 1024))])      # 32: the one transform plan without a radix-8 stage
def test_spectral_loss_backward_vs_analytic_oracle(ddsp, batch, n, sizes):
  rng = np.random.default_rng(n)
  t = (0.3 * rng.standard_normal((batch, n))).astype(np.float32)
  a = (0.8 * t + 0.05 * rng.standard_normal((batch, n))).astype(np.float32)
  a[0, n // 2: n // 2 + n // 8] = 0.0                              # exact zeros: no gradient through |.| and safe_log there
  loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, mag_weight=1.0, logmag_weight=0.5)
  ta = ddsp.core.tf_float32(a).requires_grad_(True)
  val = loss(t, ta)
  (3.0 * val).backward()
  ref = 3.0 * O.spectral_loss_backward(t, a, sizes, 1.0, 0.5)
  # d|x|/dx is a sign: where a bin's two magnitudes (or their logs) agree to rounding, fp32 and fp64 may pick
  # different signs, which moves that one frame's samples by a few times the tolerance (seen in the last, mostly
  # zero-padded frame of the 20 000-sample case under the CPU emulation's exact sin / cos).  Such a frame holds at
  # most 0.1 % of the samples; everything else must meet the tolerance, and nothing may be far off.
  atol = 1e-9 + 2e-4 * np.abs(ref).max()
  err = np.abs(npy(ta.grad) - ref)
  assert (err > atol).mean() <= 1e-3 and err.max() <= 10 * atol, (float((err > atol).mean()), float(err.max()), atol)
  np.testing.assert_allclose(float(val.detach()), float(O.spectral_loss(t, a, sizes, logmag_weight=0.5, dtype=np.float64)),
                             rtol=2e-5)


def test_training_loop_with_native_loss(ddsp):                       # ae.gin: synths -> SpectralLoss, all kernels native
  rng = np.random.default_rng(21)
  b, f, k, n = 2, 50, 40, 3200
  f0 = 220 + rng.standard_normal((b, f, 1))
  harm = ddsp.synths.Harmonic(n_samples=n)
  loss_fn = ddsp.losses.SpectralLoss(fft_sizes=(512, 256, 128, 64), mag_weight=1.0, logmag_weight=1.0)
  with torch.no_grad():
    target = harm(rng.standard_normal((b, f, 1)) + 1.0, rng.standard_normal((b, f, k)), f0)
  amps = ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True)
  hd = ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True)
  opt = torch.optim.Adam([amps, hd], lr=0.05)
  history = []
  for _ in range(30):
    opt.zero_grad()
    loss = loss_fn(target, harm(amps, hd, f0))
    loss.backward()
    opt.step()
    history.append(float(loss.detach()))
  assert history[-1] < 0.7 * history[0], history


def test_fir_filter_and_filtered_noise_reverb(ddsp):                 # effects.py:200-322
  rng = np.random.default_rng(17)
  b, n, f, m = 2, 3200, 50, 33
  audio = rng.standard_normal((b, n)).astype(np.float32)
  mags = rng.standard_normal((b, f, m)).astype(np.float32)
  out = npy(ddsp.effects.FIRFilter(window_size=33)(audio, mags))
  ref = O.frequency_filter(audio, O.exp_sigmoid(mags, dtype=np.float64), window_size=33, dtype=np.float64)
  np.testing.assert_allclose(out, ref, rtol=0, atol=noise_tol(ref))
  # FilteredNoiseReverb: IR = FilteredNoise(n_samples=reverb_length)(magnitudes), then Reverb
  l, fr, nb = 2400, 50, 16
  rmags = rng.standard_normal((b, fr, nb)).astype(np.float32)
  rev = ddsp.effects.FilteredNoiseReverb(reverb_length=l, window_size=17, n_frames=fr, n_filter_banks=nb, seed=5)
  got = npy(rev(audio, rmags))
  noise = O.device_uniform_noise(b, l, seed=5)                       # call counter 0 of the inner synth
  ir = O.filtered_noise(rmags, noise, 17, O.exp_sigmoid, initial_bias=-3.0, dtype=np.float64)
  ref = O.reverb(audio, ir, add_dry=True, dtype=np.float64)
  np.testing.assert_allclose(got, ref, rtol=0, atol=reverb_tol(ref))
  with pytest.raises(ValueError, match='Must provide "magnitudes" tensor'):
    ddsp.effects.FilteredNoiseReverb().get_controls(audio)


@pytest.mark.parametrize('batch,n_frames,m,n,window_size,scale', [
    (2, 25, 33, 1600, 17, True),        # cropped IR (window < 2(M-1)), 33 taps
    (1, 500, 32, 24000, 257, True),     # vst.gin's FilteredNoiseReverb shape: frame size 48, 62 taps
    (2, 7, 9, 100, 0, False),           # ragged tail, tiny IR, scale_fn=None
    (2, 10, 129, 2560, 0, True)])       # 256 taps: two tap chunks per frame
def test_filtered_noise_backward_generic_shapes(ddsp, batch, n_frames, m, n, window_size, scale):
  rng = np.random.default_rng(m)
  mags = (rng.standard_normal((batch, n_frames, m)) + (2.0 if scale else 0.0)).astype(np.float32)
  if not scale:
    mags = np.abs(mags)
  noise = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
  g = rng.standard_normal((batch, n)).astype(np.float32)
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=window_size,
                                    scale_fn=ddsp.core.exp_sigmoid if scale else None, initial_bias=-3.0)
  tm = ddsp.core.tf_float32(mags).requires_grad_(True)
  (synth(tm, noise=noise) * ddsp.core.tf_float32(g)).sum().backward()
  ref = O.filtered_noise_backward(mags, noise, g, window_size, O.exp_sigmoid if scale else None, initial_bias=-3.0)
  np.testing.assert_allclose(npy(tm.grad), ref, rtol=0, atol=1e-6 + 5e-5 * np.abs(ref).max())


def test_vst_dag_with_trainable_filtered_noise_reverb_trains(ddsp):   # gin/models/vst/vst.gin:66-87
  rng = np.random.default_rng(30)
  b, f, k, m, n = 2, 50, 60, 65, 3200
  feats = {'amps': ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True),
           'harmonic_distribution': ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True),
           'f0_hz': ddsp.core.tf_float32(200 + rng.standard_normal((b, f, 1))),
           'noise_magnitudes': ddsp.core.tf_float32(rng.standard_normal((b, f, m))).requires_grad_(True)}
  rev = ddsp.effects.FilteredNoiseReverb(trainable=True, reverb_length=1200, n_frames=25, n_filter_banks=32,
                                         name='reverb')
  rev.build(device=torch.device(DEV))
  rev._magnitudes.requires_grad_(True)
  crop = ddsp.processors.Crop(frame_size=64, crop_location='back')
  dag = [(ddsp.synths.Harmonic(n_samples=n), ['amps', 'harmonic_distribution', 'f0_hz']),
         (ddsp.synths.FilteredNoise(n_samples=n, window_size=0), ['noise_magnitudes']),
         (ddsp.processors.Add(), ['filtered_noise/signal', 'harmonic/signal']),
         (rev, ['add/signal']),
         (crop, ['reverb/signal'])]
  group = ddsp.processors.ProcessorGroup(dag=dag)
  audio = group(feats)
  assert tuple(audio.shape) == (b, n - 64) and audio.requires_grad
  target = ddsp.core.tf_float32(0.1 * rng.standard_normal((b, n - 64)))
  loss = ddsp.losses.SpectralLoss(fft_sizes=(256, 128, 64), logmag_weight=1.0)(target, audio.contiguous())
  loss.backward()
  for t in (feats['amps'], feats['harmonic_distribution'], feats['noise_magnitudes'], rev._magnitudes):
    assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0


def test_vst_48k_configuration_full_size(ddsp):                     # gin/models/vst/vst_48k.gin:14-17, 50-112
  """The one shipped configuration beyond the others' limits until the end of round 5: 48 kHz, clips of 192 960 samples on 201
  frames (hop 960), 'linear' envelopes with the angular cumsum, a trainable FilteredNoiseReverb of 72 000 taps (18 partitions of
  the FFT convolution) and a SpectralLoss on frames of 6144 .. 192 samples (3 * 2^k: the enclosing power-of-two transforms).
  Forward through the DAG, the loss against the fp64 oracle on the audio the DAG made, gradients to every trainable input."""
  rng = np.random.default_rng(48)
  b, f, k, m, n, sr = 2, 201, 60, 65, 192960, 48000
  feats = {'amps': ddsp.core.tf_float32(rng.standard_normal((b, f, 1))).requires_grad_(True),
           'harmonic_distribution': ddsp.core.tf_float32(rng.standard_normal((b, f, k))).requires_grad_(True),
           'f0_hz': ddsp.core.tf_float32(220 + 2 * rng.standard_normal((b, f, 1))),
           'noise_magnitudes': ddsp.core.tf_float32(rng.standard_normal((b, f, m))).requires_grad_(True)}
  rev = ddsp.effects.FilteredNoiseReverb(trainable=True, reverb_length=72000, n_frames=500, n_filter_banks=32,
                                         initial_bias=-4.0, name='reverb')
  rev.build(device=torch.device(DEV))
  rev._magnitudes.requires_grad_(True)
  dag = [(ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method='linear', use_angular_cumsum=True),
          ['amps', 'harmonic_distribution', 'f0_hz']),
         (ddsp.synths.FilteredNoise(n_samples=n, window_size=0), ['noise_magnitudes']),
         (ddsp.processors.Add(), ['filtered_noise/signal', 'harmonic/signal']),
         (rev, ['add/signal']),
         (ddsp.processors.Crop(frame_size=960, crop_location='back'), ['reverb/signal'])]
  audio = ddsp.processors.ProcessorGroup(dag=dag)(feats)
  assert tuple(audio.shape) == (b, n - 960) and audio.requires_grad and bool(torch.isfinite(audio).all())
  target = ddsp.core.tf_float32(0.1 * rng.standard_normal((b, n - 960)))
  sizes = (6144, 3072, 1536, 768, 384, 192)
  loss = ddsp.losses.SpectralLoss(fft_sizes=sizes, loss_type='L1', mag_weight=1.0, logmag_weight=1.0)(target, audio.contiguous())
  ref = float(O.spectral_loss(npy(target), npy(audio.detach()), sizes, mag_weight=1.0, logmag_weight=1.0, dtype=np.float64))
  np.testing.assert_allclose(float(loss.detach()), ref, rtol=5e-5)
  loss.backward()
  for t in (feats['amps'], feats['harmonic_distribution'], feats['noise_magnitudes'], rev._magnitudes):
    assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0


def test_spectral_loss_fused_and_separate_gradient_entries_agree(ddsp):
  rng = np.random.default_rng(44)
  t = ddsp.core.tf_float32(0.3 * rng.standard_normal((2, 5000)))
  a = ddsp.core.tf_float32(0.25 * rng.standard_normal((2, 5000)))
  loss = ddsp.losses.SpectralLoss(fft_sizes=(1024, 256, 64), logmag_weight=1.0)
  value, grad = loss._value_and_grad(t, a)                          # ddsp_spectral_loss_value_and_grad_f32
  sep = loss._backward(t, a, torch.ones((), device=DEV))         # ddsp_spectral_loss_backward_f32
  fwd = loss._forward(t, a)                                         # ddsp_spectral_loss_f32
  np.testing.assert_allclose(float(value), float(fwd), rtol=1e-6)
  scale = float(grad.abs().max())
  assert float((grad - sep).abs().max()) <= 1e-5 * scale            # same kernel, atomics order only


# ---- random shapes: whatever the hand-picked cases above did not think of ---------------------------------------------
def _harmonic_exact(amps, hd, f0, n, sr, method, with_knife_edges=False):
  """Harmonic.__call__ in exact (fp64) arithmetic, written from the formulas alone (linear f0 ramps with weights r / hop,
  raised-cosine or linear amplitude envelopes, inclusive cumsum, frame- and audio-rate Nyquist masks): independent of
  oracle/ddsp_oracle.py, whose "truth" mode keeps TF's fp32 resize POSITIONS (t * scale rounded to fp32)."""
  b, f, k = hd.shape
  hop = n // f
  a = O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64)
  h = O.exp_sigmoid(hd.astype(np.float64), dtype=np.float64)
  fr = f0.astype(np.float64)
  kk = np.arange(1, k + 1)
  h = np.where(fr * kk >= sr / 2, 0.0, h)
  s = h.sum(-1, keepdims=True)
  h = h / np.where(s == 0, 1e-7, s)
  amp = a * h
  t = np.arange(n)
  j, r = t // hop, t % hop
  j1 = np.minimum(j + 1, f - 1)
  lerp = r / hop
  ft = fr[:, j, 0] + (fr[:, j1, 0] - fr[:, j, 0]) * lerp[None]
  ph = np.cumsum(ft / sr, axis=1)
  wn = lerp if method == 'linear' else 0.5 - 0.5 * np.cos(np.pi * lerp)
  out = np.zeros((b, n))
  out32 = np.zeros((b, n))                   # the same sum with the audio-rate mask decided as the reference decides it
  knife = np.zeros((b, n), dtype=bool)
  hits = np.zeros((b, n), dtype=bool)        # exactly on Nyquist on a sloping ramp (with_knife_edges='and exact hits')
  # the reference's decision (core.py:942-944 on what core.resample made of get_harmonic_frequencies' fp32 products): legacy
  # bilinear resize, top + (bottom - top) * lerp with every step rounded to fp32, compared with fl32(sr / 2).
  # lerp = r / hop is exact in fp32 for frame sizes that are powers of two; for others (192) TF's position t * fl32(F / N) is up
  # to 1.5e-5 off and the kernels' r * fl32(1 / hop) is what is restated here (known limits).
  f32 = np.float32
  fr32 = f0.astype(f32)[:, :, 0]
  lerp32 = (r.astype(f32) * (f32(1.0) / f32(hop))).astype(f32)
  for q in range(1, k + 1):
    aq = amp[:, j, q - 1] * (1 - wn)[None] + amp[:, j1, q - 1] * wn[None]
    sq = np.sin(2 * np.pi * q * ph)
    out += np.where(ft * q >= sr / 2, 0.0, aq) * sq
    top, bot = (fr32[:, j] * f32(q)).astype(f32), (fr32[:, j1] * f32(q)).astype(f32)
    fk32 = (top + ((bot - top).astype(f32) * lerp32[None]).astype(f32)).astype(f32)
    out32 += np.where(fk32 >= f32(sr / 2.0), 0.0, aq) * sq
    # samples where a harmonic sits within fp32 rounding of Nyquist: the reference's fp32 comparison (which the kernels
    # reproduce) and the fp64 one may fall on different sides
    d = ft * q - sr / 2                      # (exactly on Nyquist - f0 = 200 Hz, harmonic 40 - is not ambiguous: both say >=)
    knife |= (d != 0) & (np.abs(d) <= 4e-7 * sr)
    # A SLOPING ramp that lands exactly on Nyquist: this function and the kernels' fp32 op order agree on it (masked), but an
    # evaluation at TF's fp32 resize POSITION - oracle.harmonic_backward's: fl32(4096 fl32(1 / 192)) = 21.333334 for 21 1/3 - finds
    # the harmonic 1e-6 Hz to one side or the other.  Not knife edges for the forward comparisons; comparisons against that oracle
    # leave them out as well (fuzz seed harmonic_bwd:41016811, round 6: frames of 192 samples, r = 64, top - bot divisible by three)
    hits |= (d == 0) & (top != bot)
  if with_knife_edges == 'fp32 mask':
    return out, knife, out32
  if with_knife_edges == 'and exact hits':
    return out, knife, hits
  return (out, knife) if with_knife_edges else out


def assert_knife_edges_take_the_fp32_side(got, exact32, knife, atol, what):
  """the samples the exact-arithmetic comparison leaves out - a harmonic within fp32 rounding of Nyquist,
  where the fp64 comparison and the reference's fp32 one may fall on different sides - are held to the SAME sum with the mask
  decided in the reference's fp32 op order (the crossing harmonic's amplitude present or absent: a difference of that
  harmonic's whole amplitude, orders of magnitude above the tolerance, if the kernel decided otherwise).  With this every
  sample is checked against something; and away from the knife edges the two references are the same sum."""
  if knife.any():
    err = np.abs(got - exact32)[knife]
    assert err.max() <= atol, (what, 'knife-edge samples', int(knife.sum()), float(err.max()), atol)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_harmonic_random_shapes_vs_exact_arithmetic(ddsp, seed):
  """Random (batch, frames, frame size 64 / 128 / 192, K, f0 regime, envelope method) through the default kernels
  against exact arithmetic, HARM_TABLE_ATOL.  Against the oracle's truth mode too when the frame size is a power of two;
  for 192 that mode (TF's fp32 positions: up to 1.5e-5 off r / hop late in a clip) is up to 5e-4 away from exact
  arithmetic when f0 jumps by tens of Hz from frame to frame - from the kernels, which use r / hop, just as far
  (known limits)."""
  rng = np.random.default_rng(seed)
  cases = 12 if DEV == 'cuda' else 3                         # (the emulated run keeps three)
  for _ in range(cases):
    hop = int(rng.choice([64, 64, 128, 192]))
    f = int(rng.integers(1, 120 if DEV == 'cuda' else 24))
    n = f * hop
    k = 4 * int(rng.integers(1, 33)) if rng.integers(0, 2) else int(rng.integers(1, 129))       # any count up to 128
    b = int(rng.integers(1, 4))
    sr = 16000
    base = float(rng.choice([40.0, 70.0, 200.0, 440.0, 1000.0, sr / 2 / k * 0.999]))
    f0 = np.abs(base + rng.standard_normal((b, f, 1)) * float(rng.choice([0.0, 1.0, 30.0]))).astype(np.float32)
    amps = rng.standard_normal((b, f, 1)).astype(np.float32)
    hd = rng.standard_normal((b, f, k)).astype(np.float32)
    method = str(rng.choice(['window', 'linear']))
    got = npy(ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)(amps, hd, f0))
    what = dict(hop=hop, frames=f, k=k, batch=b, base=base, method=method)
    scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
    exact, knife, exact32 = _harmonic_exact(amps, hd, f0, n, sr, method, with_knife_edges='fp32 mask')
    assert knife.mean() <= 1e-2, what                        # (a handful of samples per clip at most)
    assert np.abs(got - exact)[~knife].max() <= HARM_TABLE_ATOL * scale, what
    assert_knife_edges_take_the_fp32_side(got, exact32, knife, HARM_TABLE_ATOL * scale, what)
    if hop in (64, 128):
      truth = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=sr, amp_resample_method=method, dtype=np.float64)
      assert np.abs(got - truth)[~knife].max() <= HARM_TABLE_ATOL * scale, what


@pytest.mark.parametrize('k,method', [(99, 'window'), (160, 'linear')])
def test_harmonic_sweep_through_nyquist_on_frames_of_eight(ddsp, k, method):
  """tools/fuzz_parity.py harmonic:67005602 / 67051142 (round 6, seed 67's two stops): a 60 -> 900 Hz sweep over 64 frames of EIGHT
  samples at 44.1 kHz takes harmonics 25 .. k through Nyquist inside 512 samples - one crossing sample per harmonic, a dozen of them
  within fp32 rounding of Nyquist (2-3 % of the clip: the tool's non-vacuity guard stopped there, the kernels were never compared).
  Every other sample against exact arithmetic, the knife edges on the reference's fp32 side; the guard's new form - one knife-edge
  sample per (row, harmonic) that crosses - holds with room."""
  hop, f, sr, b = 8, 64, 44100, 1
  n = f * hop
  rng = np.random.default_rng(67000000 + k)
  f0 = (np.linspace(60.0, 900.0, f)[None, :, None] * np.ones((b, 1, 1))).astype(np.float32)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  got = npy(ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)(amps, hd, f0))
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  exact, knife, exact32 = _harmonic_exact(amps, hd, f0, n, sr, method, with_knife_edges='fp32 mask')
  fk = f0.astype(np.float64) * np.arange(1, k + 1, dtype=np.float64)[None, None, :]
  crossing = int(((fk.min(axis=1) < 0.5 * sr) & (fk.max(axis=1) >= 0.5 * sr)).sum())
  assert crossing >= 70 and int(knife.sum()) <= crossing and knife.mean() <= 0.1
  atol = (HARM_TABLE_ATOL if k <= 200 else HARM_TRUTH_ATOL) * scale
  assert np.abs(got - exact)[~knife].max() <= atol
  assert_knife_edges_take_the_fp32_side(got, exact32, knife, atol, dict(k=k, method=method))


def test_streaming_wild_f0_on_frames_of_252_samples(ddsp):
  """tools/fuzz_parity.py streaming:89027392 (round 6): four calls of core.streaming_harmonic_synthesis, 5 frames of 252 samples,
  60 harmonics, f0 drawn U(60, 600) Hz PER FRAME.  The reference's frequency envelope takes TF's fp32 resize position
  fl32(t fl32(5 / 1260)) (up to pos 2^-23 frames off); with steps of hundreds of Hz that alone moves the result by more than this
  family's tolerance - the fp64 oracle with exact positions against itself with TF's: 3.2e-4 on the second call.  The kernels
  take r / hop on clips that are whole frames (known limits): held to the tolerance against the exact-position oracle,
  phase carried call to call, and the distance to the TF-position oracle explained by that oracle pair's own distance."""
  rng = np.random.default_rng(89027392)
  b, f = int(rng.integers(1, 3)), int(rng.integers(2, 6))
  k = int(rng.choice([1, 20, 60, 100]))
  n = int(rng.choice([64, 320, 512, 1000, int(rng.integers(f, 1500))]))
  sr = int(rng.choice([16000, 48000]))
  method = str(rng.choice(['linear', 'linear', 'window', 'nearest', 'cubic']))
  n = f * int(rng.integers(2, 300))
  calls = int(rng.integers(1, 5)); with_hd = bool(rng.integers(0, 4))
  assert (b, f, k, n, sr, method, calls, with_hd) == (1, 5, 60, 1260, 16000, 'window', 4, True)
  phase = np.zeros((b, 1, 1), np.float32); phase_tf = np.zeros((b, 1, 1)); phase_x = np.zeros((b, 1, 1))
  moved_most = 0.0
  for _ in range(calls):
    f0 = rng.uniform(60.0, 600.0, (b, f, 1)).astype(np.float32)
    amps = rng.uniform(0.1, 1.0, (b, f, 1)).astype(np.float32)
    hd = rng.uniform(0.0, 1.0, (b, f, k)).astype(np.float32)
    got, phase = ddsp.core.streaming_harmonic_synthesis(f0, amps, hd, initial_phase=phase, n_samples=n, sample_rate=sr,
                                                        amp_resample_method=method)
    ref_tf, phase_tf = O.streaming_harmonic_synthesis(f0, amps, hd, initial_phase=phase_tf, n_samples=n, sample_rate=sr,
                                                      amp_resample_method=method, dtype=np.float64)
    with O.exact_resize_positions():
      ref_x, phase_x = O.streaming_harmonic_synthesis(f0, amps, hd, initial_phase=phase_x, n_samples=n, sample_rate=sr,
                                                      amp_resample_method=method, dtype=np.float64)
    got, phase = npy(got), npy(phase)
    tol = 3e-4 * max(1.0, float(np.abs(ref_x).max()))
    moved = float(np.abs(ref_tf - ref_x).max())
    moved_most = max(moved_most, moved)
    assert np.abs(got - ref_x).max() <= 0.2 * tol                     # (measured 4e-6)
    assert np.abs(got - ref_tf).max() <= 0.2 * tol + moved            # ... and no further from TF's positions than they are from exact
    dphi = np.abs(((phase.astype(np.float64) - phase_x + np.pi) % (2 * np.pi)) - np.pi).max()
    assert dphi <= 2e-5
  assert 2.5e-4 < moved_most < 4e-4                                   # the premise: the position rounding alone is worth the tolerance


def test_materialised_chain_backward_with_a_quiet_knife_edge(ddsp):
  """tools/fuzz_parity.py harmonic_chain:83029253 (round 6): 6 frames -> 670 samples at 48 kHz, cubic envelopes, f0 = 440 Hz +- 1 %.
  Harmonic 54 of row 0 is 4.4e-4 Hz below Nyquist (1.8e-8 sr) at sample 144 in exact arithmetic and at or above it in the fp32
  comparison of the kernels and of the reference (core.py:942-944) - a knife edge, but a QUIET one: the harmonic is 2.0e-4 of the
  audio there, under the forward tolerance, so the tool's `e > atol` never set its cotangent aside, and dL/d harmonic_distribution
  [0, 2, 53] came out 6.2e-4 from the oracle's (tolerance 3.5e-4): one sample's worth of one harmonic.  With the knife edges found
  from the frequencies themselves the case sits at a twentieth of the tolerance; every other harmonic was there all along."""
  rng = np.random.default_rng(83029253)
  f = int(rng.integers(2, 40)); k = int(rng.choice([1, 7, 20, 60, 100])); b = int(rng.integers(1, 3)); sr = int(rng.choice([16000, 48000]))
  method = str(rng.choice(['nearest', 'cubic', 'linear'])); n = int(rng.integers(f, f * 120))
  base = float(rng.choice([70.0, 110.0, 220.0, 440.0]))
  assert (f, k, b, sr, method, n, base) == (6, 60, 2, 48000, 'cubic', 670, 440.0)
  f0 = np.abs(base * (1.0 + 0.01 * rng.standard_normal((b, f, 1)))).astype(np.float32)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)
  ta = ddsp.core.tf_float32(amps).requires_grad_(True)
  th = ddsp.core.tf_float32(hd).requires_grad_(True)
  audio = synth(ta, th, f0)
  g = rng.standard_normal((b, n)).astype(np.float32)
  truth = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=sr, amp_resample_method=method, dtype=np.float64)
  f_env = O.resample(f0.astype(np.float64), n, method='linear', dtype=np.float64)[:, :, 0]
  ks = np.arange(1, k + 1, dtype=np.float64)
  step = np.abs(np.diff(f0.astype(np.float64)[:, :, 0], axis=1)).max()
  d = np.abs(f_env[:, :, None] * ks[None, None, :] - 0.5 * sr)
  near = (d <= (4e-7 * sr + f * 2.0 ** -22 * step * ks)[None, None, :])
  assert near.sum() == 1 and near[0, 144, 53]                                   # the one knife edge, where the docstring says
  knife = near.any(axis=-1)
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  e = np.abs(npy(audio.detach()) - truth)
  assert e[~knife].max() <= 8 * HARM_TRUTH_ATOL * scale                         # (cubic envelopes: fuzz_parity's allowance)
  # at the knife edge the kernel took the fp32 side: harmonic 54 masked - the audio differs from the fp64 oracle's by that harmonic
  assert 1e-4 < e[0, 144] < 3e-4
  g[knife] = 0.0
  audio.backward(ddsp.core.tf_float32(g))
  ga, gh = O.harmonic_backward(amps, hd, f0, g, n, sr, O.exp_sigmoid, True, method)
  assert np.abs(npy(ta.grad) - ga).max() <= 1e-5 + 2e-4 * np.abs(ga).max()
  assert np.abs(npy(th.grad) - gh).max() <= 1e-5 + 2e-4 * np.abs(gh).max()


@pytest.mark.parametrize('k', [99, 61, 5, 1])
def test_harmonic_counts_that_are_not_multiples_of_four(ddsp, k):
  """The reference's own test shape has 99 harmonics (ddsp/processors_test.py:28-73): rows of the distribution that are
  not 16 bytes apart take the default (wavetable) kernel too - 4-byte row loads, the last lane's tail dead.  Signal and
  controls dict against the fp64 oracle; the same samples whether the controls are asked for or not."""
  b, f, hop = 2, 37, 64
  n = f * hop
  rng = np.random.default_rng(100 + k)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = np.abs(110.0 + 20.0 * rng.standard_normal((b, f, 1))).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n)
  got = npy(synth(amps, hd, f0))
  truth = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=16000, dtype=np.float64)
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  assert np.abs(got - truth).max() <= HARM_TABLE_ATOL * scale
  full = synth(amps, hd, f0, return_outputs_dict=True)
  np.testing.assert_array_equal(npy(full['signal']), got)
  ctl = O.harmonic_get_controls(amps, hd, f0, 16000, dtype=np.float64)
  np.testing.assert_allclose(npy(full['controls']['harmonic_distribution']), ctl['harmonic_distribution'], rtol=2e-5, atol=1e-7)
  np.testing.assert_allclose(npy(full['controls']['amplitudes']), ctl['amplitudes'], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize('k,hop,sr,base,jitter', [
    (200, 192, 48000, 100.0, 1.0),          # config 5's shape (gin/models/vst/vst_48k.gin:16-17,102): every harmonic live
    (200, 64, 48000, 119.0, 8.0),           # the top harmonics cross Nyquist inside frames (200 * 120 Hz = 24 kHz)
    (199, 192, 48000, 60.0, 0.0),           # rows that are not 16 bytes apart
    (129, 128, 16000, 40.0, 2.0),           # the smallest count on this path; 16 kHz: most harmonics above Nyquist
    (160, 192, 48000, 440.0, 30.0),         # scattered table reads, f0 jumping from frame to frame
])
def test_harmonic_129_to_200_harmonics_on_the_wavetable_kernel(ddsp, k, hop, sr, base, jitter):
  """129 .. 200 harmonics take the wavetable kernel too since round 3 (ten taps on the same 512 points, the constant
  factor streamed from L2, rows spread over whole wavefronts: csrc/harmonic_table.hip, WIDE): signal against exact
  arithmetic at HARM_TABLE_ATOL and within HARM_TRUTH_ATOL of the direct sum, the controls dict against the fp64 oracle,
  the same samples whether the controls are asked for or not, a row alone = the row in its batch, processors.Add fused."""
  b, f = 3, (45 if DEV == 'cuda' else 33)
  n = f * hop
  rng = np.random.default_rng(1000 + k + hop)
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  hd[1, :, k // 2:] += 4.0                                    # one clip with its weight in the top harmonics (1 / psi_hat <= 27 there)
  f0 = np.abs(base + jitter * rng.standard_normal((b, f, 1))).astype(np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  got = npy(synth(amps, hd, f0))
  exact, knife, exact32 = _harmonic_exact(amps, hd, f0, n, sr, 'window', with_knife_edges='fp32 mask')
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  assert knife.mean() <= 1e-2
  assert np.abs(got - exact)[~knife].max() <= HARM_TABLE_ATOL * scale
  assert_knife_edges_take_the_fp32_side(got, exact32, knife, HARM_TABLE_ATOL * scale, (k, hop, sr, base))
  direct = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  direct.kernel = 'direct'
  sum_ = npy(direct(amps, hd, f0))
  assert not np.array_equal(got, sum_)                         # (two kernels)
  assert np.abs(got - sum_)[~knife].max() <= HARM_TRUTH_ATOL * scale
  assert_knife_edges_take_the_fp32_side(sum_, exact32, knife, HARM_TRUTH_ATOL * scale, ('direct sum', k, hop, sr, base))
  full = synth(amps, hd, f0, return_outputs_dict=True)
  np.testing.assert_array_equal(npy(full['signal']), got)
  ctl = O.harmonic_get_controls(amps, hd, f0, sr, dtype=np.float64)
  np.testing.assert_allclose(npy(full['controls']['harmonic_distribution']), ctl['harmonic_distribution'], rtol=2e-5, atol=1e-7)
  np.testing.assert_allclose(npy(full['controls']['amplitudes']), ctl['amplitudes'], rtol=2e-5, atol=1e-7)
  np.testing.assert_array_equal(npy(synth(amps[1:2], hd[1:2], f0[1:2])), got[1:2])
  z = rng.standard_normal((b, n)).astype(np.float32)
  np.testing.assert_array_equal(npy(synth.call_add(amps, hd, f0, z)), got + z)


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_filtered_noise_random_shapes_vs_oracle(ddsp, seed):
  """Random (batch, frames, frame size 64 .. 256, ragged length, supplied or generated noise) through the default
  kernel (tile edges, the rows' short last tiles, frames that share a tap row) against the fp64 oracle."""
  rng = np.random.default_rng(seed)
  cases = 20 if DEV == 'cuda' else 3
  for _ in range(cases):
    fs = int(rng.choice([64, 64, 64, 128, 192, 256]))
    f = int(rng.integers(1, 260 if DEV == 'cuda' else 70))
    n = f * fs - int(rng.integers(0, fs))
    if n < 1:
      continue
    b = int(rng.integers(1, 5))
    mags = (rng.standard_normal((b, f, 65)) + 3.0).astype(np.float32)
    synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, seed=5)
    if rng.integers(0, 2):
      noise = rng.uniform(-1, 1, (b, n)).astype(np.float32)
      got = npy(synth(mags, noise=noise))
    else:
      noise = O.device_uniform_noise(b, n, seed=5)
      got = npy(synth(mags))
    ref = O.filtered_noise(mags, noise, 0, O.exp_sigmoid, dtype=np.float64)
    np.testing.assert_allclose(got, ref, rtol=0, atol=noise_tol(ref), err_msg=str(dict(fs=fs, frames=f, n=n, batch=b)))


@pytest.mark.parametrize('m,ws,n_frames,n,batch', [
    (100, 0, 250, 16000, 3),       # synths_test.py:43-50's 100 magnitudes: a 198-tap filter
    (100, 257, 250, 16000, 2),     # ... under the constructor's default window: 257 is beyond 198 taps, the full window
    (256, 0, 120, 7680, 2),        # 510 taps
    (256, 257, 120, 7680, 2),      # ... cropped to 257 (core.py:1520-1527)
    (65, 0, 160, 16000, 3),        # frames of 100 samples: pieces of 64 and 36
    (40, 0, 90, 17280 - 50, 2),    # frames of 192 with a ragged end
    (12, 0, 300, 6000, 2),         # frames of 20 samples
    (129, 65, 64, 32768, 2),       # frames of 512
    (256, 0, 65, 324, 3),          # 510 taps on frames of 5 samples: 116 tap rows per 64 outputs - no tiled kernel holds them; the
    (129, 257, 10, 50, 4),         # plain sum of csrc/general.hip takes over (tools/fuzz_parity.py found these returning UNSUPPORTED)
])
def test_filtered_noise_general_shapes_on_the_matrix_cores(ddsp, m, ws, n_frames, n, batch):
  """filtered_noise_general.hip (noise_ir_gemm_kernel, tv_fir_mfma_kernel): any band count, window and frame size against
  the fp64 oracle with supplied noise; generated noise against the same call with the generator's samples handed in;
  bit equality run to run and of a row alone with the row inside the batch (the LDS adds never see three terms)."""
  rng = np.random.default_rng(m * 7 + n_frames)
  mags = (rng.standard_normal((batch, n_frames, m)) + 2.0).astype(np.float32)
  noise = rng.uniform(-1, 1, (batch, n)).astype(np.float32)
  synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=ws, seed=11)
  got = npy(synth(mags, noise=noise))
  ref = O.filtered_noise(mags, noise, ws, O.exp_sigmoid, dtype=np.float64)
  assert got.shape == ref.shape
  np.testing.assert_allclose(got, ref, rtol=0, atol=noise_tol(ref))
  np.testing.assert_array_equal(npy(synth(mags, noise=noise)), got)
  np.testing.assert_array_equal(npy(synth(mags[1:2], noise=noise[1:2])), got[1:2])
  # the controls beside the audio (the IR kernel writes them on the way in)
  full = synth(mags, noise=noise, return_outputs_dict=True)
  np.testing.assert_array_equal(npy(full['signal']), got)
  np.testing.assert_allclose(npy(full['controls']['magnitudes']), O.exp_sigmoid(mags.astype(np.float64) - 5.0, dtype=np.float64),
                             rtol=2e-5, atol=1e-9)
  # generated noise: the stream is this library's contract (oracle.device_uniform_noise restates it)
  gen = ddsp.synths.FilteredNoise(n_samples=n, window_size=ws, seed=11)
  z = O.device_uniform_noise(batch, n, seed=11)
  got_gen = npy(gen(mags))
  ref_gen = O.filtered_noise(mags, z, ws, O.exp_sigmoid, dtype=np.float64)
  np.testing.assert_allclose(got_gen, ref_gen, rtol=0, atol=noise_tol(ref_gen))


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_filtered_noise_general_random_shapes(ddsp, seed):
  """Random band counts (3 .. 140), window sizes (full, cropped, odd), frame sizes (5 .. 400), ragged lengths, batch sizes: the
  tile cuts of tv_fir_mfma_kernel (pieces, runs, history, the fused design for <= 128 bands, two launches above) against the
  fp64 oracle."""
  rng = np.random.default_rng(seed)
  cases = 12 if DEV == 'cuda' else 3
  for _ in range(cases):
    m = int(rng.choice([3, 5, 17, 33, 64, 66, 100, 129, 140]))
    l0 = 2 * (m - 1)
    ws = int(rng.choice([0, l0 + 5, max(3, l0 // 2 + 1), max(3, (l0 // 3) | 1)]))
    fs = int(rng.choice([5, 16, 37, 64, 100, 128, 400]))
    f = int(rng.integers(1, 60 if DEV == 'cuda' else 12))
    n = f * fs - int(rng.integers(0, fs))
    if n < 1:
      continue
    b = int(rng.integers(1, 4))
    mags = (rng.standard_normal((b, f, m)) + 2.0).astype(np.float32)
    noise = rng.uniform(-1, 1, (b, n)).astype(np.float32)
    synth = ddsp.synths.FilteredNoise(n_samples=n, window_size=ws)
    got = npy(synth(mags, noise=noise))
    ref = O.filtered_noise(mags, noise, ws, O.exp_sigmoid, dtype=np.float64)
    assert got.shape == ref.shape, dict(m=m, ws=ws, fs=fs, frames=f, n=n, batch=b)
    np.testing.assert_allclose(got, ref, rtol=0, atol=noise_tol(ref), err_msg=str(dict(m=m, ws=ws, fs=fs, frames=f, n=n, batch=b)))


def test_fft_convolve_general_shapes_on_the_matrix_cores(ddsp):
  """core.fft_convolve (ddsp/core.py:1382-1473) through tv_fir_mfma_kernel: odd tap counts, one filter for the whole batch,
  explicit delay compensation, a single frame (a time-invariant filter), frames shorter than 16 samples."""
  rng = np.random.default_rng(77)
  for b, bir, f, l, n, dc in [(3, 3, 40, 97, 5000, -1), (4, 1, 25, 200, 6400, -1), (2, 2, 1, 63, 3000, 0),
                              (2, 2, 500, 33, 4000, 5), (1, 1, 7, 3, 700, -1), (2, 2, 10, 400, 2560, -1)]:
    audio = rng.standard_normal((b, n)).astype(np.float32)
    ir = (rng.standard_normal((bir, f, l)) / np.sqrt(l)).astype(np.float32)
    got = npy(ddsp.core.fft_convolve(audio, ir, delay_compensation=dc))
    ref = O.fft_convolve(audio, ir, delay_compensation=dc, dtype=np.float64)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 + 1e-5 * np.abs(ref).max(), err_msg=str((b, bir, f, l, n, dc)))
    np.testing.assert_array_equal(npy(ddsp.core.fft_convolve(audio, ir, delay_compensation=dc)), got)


# ---- processors.Add fused into the Harmonic kernel (ddsp/processors.py:162-176; gin/models/ae.gin:49-56) --------------
def test_processor_group_fused_add_is_bit_identical(ddsp):
  """ProcessorGroup[Harmonic, FilteredNoise, Add] asked for the signal only runs Harmonic with the Add fused in
  (ddsp_harmonic_add_f32); asked for the outputs dict it runs the three processors.  Same samples, bit for bit; with a
  Reverb behind the Add too; and Harmonic.call_add falls back to the two calls where the wavetable kernel does not apply."""
  n_frames, n = 100, 6400
  x = canonical_inputs(3, seed=19, n_frames=n_frames)
  features = {'amps': x['amplitudes'], 'harmonic_distribution': x['harmonic_distribution'],
              'f0_hz': x['f0_hz'], 'magnitudes': x['magnitudes']}

  def group(extra=()):
    harmonic = ddsp.synths.Harmonic(n_samples=n, name='harmonic')
    noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, name='filtered_noise', seed=3)
    add = ddsp.processors.Add(name='add')
    dag = [(harmonic, ['amps', 'harmonic_distribution', 'f0_hz']), (noise, ['magnitudes']),
           (add, ['filtered_noise/signal', 'harmonic/signal'])] + list(extra)
    return ddsp.processors.ProcessorGroup(dag=dag, name='processor_group')
  g1, g2 = group(), group()
  assert g1._fused_add_plan() == (0, 2)
  with torch.no_grad():
    fused = npy(g1(features))
  full = g2(features, return_outputs_dict=True)
  np.testing.assert_array_equal(fused, npy(full['signal']))
  np.testing.assert_array_equal(fused, npy(full['controls']['harmonic']['signal']) + npy(full['controls']['filtered_noise']['signal']))
  # a node behind the Add
  rev = ddsp.effects.Reverb(trainable=True, reverb_length=500, name='reverb')
  rev.build(device=torch.device(DEV))
  rev._ir = ddsp.core.tf_float32(np.random.default_rng(2).standard_normal(500) * 0.05)
  g3, g4 = group([(rev, ['add/signal'])]), group([(rev, ['add/signal'])])
  with torch.no_grad():
    np.testing.assert_array_equal(npy(g3(features)), npy(g4(features, return_outputs_dict=True)['signal']))
  # something else reads the harmonic signal: no fusion
  crop = ddsp.processors.Crop(frame_size=64, name='crop')
  g5 = group([(crop, ['harmonic/signal'])])
  assert g5._fused_add_plan() is None
  # 99 harmonics (rows that are not 16 bytes apart: the wavetable kernel's 4-byte row loads), fused as well
  h = ddsp.synths.Harmonic(n_samples=n)
  hd99 = x['harmonic_distribution'][..., :99]
  z = npy(full['controls']['filtered_noise']['signal'])
  np.testing.assert_array_equal(npy(h.call_add(x['amplitudes'], hd99, x['f0_hz'], z)),
                                npy(h(x['amplitudes'], hd99, x['f0_hz'])) + z)
  # call_add on a shape the wavetable kernel does not take (a hop of 96 samples): the two calls
  n96 = n_frames * 96
  h96 = ddsp.synths.Harmonic(n_samples=n96)
  z96 = np.random.default_rng(4).standard_normal((3, n96)).astype(np.float32)
  np.testing.assert_array_equal(npy(h96.call_add(x['amplitudes'], x['harmonic_distribution'], x['f0_hz'], z96)),
                                npy(h96(x['amplitudes'], x['harmonic_distribution'], x['f0_hz'])) + z96)


def test_processor_group_fused_add_is_differentiable(ddsp):
  """Round 4: with inputs that require grad the ae.gin group still runs Harmonic + Add as ONE launch (one autograd node,
  _HarmonicAddFunction): same samples and the same gradients, bit for bit, as the three processors run one by one."""
  n_frames, n = 60, 3840
  x = canonical_inputs(2, seed=29, n_frames=n_frames)

  def build():
    harmonic = ddsp.synths.Harmonic(n_samples=n, name='harmonic')
    noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, name='filtered_noise', seed=5)
    add = ddsp.processors.Add(name='add')
    dag = [(harmonic, ['amps', 'harmonic_distribution', 'f0_hz']), (noise, ['magnitudes']),
           (add, ['harmonic/signal', 'filtered_noise/signal'])]
    return ddsp.processors.ProcessorGroup(dag=dag, name='processor_group')

  def features():
    t = ddsp.core.tf_float32
    return {'amps': t(x['amplitudes']).requires_grad_(True), 'harmonic_distribution': t(x['harmonic_distribution']).requires_grad_(True),
            'f0_hz': t(x['f0_hz']), 'magnitudes': t(x['magnitudes']).requires_grad_(True)}
  w = ddsp.core.tf_float32(np.random.default_rng(30).standard_normal((2, n)))
  f1, f2 = features(), features()
  g1, g2 = build(), build()
  out = g1(f1)                                        # the fused plan
  assert out.requires_grad and type(out.grad_fn).__name__.startswith('_HarmonicAddFunction')
  ref = g2(f2, return_outputs_dict=True)['signal']    # every processor on its own
  np.testing.assert_array_equal(npy(out), npy(ref))
  out.backward(w)
  ref.backward(w)
  for key in ('amps', 'harmonic_distribution', 'magnitudes'):
    assert f1[key].grad is not None
    np.testing.assert_array_equal(npy(f1[key].grad), npy(f2[key].grad), err_msg=key)


def test_processor_group_fused_add_with_a_trainable_module_inside(ddsp):
  """ADVICE r3 (processors.py:98): the gradient may come from a module INSIDE the DAG while the DAG's inputs are detached - a
  trainable Reverb ahead of the Add (the Add then gets a signal that requires grad), or a re-bound / subclassed group.  The
  short-cut through Harmonic.call_add must then take the two differentiable calls, not raise: same samples as the unfused
  group, and dL/d ir reaches the Reverb."""
  n_frames, n = 50, 3200
  x = canonical_inputs(2, seed=23, n_frames=n_frames)
  features = {'amps': x['amplitudes'], 'harmonic_distribution': x['harmonic_distribution'], 'f0_hz': x['f0_hz'],
              'magnitudes': x['magnitudes']}

  def build():
    harmonic = ddsp.synths.Harmonic(n_samples=n, name='harmonic')
    noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, name='filtered_noise', seed=3)
    rev = ddsp.effects.Reverb(trainable=True, reverb_length=300, name='reverb')
    rev.build(device=torch.device(DEV))
    rev._ir = ddsp.core.tf_float32(np.random.default_rng(5).standard_normal(300) * 0.05).requires_grad_(True)
    add = ddsp.processors.Add(name='add')
    dag = [(noise, ['magnitudes']), (rev, ['filtered_noise/signal']),
           (harmonic, ['amps', 'harmonic_distribution', 'f0_hz']), (add, ['reverb/signal', 'harmonic/signal'])]
    return ddsp.processors.ProcessorGroup(dag=dag, name='processor_group'), rev
  g1, rev1 = build()
  g2, rev2 = build()
  assert g1._fused_add_plan() == (2, 3)
  out = g1(features)                                   # inputs detached, the Reverb's impulse response requires grad
  assert out.requires_grad
  ref = g2(features, return_outputs_dict=True)['signal']
  np.testing.assert_array_equal(npy(out), npy(ref))
  w = ddsp.core.tf_float32(np.random.default_rng(6).standard_normal((2, n)))
  out.backward(w)
  ref.backward(w)
  assert rev1._ir.grad is not None and float(rev1._ir.grad.abs().max()) > 0
  np.testing.assert_array_equal(npy(rev1._ir.grad), npy(rev2._ir.grad))
  # a module re-bound after the plan was made: worked out again, not served from the cache
  g1.harmonic = ddsp.synths.Harmonic(n_samples=n, name='harmonic')
  assert g1._fused_add_plan() == (2, 3)
  # a subclass that overrides get_signal is never short-cut
  class Halved(ddsp.processors.ProcessorGroup):
    def get_signal(self, outputs):
      return ddsp.processors.ProcessorGroup.get_signal(self, outputs) * 0.5
  def dag():                                           # (modules of its own per group: a FilteredNoise counts its calls)
    harmonic = ddsp.synths.Harmonic(n_samples=n, name='harmonic')
    noise = ddsp.synths.FilteredNoise(n_samples=n, window_size=0, name='filtered_noise', seed=3)
    return [(harmonic, ['amps', 'harmonic_distribution', 'f0_hz']), (noise, ['magnitudes']),
            (ddsp.processors.Add(name='add'), ['filtered_noise/signal', 'harmonic/signal'])]
  with torch.no_grad():
    halved = npy(Halved(dag=dag())(features))
    plain = npy(ddsp.processors.ProcessorGroup(dag=dag())(features, return_outputs_dict=True)['signal'])
  np.testing.assert_array_equal(halved, plain * 0.5)


@pytest.mark.parametrize('k,hop', [(100, 64), (128, 64), (60, 128), (99, 64)])
def test_harmonic_table_sizes_follow_f0(ddsp, k, hop):
  """Round 4 (csrc/harmonic_table.hip, "table size"): a segment of 62 frames is tabulated on 512, 256, 128 or 64 points by the
  number of harmonics its smallest f0 leaves below Nyquist.  A clip whose f0 visits every class - steps between notes, a glide
  from 90 Hz to 1.9 kHz, a segment with one frame of silence (f0 = 0: every harmonic "below Nyquist", 512 points) - against exact
  arithmetic at the wavetable kernel's tolerance; and the bits of a row do not depend on how the batch is cut into blocks and
  chunks (the sizes are a function of the row's own segments): rows alone, as sub-batches, in a batch."""
  f, sr = 62 * 5 + 17, 16000
  n = f * hop
  rng = np.random.default_rng(700 + k + hop)
  b = 5
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = np.zeros((b, f, 1), np.float32)
  notes = [110.0, 170.0, 330.0, 700.0, 1400.0, 240.0]
  f0[0, :, 0] = np.repeat(notes, 62)[:f] + rng.standard_normal(f)                   # a note per segment: every table size
  f0[1, :, 0] = 90.0 * (1900.0 / 90.0) ** (np.arange(f) / (f - 1.0))                 # a glide through all of them
  f0[2, :, 0] = 420.0 + 6.0 * np.sin(np.arange(f) * 0.14)                            # vibrato inside one class
  f0[3, :, 0] = np.where(np.arange(f) % 97 == 40, 0.0, 650.0 + rng.standard_normal(f))     # frames of silence in a high note
  f0[4, :, 0] = np.repeat([158.0, 155.0, 158.0, 155.0, 158.0, 155.0], 62)[:f] + 0.5 * rng.standard_normal(f)   # either side of a threshold
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr)
  full = npy(synth(amps, hd, f0))
  assert full.shape == (b, n) and np.isfinite(full).all()
  exact, knife, exact32 = _harmonic_exact(amps, hd, f0, n, sr, 'window', with_knife_edges='fp32 mask')
  scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
  assert knife.mean() <= 1e-2
  for r in range(b):
    parity_check(np.where(knife[r], exact[r], full[r]), exact[r], HARM_TABLE_ATOL * scale, 'table sizes by f0, row %d, K = %d' % (r, k))
  assert_knife_edges_take_the_fp32_side(full, exact32, knife, HARM_TABLE_ATOL * scale, ('table sizes by f0', k))
  for r in range(b):
    np.testing.assert_array_equal(npy(synth(amps[r:r + 1], hd[r:r + 1], f0[r:r + 1])), full[r:r + 1], err_msg='row %d alone' % r)
  np.testing.assert_array_equal(npy(synth(amps[1:4], hd[1:4], f0[1:4])), full[1:4])
  np.testing.assert_array_equal(npy(synth(amps[3:], hd[3:], f0[3:])), full[3:])


@pytest.mark.parametrize('hop,k', [(100, 100), (96, 60), (40, 30), (200, 128), (150, 160)])
def test_harmonic_frame_sizes_that_are_not_multiples_of_64(ddsp, hop, k):
  """Round 4: frames that are not whole tiles of 64 samples - 640 frames of 100 samples is the reference's
  own test shape (ddsp/synths_test.py, processors_test.py:36-41) - run on the wavetable kernel with the frame's last tile cut
  short (7 x the canonical time on the one-thread-per-sample kernels until round 3).  Against exact arithmetic; rows alone and in
  a batch bit-equal; Harmonic with processors.Add fused in equal to the two calls."""
  f, sr = 77, 16000
  n = f * hop
  rng = np.random.default_rng(900 + hop)
  b = 3
  amps = rng.standard_normal((b, f, 1)).astype(np.float32)
  hd = rng.standard_normal((b, f, k)).astype(np.float32)
  f0 = np.abs(40.0 + 30.0 * rng.standard_normal((b, f, 1))).astype(np.float32)
  for method in ('window', 'linear'):
    synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=sr, amp_resample_method=method)
    full = npy(synth(amps, hd, f0))
    assert full.shape == (b, n) and np.isfinite(full).all()
    exact, knife, exact32 = _harmonic_exact(amps, hd, f0, n, sr, method, with_knife_edges='fp32 mask')
    scale = max(1.0, float(O.exp_sigmoid(amps.astype(np.float64), dtype=np.float64).max()))
    parity_check(np.where(knife, exact, full), exact, HARM_TABLE_ATOL * scale, 'hop %d, K = %d, %s' % (hop, k, method))
    assert_knife_edges_take_the_fp32_side(full, exact32, knife, HARM_TABLE_ATOL * scale, (hop, k, method))
    np.testing.assert_array_equal(npy(synth(amps[1:2], hd[1:2], f0[1:2])), full[1:2])
    z = rng.standard_normal((b, n)).astype(np.float32)
    np.testing.assert_array_equal(npy(synth.call_add(amps, hd, f0, z)), full + z)


def test_frame_rate_nyquist_mask_takes_the_reference_fp32_side(ddsp, harm_kernel):
  """f0 = fp32(8000 / 75) = 106.666664 Hz at 16 kHz: harmonic 75 is 7999.9998 Hz in exact arithmetic - below Nyquist - and 8000.0
  in the fp32 product the reference forms (get_harmonic_frequencies, remove_above_nyquist: core.py:1028-1045, 869-891): masked,
  and normalize_harmonics (core.py:894-907) then divides the distribution by the sum of the OTHER harmonics.  The kernels take
  the reference's side, for the whole clip (tools/fuzz_parity.py met this as a 'failure' of its fp64 checker: an f0 sweep whose
  steps land on 8000 / k).  With equal weight on harmonics 10 and 75: harmonic 10 at amplitude 1, where exact arithmetic keeps
  both alive and gives it 1/2."""
  f0_value = np.float32(8000.0 / 75.0)
  assert np.float32(f0_value * np.float32(75.0)) == np.float32(8000.0) and float(f0_value) * 75.0 < 8000.0
  b, f, k, hop = 1, 8, 100, 64
  n = f * hop
  amps = np.ones((b, f, 1), np.float32)
  hd = np.zeros((b, f, k), np.float32); hd[:, :, 9] = 1.0; hd[:, :, 74] = 1.0
  f0 = np.full((b, f, 1), f0_value, np.float32)
  synth = ddsp.synths.Harmonic(n_samples=n, sample_rate=16000, scale_fn=None)
  full = synth(amps, hd, f0, return_outputs_dict=True)
  out = npy(full['signal'])
  t = np.arange(1, n + 1) / 16000.0
  fp32_side = np.sin(2 * np.pi * 10 * float(f0_value) * t)[None]           # harmonic 10 alone, the whole amplitude
  np.testing.assert_allclose(out, fp32_side, rtol=0, atol=HARM_TRUTH_ATOL)
  ctl = npy(full['controls']['harmonic_distribution'])
  assert np.all(ctl[..., 9] == 1.0) and np.all(ctl[..., 74] == 0.0)
  faithful = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=16000, scale_fn=None, dtype=np.float32)
  np.testing.assert_allclose(out, faithful, rtol=0, atol=HARM_FAITHFUL_ATOL)      # the reference's own fp32 chain agrees
  exact = O.harmonic(amps, hd, f0, n_samples=n, sample_rate=16000, scale_fn=None, dtype=np.float64)
  assert np.abs(out - exact).max() > 0.4                      # ... fp64 hears harmonic 10 at half the amplitude
