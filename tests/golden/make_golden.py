"""Generate tests/golden/*.npz by running the REFERENCE'S OWN source files.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

`tf_numpy_shim.install()` makes `ddsp/core.py`, `ddsp/synths.py` and
`ddsp/processors.py` importable, unmodified, on top of a numpy stand-in for the
TensorFlow ops they call.  Every array stored below is therefore produced by the
reference's code (index math, crops, windows, masks, op order) with TF-op semantics
restated per SURVEY.md Appendix A; it is NOT an output of real TensorFlow.
The fixtures are small (seconds of CPU) and committed; the GPU box never runs this.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

# DDSP_GOLDEN_BACKEND=tf (set by make_golden_tf.py): the same cases on REAL TensorFlow, written beside the fixtures as NAME.tf.npz
BACKEND = os.environ.get('DDSP_GOLDEN_BACKEND', 'shim')
REFERENCE = os.environ.get('DDSP_REFERENCE_ROOT', '/root/reference')
SUFFIX = '.npz' if BACKEND == 'shim' else '.tf.npz'
if BACKEND == 'shim':
  import tf_numpy_shim  # noqa: E402
  tf_numpy_shim.install(REFERENCE)
else:
  import make_golden_tf  # noqa: E402
  make_golden_tf.install(REFERENCE)
from ddsp import core, effects, losses, processors, spectral_ops, synths  # noqa: E402  (the reference's files)


def a(x):
  return np.ascontiguousarray(np.asarray(x))


def harmonic_case(seed, batch, n_frames, n_harmonics, n_samples, sample_rate, f0_lo, f0_hi,
                  amp_method, angular, scale=True, normalize=True):
  rng = np.random.default_rng(seed)
  amps = rng.standard_normal((batch, n_frames, 1)).astype(np.float32)
  hd = rng.standard_normal((batch, n_frames, n_harmonics)).astype(np.float32)
  f0 = rng.uniform(f0_lo, f0_hi, (batch, n_frames, 1)).astype(np.float32)
  if not scale:  # controls must then already be positive
    amps, hd = np.abs(amps) + 0.1, np.abs(hd) + 0.01
  synth = synths.Harmonic(n_samples=n_samples, sample_rate=sample_rate,
                          scale_fn=core.exp_sigmoid if scale else None,
                          normalize_below_nyquist=normalize,
                          amp_resample_method=amp_method, use_angular_cumsum=angular)
  out = synth(amps, hd, f0, return_outputs_dict=True)
  return dict(
      amplitudes=amps, harmonic_distribution=hd, f0_hz=f0,
      n_samples=n_samples, sample_rate=sample_rate, amp_method=amp_method,
      angular=int(angular), scale=int(scale), normalize=int(normalize),
      ctl_amplitudes=a(out['controls']['amplitudes']),
      ctl_harmonic_distribution=a(out['controls']['harmonic_distribution']),
      signal=a(out['signal']))


def noise_case(seed, batch, n_frames, n_bands, n_samples, window_size, scale=True):
  rng = np.random.default_rng(seed)
  mags = rng.standard_normal((batch, n_frames, n_bands)).astype(np.float32)
  if not scale:
    mags = np.abs(mags)
  noise = rng.uniform(-1.0, 1.0, (batch, n_samples)).astype(np.float32)
  synth = synths.FilteredNoise(n_samples=n_samples, window_size=window_size,
                               scale_fn=core.exp_sigmoid if scale else None)
  controls = synth.get_controls(mags)
  # get_signal draws its own tf.random.uniform; inject the noise through the same
  # core.frequency_filter call get_signal makes (synths.py:192-196).
  ir = core.frequency_impulse_response(controls['magnitudes'], window_size=window_size)
  signal = core.frequency_filter(noise, controls['magnitudes'], window_size=window_size)
  return dict(magnitudes=mags, noise=noise, n_samples=n_samples, window_size=window_size,
              scale=int(scale), ctl_magnitudes=a(controls['magnitudes']),
              impulse_response=a(ir), signal=a(signal))


def reverb_case(seed, batch, n_samples, ir_size, ir_batch, add_dry, trainable=False, ir_rank=2):
  rng = np.random.default_rng(seed)
  audio = rng.standard_normal((batch, n_samples)).astype(np.float32)
  decay = np.exp(-np.arange(ir_size) / (0.25 * ir_size))
  ir = (rng.standard_normal((ir_batch, ir_size)) * decay).astype(np.float32)
  rev = effects.Reverb(trainable=trainable, reverb_length=ir_size, add_dry=add_dry)
  if trainable:                     # the variable tf would create in build() (effects.py:71-80)
    rev._ir = ir[0]
    rev.built = True
    signal = rev(audio)
    ir_in = ir[0]
  else:
    ir_in = ir[:, :, None] if ir_rank == 3 else ir
    signal = rev(audio, ir_in)
  return dict(audio=audio, ir=ir_in, add_dry=int(add_dry), trainable=int(trainable), signal=a(signal))


def main():
  cases = {}
  # --- Harmonic: hop 64 (the canonical hop), some harmonics crossing Nyquist ---
  cases['harmonic_window_cumsum'] = harmonic_case(
      1, 2, 25, 12, 1600, 16000, 300.0, 900.0, 'window', False)
  cases['harmonic_window_angular'] = harmonic_case(
      2, 2, 25, 12, 1600, 16000, 300.0, 900.0, 'window', True)
  cases['harmonic_linear_cumsum'] = harmonic_case(
      3, 2, 25, 12, 1600, 16000, 100.0, 400.0, 'linear', False)
  # all 100 harmonics live (the headline regime), short clip
  cases['harmonic_k100_live'] = harmonic_case(
      4, 1, 20, 100, 1280, 16000, 69.0, 71.0, 'window', False)
  # hop 192 (48 kHz / 250 Hz): non power-of-two hop, K=40, angular (vst_48k.gin)
  cases['harmonic_hop192_angular'] = harmonic_case(
      5, 1, 10, 40, 1920, 48000, 200.0, 700.0, 'linear', True)
  # scale_fn=None and normalize_below_nyquist=False (synths_test.py:26-30 style)
  cases['harmonic_noscale_nonorm'] = harmonic_case(
      6, 2, 16, 20, 1024, 16000, 350.0, 600.0, 'window', False, scale=False,
      normalize=False)
  # --- FilteredNoise ---
  cases['noise_m65_w257'] = noise_case(11, 2, 25, 65, 1600, 257)      # ae.gin shape
  cases['noise_m65_w0'] = noise_case(12, 2, 25, 65, 1600, 0)
  cases['noise_m33_w17'] = noise_case(13, 2, 10, 33, 640, 17)         # cropped IR branch
  cases['noise_m17_w16_even'] = noise_case(14, 1, 8, 17, 512, 16)     # even window -> 15 taps
  cases['noise_ragged'] = noise_case(15, 2, 7, 9, 100, 0, scale=False)  # N % F != 0 (pad_end)

  # --- effects.Reverb: ir from the network [B, L(,1)], and a trainable Reverb's single IR ---
  cases['reverb_b2_dry'] = reverb_case(31, 2, 1000, 300, 2, True)
  cases['reverb_b2_wet_rank3'] = reverb_case(32, 2, 777, 1200, 2, False, ir_rank=3)   # IR longer than the audio
  cases['reverb_trainable'] = reverb_case(33, 3, 640, 200, 1, True, trainable=True)

  # --- effects.ExpDecayReverb: network gain / decay [B, 1]; trainable (gain 2.0, decay 4.0) ---
  def exp_decay_case(seed, batch, n_samples, ir_size, add_dry, trainable):
    r = np.random.default_rng(seed)
    audio = r.standard_normal((batch, n_samples)).astype(np.float32)
    rev = effects.ExpDecayReverb(trainable=trainable, reverb_length=ir_size, add_dry=add_dry)
    # the burst _get_ir is about to draw: the stand-in's tf.random.uniform is a fixed stream per call
    noise = a(tf_numpy_shim.build_tf_module()[0].random.uniform([1, ir_size], minval=-1.0, maxval=1.0))
    if trainable:                   # the weights tf would create in build() (effects.py:153-168)
      rev._gain = np.full((1,), 2.0, np.float32)
      rev._decay = np.full((1,), 4.0, np.float32)
      rev.built = True
      out = rev(audio, return_outputs_dict=True)
      d = dict(gain=rev._gain, decay=rev._decay)
    else:
      gain = r.standard_normal((batch, 1)).astype(np.float32)
      decay = r.uniform(-1.0, 3.0, (batch, 1)).astype(np.float32)
      out = rev(audio, gain, decay, return_outputs_dict=True)
      d = dict(gain=gain, decay=decay)
    d.update(audio=audio, noise=noise, add_dry=int(add_dry), trainable=int(trainable),
             ir=a(out['controls']['ir']), signal=a(out['signal']))
    return d
  cases['exp_decay_reverb_b3'] = exp_decay_case(34, 3, 800, 250, True, False)
  cases['exp_decay_reverb_trainable'] = exp_decay_case(35, 2, 640, 1500, False, True)   # IR longer than the audio

  # --- losses.SpectralLoss (mag + logmag terms, L1), and the magnitudes of one scale ---
  rng = np.random.default_rng(41)
  tgt = (rng.standard_normal((2, 3000)) * 0.3).astype(np.float32)
  aud = (tgt + 0.05 * rng.standard_normal((2, 3000))).astype(np.float32)
  aud[1, 1500:] = 0.0                                   # exact zeros: the safe_log branch
  cases['spectral_loss'] = dict(
      target_audio=tgt, audio=aud,
      loss_default=a(losses.SpectralLoss()(tgt, aud)),
      loss_ae_gin=a(losses.SpectralLoss(mag_weight=1.0, logmag_weight=1.0)(tgt, aud)),      # ae.gin:36-41
      loss_two_scales=a(losses.SpectralLoss(fft_sizes=(512, 64), logmag_weight=0.5)(tgt, aud)),
      mag_256=a(spectral_ops.compute_mag(aud, size=256)))
  # (round 2) the rest of SpectralLoss's argument space: every term, 'L2' / 'COSINE', the weights mask
  wmask = np.array([[[1.0]], [[0.25]]], np.float32)
  all_terms = dict(mag_weight=1.0, delta_time_weight=0.5, delta_freq_weight=0.25, cumsum_freq_weight=0.125, logmag_weight=0.75)
  cases['spectral_loss_terms'] = dict(
      target_audio=tgt, audio=aud, weights=wmask, fft_sizes=np.array([512, 128, 64]),
      **{k: np.float32(v) for k, v in all_terms.items()},
      l1=a(losses.SpectralLoss(fft_sizes=(512, 128, 64), **all_terms)(tgt, aud)),
      l2=a(losses.SpectralLoss(fft_sizes=(512, 128, 64), loss_type='L2', **all_terms)(tgt, aud)),
      cosine=a(losses.SpectralLoss(fft_sizes=(512, 128, 64), loss_type='COSINE', **all_terms)(tgt, aud)),
      l1_weighted=a(losses.SpectralLoss(fft_sizes=(512, 128, 64), **all_terms)(tgt, aud, weights=wmask)),
      l2_weighted=a(losses.SpectralLoss(fft_sizes=(512, 128, 64), loss_type='L2', **all_terms)(tgt, aud, weights=wmask)),
      cosine_weighted=a(losses.SpectralLoss(fft_sizes=(512, 128, 64), loss_type='COSINE', **all_terms)(tgt, aud, weights=wmask)),
      l2_delta_time_only=a(losses.SpectralLoss(fft_sizes=(256,), loss_type='L2', mag_weight=0.0, delta_time_weight=1.0)(tgt, aud)),
      l1_cumsum_only=a(losses.SpectralLoss(fft_sizes=(2048, 64), mag_weight=0.0, cumsum_freq_weight=1.0)(tgt, aud)))

  # (round 5) the loudness term (losses.py:238-242 -> spectral_ops.compute_loudness with n_fft = 2048; librosa's two functions
  # restated in the stand-in): alone, with the other terms, every loss type; the loudness curve itself
  cases['spectral_loss_loudness'] = dict(
      target_audio=tgt, audio=aud,
      loudness_audio=a(spectral_ops.compute_loudness(aud, n_fft=2048, use_tf=True)).astype(np.float32),
      l1_only=a(losses.SpectralLoss(mag_weight=0.0, loudness_weight=1.0)(tgt, aud)),
      l2_only=a(losses.SpectralLoss(loss_type='L2', mag_weight=0.0, loudness_weight=1.0)(tgt, aud)),
      cosine_only=a(losses.SpectralLoss(loss_type='COSINE', mag_weight=0.0, loudness_weight=1.0)(tgt, aud)),
      l1_all=a(losses.SpectralLoss(loudness_weight=0.5, **all_terms)(tgt, aud)))

  # --- core.streaming_harmonic_synthesis: the VST model's call (2 frames -> one hop, carried phase) ---
  rng = np.random.default_rng(51)
  def streaming_case(batch, n_frames, n_harm, n_samples, sr, method, with_hd=True, f_lo=200.0, f_hi=900.0):
    f0 = rng.uniform(f_lo, f_hi, (batch, n_frames, 1)).astype(np.float32)
    amps = rng.uniform(0.1, 1.0, (batch, n_frames, 1)).astype(np.float32)
    hd = rng.uniform(0.0, 1.0, (batch, n_frames, n_harm)).astype(np.float32) if with_hd else None
    phase0 = rng.uniform(0.0, 6.0, (batch, 1, 1)).astype(np.float32)
    audio, final_phase = core.streaming_harmonic_synthesis(
        frequencies=f0, amplitudes=amps, harmonic_distribution=hd, initial_phase=phase0,
        n_samples=n_samples, sample_rate=sr, amp_resample_method=method)
    d = dict(f0_hz=f0, amplitudes=amps, initial_phase=phase0, n_samples=n_samples, sample_rate=sr,
             amp_method=method, audio=a(audio), final_phase=a(final_phase))
    if with_hd:
      d['harmonic_distribution'] = hd
    return d
  cases['streaming_2frames_linear'] = streaming_case(1, 2, 60, 64, 16000, 'linear')
  cases['streaming_2frames_window'] = streaming_case(2, 2, 100, 320, 16000, 'window', f_lo=60.0, f_hi=90.0)
  cases['streaming_nyquist_crossing'] = streaming_case(2, 4, 30, 256, 16000, 'linear', f_lo=250.0, f_hi=600.0)
  cases['streaming_no_distribution'] = streaming_case(3, 5, 1, 640, 48000, 'linear', with_hd=False)
  # (round 2) the envelopes the closed-form kernel does not take: the reference's chain on audio-rate envelopes
  cases['streaming_cubic'] = streaming_case(2, 6, 20, 384, 16000, 'cubic')
  cases['streaming_nearest_ragged'] = streaming_case(1, 3, 10, 100, 16000, 'nearest')
  cases['streaming_linear_ragged'] = streaming_case(2, 7, 16, 450, 16000, 'linear')

  # --- core.harmonic_oscillator_bank on audio-rate inputs (core.py:966-1025), both phase accumulations ---
  hf = rng.uniform(80.0, 700.0, (2, 2300, 1)).astype(np.float32)
  ha = rng.uniform(0.0, 1.0, (2, 2300, 12)).astype(np.float32)
  hp = rng.uniform(0.0, 6.0, (2, 1, 1)).astype(np.float32)
  hob_a, hob_pa = core.harmonic_oscillator_bank(hf, ha, hp, sample_rate=16000, use_angular_cumsum=True)
  hob_c, hob_pc = core.harmonic_oscillator_bank(hf, ha, None, sample_rate=16000, use_angular_cumsum=False)
  cases['harmonic_oscillator_bank'] = dict(
      frequency=hf, amplitude_envelopes=ha, initial_phase=hp, sample_rate=16000,
      audio_angular=a(hob_a), final_phase_angular=a(hob_pa), audio_cumsum=a(hob_c), final_phase_cumsum=a(hob_pc))

  # --- resampling pieces on their own ---
  rng = np.random.default_rng(21)
  x = rng.standard_normal((2, 9, 3)).astype(np.float32)
  cases['resample'] = dict(
      x=x,
      window_576=a(core.resample(x, 576, method='window')),
      linear_576=a(core.resample(x, 576, method='linear')),
      linear_1728=a(core.resample(x, 1728, method='linear')),   # hop 192
      linear_900=a(core.resample(x, 900, method='linear')))     # hop 100
  # every method x add_endpoint, up- and down-sampling (core_test.py:219-293), and a 4-D input
  xs = (1.0 - np.sin(np.linspace(0, np.pi, 5)))[None, :, None].astype(np.float32)
  rng2 = np.random.default_rng(22)        # its own stream: the Add case below keeps drawing from rng
  x4 = rng2.standard_normal((2, 6, 3, 2)).astype(np.float32)
  d = dict(x=x, x_small=xs, x_4d=x4)
  for method in ('nearest', 'linear', 'cubic', 'window'):
    for add_endpoint in (True, False):
      tag = '%s_%s' % (method, 'endpoint' if add_endpoint else 'noendpoint')
      n_up = 576 if add_endpoint else 640                     # 9 frames: 576 = 9 * 64, 640 = 8 * 80
      d['up_' + tag] = a(core.resample(x, n_up, method=method, add_endpoint=add_endpoint))
      d['small_' + tag] = a(core.resample(xs, 160, method=method, add_endpoint=add_endpoint))
      if method != 'window':
        d['ragged_' + tag] = a(core.resample(x, 100, method=method, add_endpoint=add_endpoint))
        d['down_' + tag] = a(core.resample(d['up_' + tag], 7, method=method, add_endpoint=add_endpoint))
        d['x4d_' + tag] = a(core.resample(x4, 48, method=method, add_endpoint=add_endpoint))
  cases['resample_methods'] = d

  # --- core.fft_convolve with padding='valid' / explicit delay (core_test.py:730-757 and beyond) ---
  audio_v = rng2.standard_normal((2, 640)).astype(np.float32)
  ir_v = (rng2.standard_normal((2, 10, 33)) / np.sqrt(33)).astype(np.float32)
  ir_one = (rng2.standard_normal((1, 1, 50)) / np.sqrt(50)).astype(np.float32)
  cases['fft_convolve_crops'] = dict(
      audio=audio_v, ir=ir_v, ir_one=ir_one,
      valid_d0=a(core.fft_convolve(audio_v, ir_v, padding='valid', delay_compensation=0)),
      valid_d5=a(core.fft_convolve(audio_v, ir_v, padding='valid', delay_compensation=5)),
      valid_auto=a(core.fft_convolve(audio_v, ir_v, padding='valid', delay_compensation=-1)),
      same_d40=a(core.fft_convolve(audio_v, ir_v, padding='same', delay_compensation=40)),
      one_valid_d0=a(core.fft_convolve(audio_v, ir_one, padding='valid', delay_compensation=0)),
      one_valid_auto=a(core.fft_convolve(audio_v, ir_one, padding='valid', delay_compensation=-1)))

  # --- core.harmonic_synthesis with harmonic_shifts / 'cubic' / 'nearest' envelopes / ragged n_samples ---
  def synthesis_case(seed, batch, n_frames, n_harm, n_samples, sr, method, shifts, with_hd=True, angular=False):
    r = np.random.default_rng(seed)
    f0 = r.uniform(150.0, 500.0, (batch, n_frames, 1)).astype(np.float32)
    amps = r.uniform(0.1, 1.0, (batch, n_frames, 1)).astype(np.float32)
    hd = r.uniform(0.0, 1.0, (batch, n_frames, n_harm)).astype(np.float32) if with_hd else None
    sh = (0.02 * r.standard_normal((batch, n_frames, n_harm))).astype(np.float32) if shifts else None
    audio = core.harmonic_synthesis(frequencies=f0, amplitudes=amps, harmonic_shifts=sh,
                                    harmonic_distribution=hd, n_samples=n_samples, sample_rate=sr,
                                    amp_resample_method=method, use_angular_cumsum=angular)
    out = dict(f0_hz=f0, amplitudes=amps, n_samples=n_samples, sample_rate=sr, amp_method=method,
               angular=int(angular), audio=a(audio))
    if with_hd:
      out['harmonic_distribution'] = hd
    if shifts:
      out['harmonic_shifts'] = sh
    return out
  cases['synthesis_shifts_window'] = synthesis_case(61, 2, 10, 12, 640, 16000, 'window', True)
  cases['synthesis_shifts_only'] = synthesis_case(62, 1, 8, 6, 512, 16000, 'linear', True, with_hd=False)
  cases['synthesis_cubic'] = synthesis_case(63, 2, 10, 12, 640, 16000, 'cubic', False)
  cases['synthesis_nearest_angular'] = synthesis_case(64, 1, 10, 20, 640, 16000, 'nearest', False, angular=True)
  cases['synthesis_linear_ragged'] = synthesis_case(65, 2, 9, 10, 1000, 16000, 'linear', False)   # 1000 % 9 != 0

  # synths.Harmonic with a 'cubic' amplitude envelope (constructor argument, synths.py:59-66)
  cases['harmonic_cubic_amp'] = harmonic_case(7, 2, 25, 12, 1600, 16000, 200.0, 700.0, 'cubic', False)

  # --- Add ---
  s1 = rng.standard_normal((2, 64)).astype(np.float32)
  s2 = rng.standard_normal((2, 64)).astype(np.float32)
  cases['add'] = dict(signal_one=s1, signal_two=s2, signal=a(processors.Add()(s1, s2)))

  for name, d in cases.items():
    path = os.path.join(HERE, name + SUFFIX)
    d = {k: np.asarray(v) for k, v in d.items()}
    if os.path.exists(path):                       # leave identical fixtures alone (zip timestamps differ)
      with np.load(path) as z:
        if sorted(z.files) == sorted(d) and all(np.array_equal(z[k], d[k]) for k in d):
          print('unchanged', os.path.relpath(path))
          continue
    np.savez_compressed(path, **d)
    print('wrote', os.path.relpath(path), {k: v.shape for k, v in d.items() if v.ndim})


if __name__ == '__main__':
  main()
