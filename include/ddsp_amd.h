/*
 * ddsp_amd.h — C ABI of the MI355X (gfx950) additive-synthesis hot path.
 *
 * The reference (magenta/ddsp) has no native code and no FFI: its boundary for this
 * path is the Python class API `ddsp.processors.Processor` (ddsp/processors.py:37-76)
 * as implemented by `ddsp.synths.Harmonic` (ddsp/synths.py:55-146),
 * `ddsp.synths.FilteredNoise` (ddsp/synths.py:149-196) and `ddsp.processors.Add`
 * (ddsp/processors.py:162-176).  The entry points below are what a binding for that
 * class API calls (see INTEGRATION.md for the ctypes stub a maintainer would add);
 * each one names the reference code it replaces.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller, fp32, contiguous,
 *     row-major [batch, time, channel] exactly as the reference lays tensors out;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it and
 *     never synchronise, allocate or free;
 *   - return value: DDSP_OK or a negative DDSP_ERR_* (programmer errors only; the
 *     reference's ValueErrors are raised by the host layer before the call);
 *   - `workspace` is caller-provided scratch of at least *_workspace_bytes() bytes,
 *     16-byte aligned, contents undefined on entry and exit.
 */
#ifndef DDSP_AMD_H_
#define DDSP_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDSP_OK 0
#define DDSP_ERR_NULL_POINTER (-1)
#define DDSP_ERR_BAD_SHAPE (-2)
#define DDSP_ERR_UNSUPPORTED (-3)   /* shape outside what the kernels implement */
#define DDSP_ERR_WORKSPACE (-4)     /* workspace too small or misaligned */
#define DDSP_ERR_LAUNCH (-5)        /* hipGetLastError() != hipSuccess after a launch */

/* ---- flags for the Harmonic entry points (ddsp/synths.py:59-66 ctor kwargs) ------- */
#define DDSP_HARM_SCALE_EXP_SIGMOID 0x1u  /* scale_fn=core.exp_sigmoid (else scale_fn=None) */
#define DDSP_HARM_NORMALIZE_NYQUIST 0x2u  /* normalize_below_nyquist=True */
#define DDSP_HARM_AMP_LINEAR 0x4u         /* amp_resample_method='linear' (else 'window') */
#define DDSP_HARM_ANGULAR_CUMSUM 0x8u     /* use_angular_cumsum=True (phase) */
#define DDSP_HARM_NO_AUDIO_RATE_MASK 0x10u      /* internal to the streaming entry: no per-sample Nyquist mask */
#define DDSP_HARM_INPUTS_ARE_AMPLITUDES 0x20u   /* streaming entry: harmonic_distribution=None (core.py:1149-1150) */
#define DDSP_HARM_DIRECT_SUM 0x40u        /* ddsp_harmonic_f32: sum the harmonics sample by sample (sine recurrence on the
                                             vector ALUs) even where the matrix-core wavetable kernel applies */

/* ---- flags for the FilteredNoise entry points (ddsp/synths.py:153-163) ------------ */
#define DDSP_NOISE_SCALE_EXP_SIGMOID 0x1u /* scale_fn=core.exp_sigmoid on (mag + initial_bias) */
#define DDSP_NOISE_FIR_VECTOR_ALU 0x8u    /* ddsp_filtered_noise_f32, canonical filter (65 bands, full window, frames of 64 c samples):
                                             keep the time-varying FIR on the vector ALUs (noise_fused65_kernel) instead of the
                                             default matrix-core kernel (noise_mfma65_kernel: IR design and FIR as fp16 hi/lo-split
                                             MFMA products, fp32 accumulation) */
#define DDSP_NOISE_BITS_23 0x10u          /* generated noise (noise == NULL) with 23-bit samples - the 2^23 levels of the reference's
                                             tf.random.uniform (ddsp/synths.py:192-193) - instead of 2048 levels.  The Python mirror
                                             sets it by default since round 6 (FilteredNoise(noise_bits=23) is the class default;
                                             noise_bits=11 clears it); see ddsp_filtered_noise_f32.  The backward call
                                             takes the same flag so that it regenerates the same samples. */

/* Library / build identification: "ddsp_amd <version> gfx950". */
const char* ddsp_version(void);

/* No entry point of this library synchronises or allocates, with ONE exception: the constant operand tables of the
 * matrix-core kernels (the wavetable kernels' sine fragments per device; the general FilteredNoise / frequency_impulse_response
 * kernels' design matrix per device, band count and window size) are made on the host and copied to the device, synchronously,
 * the first time a shape needs them.  ddsp_prepare makes them NOW for the current device: call it before capturing a HIP graph
 * (a synchronous copy inside a capture is an error) or before a latency-critical first call.  n_harmonics <= 0 / n_noise_bands
 * < 2 skip their part; shapes whose kernels need no tables (more than 200 harmonics, the canonical 65-band filter with the
 * full window, whose tables are compile-time constants - it still has a design matrix for its backward pass) are no-ops. */
int ddsp_prepare(int n_harmonics, int n_noise_bands, int window_size);

/* ------------------------------------------------------------------------------------
 * Harmonic.get_controls  (ddsp/synths.py:94-121; core.exp_sigmoid core.py:386-404,
 * core.normalize_harmonics core.py:894-907, remove_above_nyquist core.py:869-891,
 * safe_divide core.py:207-210).
 *   amplitudes            [B,F,1]  in
 *   harmonic_distribution [B,F,K]  in
 *   f0_hz                 [B,F,1]  in
 *   ctl_amplitudes        [B,F,1]  out
 *   ctl_harmonic_distribution [B,F,K] out
 * flags: DDSP_HARM_SCALE_EXP_SIGMOID, DDSP_HARM_NORMALIZE_NYQUIST.
 */
int ddsp_harmonic_controls_f32(const float* amplitudes, const float* harmonic_distribution,
                               const float* f0_hz, float* ctl_amplitudes,
                               float* ctl_harmonic_distribution, int B, int F, int K,
                               int sample_rate, unsigned flags, void* stream);

/* ------------------------------------------------------------------------------------
 * Harmonic.get_signal  (ddsp/synths.py:123-146 -> core.harmonic_synthesis
 * core.py:1048-1111: get_harmonic_frequencies :1028-1045, resample 'linear' :573-642,
 * upsample_with_windows :645-714, oscillator_bank :912-962, angular_cumsum :800-866).
 * Inputs are CONTROLS (already scaled / normalised).  audio [B,N] out.
 * flags: DDSP_HARM_AMP_LINEAR, DDSP_HARM_ANGULAR_CUMSUM.
 * Requires N % F == 0 (the reference requires it for 'window'; DDSP_ERR_UNSUPPORTED
 * otherwise).
 */
size_t ddsp_harmonic_workspace_bytes(int B, int F, int K, int N);
int ddsp_harmonic_signal_f32(const float* ctl_amplitudes,
                             const float* ctl_harmonic_distribution, const float* f0_hz,
                             float* audio, void* workspace, size_t workspace_bytes, int B,
                             int F, int K, int N, int sample_rate, unsigned flags,
                             void* stream);

/* Validation-only variant of ddsp_harmonic_signal_f32 that follows the reference's fp32 op order
 * exactly, including the strictly sequential phase accumulation of tf.cumsum (ddsp/core.py:955)
 * or angular_cumsum (:800-866, chunk 1000).  One lane per (row, harmonic), serial in time: slow by
 * construction, used by the parity tests to compare full-length clips with the fp32-faithful
 * oracle.  flags: DDSP_HARM_AMP_LINEAR, DDSP_HARM_ANGULAR_CUMSUM.  audio [B,N] out. */
int ddsp_harmonic_signal_tf_order_f32(const float* ctl_amplitudes,
                                      const float* ctl_harmonic_distribution, const float* f0_hz,
                                      float* audio, int B, int F, int K, int N, int sample_rate,
                                      unsigned flags, void* stream);

/* ------------------------------------------------------------------------------------
 * Backward pass of ddsp_harmonic_f32 (inputs_are_controls = 0) / ddsp_harmonic_signal_f32
 * (inputs_are_controls = 1): the gradients tf.GradientTape forms through
 * ddsp/synths.py:94-146 in ddsp/training/trainers.py:162-171.
 *   grad_audio [B,N] in;  grad_amplitudes [B,F,1], grad_harmonic_distribution [B,F,K] out.
 * f0_hz is a constant of this call (its gradient is ddsp_harmonic_f0_grad_f32); the Nyquist masks
 * have zero gradient, as tf.where gives them.  flags as ddsp_harmonic_f32.  K <= 256, N/F <= 2048.
 * workspace: ddsp_harmonic_backward_workspace_bytes(B,F,K,N).
 */
size_t ddsp_harmonic_backward_workspace_bytes(int B, int F, int K, int N);
int ddsp_harmonic_backward_f32(const float* amplitudes, const float* harmonic_distribution,
                               const float* f0_hz, const float* grad_audio, float* grad_amplitudes,
                               float* grad_harmonic_distribution, void* workspace,
                               size_t workspace_bytes, int B, int F, int K, int N, int sample_rate,
                               unsigned flags, int inputs_are_controls, void* stream);
/* The frame-rate half of that backward pass on its own: dL/d (amplitudes * harmonic_distribution) [B,F,K] in
 * (core.harmonic_synthesis' product, ddsp/core.py:1097), through Harmonic.get_controls (exp_sigmoid, the frame-rate Nyquist
 * mask, safe_divide by the sum: ddsp/synths.py:94-121) to dL/d amplitudes [B,F,1] and dL/d harmonic_distribution [B,F,K].
 * With ddsp_oscillator_bank_grad_amplitudes_f32 and ddsp_resample_ex_backward_f32 the backward pass of the materialised
 * chain ('nearest' / 'cubic' envelopes, n_samples that is not a multiple of n_frames).  K <= 512. */
int ddsp_harmonic_controls_backward_f32(const float* amplitudes, const float* harmonic_distribution, const float* f0_hz,
                                        const float* grad_harmonic_amplitudes, float* grad_amplitudes,
                                        float* grad_harmonic_distribution, int B, int F, int K, int sample_rate,
                                        unsigned flags, int inputs_are_controls, void* stream);

/* ------------------------------------------------------------------------------------
 * core.streaming_harmonic_synthesis (ddsp/core.py:1114-1164) with harmonic_oscillator_bank
 * (core.py:966-1025, angular cumsum): one chunk of audio from frame-wise controls with the
 * fundamental's phase carried in and out (the VST model, ddsp/training/inference.py:446-472,
 * calls it with F = 2 frames per hop).
 *   amplitudes [B,F,1], harmonic_distribution [B,F,K] (normalised with the frame-rate Nyquist
 *   mask, core.py:1141-1148; with DDSP_HARM_INPUTS_ARE_AMPLITUDES the harmonic amplitudes are
 *   amplitudes * harmonic_distribution as given - pass ones [B,F,1] for the reference's
 *   harmonic_distribution=None), f0_hz [B,F,1],
 *   initial_phase [B] radians or NULL (= 0), audio [B,N] out, final_phase [B] out or NULL:
 *   (sum of omega mod 2 pi) + initial_phase, as the reference returns it.
 * There is no audio-rate Nyquist mask on this path (as in the reference).  N % F == 0.
 * flags: DDSP_HARM_AMP_LINEAR (amp_resample_method, default 'linear' in the reference),
 * DDSP_HARM_INPUTS_ARE_AMPLITUDES.  workspace: ddsp_harmonic_workspace_bytes(B,F,K,N).
 */
int ddsp_harmonic_streaming_f32(const float* amplitudes, const float* harmonic_distribution,
                                const float* f0_hz, const float* initial_phase, float* audio,
                                float* final_phase, void* workspace, size_t workspace_bytes,
                                int B, int F, int K, int N, int sample_rate, unsigned flags,
                                void* stream);

/* ------------------------------------------------------------------------------------
 * Harmonic.__call__  == get_signal(**get_controls(...))  (ddsp/processors.py:53-68),
 * fused: raw network outputs in, audio out, the [B,F,K] controls never round-trip
 * through fp32 HBM tensors unless the caller asks for them.
 *   ctl_amplitudes / ctl_harmonic_distribution: NULL, or out buffers that receive the
 *   controls dict (return_outputs_dict=True).
 * flags: all DDSP_HARM_*.
 */
int ddsp_harmonic_f32(const float* amplitudes, const float* harmonic_distribution,
                      const float* f0_hz, float* audio, float* ctl_amplitudes,
                      float* ctl_harmonic_distribution, void* workspace,
                      size_t workspace_bytes, int B, int F, int K, int N, int sample_rate,
                      unsigned flags, void* stream);

/* Harmonic.__call__ with the processors.Add that follows it fused in (ddsp/processors.py:162-176;
 * every shipped DAG ends Harmonic, FilteredNoise, Add: ddsp/training/gin/models/ae.gin:49-56):
 *   audio[B,N] = Harmonic(amplitudes, harmonic_distribution, f0_hz) + add_signal[B,N]
 * in one launch - one [B,N] stream written (14.44 instead of 18.44 bytes per sample for the
 * Harmonic + FilteredNoise + Add group).  add_signal may be `audio` itself.
 * Bit-identical to ddsp_harmonic_f32 followed by ddsp_add_f32.  Returns DDSP_ERR_UNSUPPORTED where
 * the wavetable kernel does not apply (hop % 64 != 0, K > 200, non-default flags): the caller
 * then uses the two separate entries. */
int ddsp_harmonic_add_f32(const float* amplitudes, const float* harmonic_distribution,
                          const float* f0_hz, const float* add_signal, float* audio, int B, int F,
                          int K, int N, int sample_rate, unsigned flags, void* stream);

/* ------------------------------------------------------------------------------------
 * FilteredNoise.get_controls  (ddsp/synths.py:165-179):
 *   ctl_magnitudes[B,F,M] = exp_sigmoid(magnitudes + initial_bias)   (or identity copy).
 */
int ddsp_filtered_noise_controls_f32(const float* magnitudes, float* ctl_magnitudes, int B,
                                     int F, int M, float initial_bias, unsigned flags,
                                     void* stream);

/* ------------------------------------------------------------------------------------
 * core.frequency_impulse_response  (ddsp/core.py:1534-1565 +
 * apply_window_to_impulse_response :1477-1531): ctl magnitudes [B,F,M] -> causal,
 * windowed FIR taps [B,F,L].  ddsp_fir_size() gives L for (M, window_size).
 */
int ddsp_fir_size(int M, int window_size);
/* core.apply_window_to_impulse_response (core.py:1477-1531) on its own, any response length: [rows, L0] zero-phase (causal != 0:
 * causal) responses -> [rows, ddsp_window_impulse_response_size(L0, window_size)] windowed causal ones. */
int ddsp_window_impulse_response_size(int L0, int window_size);
int ddsp_apply_window_to_impulse_response_f32(const float* impulse_response, float* out, long rows, int L0, int window_size,
                                              int causal, void* stream);
int ddsp_frequency_impulse_response_f32(const float* ctl_magnitudes, float* impulse_response,
                                        int B, int F, int M, int window_size, void* stream);

/* ------------------------------------------------------------------------------------
 * FilteredNoise.get_signal  (ddsp/synths.py:181-196 -> core.frequency_filter
 * core.py:1628-1655 -> fft_convolve :1382-1473, crop_and_compensate_delay :1338-1379).
 *   magnitudes [B,F,M]: raw (flags has DDSP_NOISE_SCALE_EXP_SIGMOID: get_controls is
 *                       fused in) or controls (flag clear: pure get_signal).
 *   noise      [B,N] or NULL.  NULL: uniform noise in (-1,1) is generated on chip, the
 *              stand-in for the reference's tf.random.uniform (synths.py:192-193), whose
 *              stateful stream cannot be reproduced outside TensorFlow.  The generator is
 *              Philox4x32-10 with key = (seed low word, seed high word) and counter
 *              (n >> 3, batch_offset + row, c2, 0) for sample n of a row; this library's
 *              contract (csrc/common.h; oracle/ddsp_oracle.py::device_uniform_noise
 *              restates both forms bit for bit):
 *                flag clear (11 bits, c2 = 0; FilteredNoise(noise_bits=11)): eight samples per block; sample n is the
 *                  11-bit field k = bits [10:0] (n even) or [26:16] (n odd) of word
 *                  (n >> 1) & 3, value (2 k - 2047) / 2048 - 2048 equally spaced levels,
 *                  zero mean, variance 1/3 to 2e-7, every value exactly an fp16 number
 *                  (the matrix-core FIR then needs no lo part for its noise operand);
 *                DDSP_NOISE_BITS_23 (c2 = 1 + ((n >> 2) & 1); what FilteredNoise() asks for by default): four samples per block;
 *                  sample n is word n & 3: its top 23 bits as the mantissa of u in [1,2),
 *                  value 2 u - 3 - the 2^23 levels TensorFlow's fp32 uniforms have.
 *              Non-NULL is the parity entry: the same maths as effects.FIRFilter
 *              (ddsp/effects.py:311-324) / core.frequency_filter on supplied audio, at
 *              any scale (it is normalised by a power of two before the fp16 hi/lo
 *              split of the matrix-core kernels and the outputs are scaled back).
 *   audio      [B,N] out.   ctl_magnitudes: NULL or [B,F,M] out (controls dict).
 * Frames: frame_size = ceil(N/F) and ceil(N/frame_size) must equal F (the reference's
 * ValueError, core.py:1451-1457) else DDSP_ERR_BAD_SHAPE.
 * Workspace: the taps [B,F,L] for the shapes that design them in a launch of their own,
 * plus - for filters that reach across more frames than a tiled kernel holds (510 taps
 * on frames of 5 samples: the plain sum of ddsp_fft_convolve_f32 takes those) - a row of
 * generated noise per clip; always ask ddsp_filtered_noise_workspace_bytes() with the
 * call's own (B, F, M, N, window_size).
 */
size_t ddsp_filtered_noise_workspace_bytes(int B, int F, int M, int N, int window_size);
int ddsp_filtered_noise_f32(const float* magnitudes, const float* noise, float* audio,
                            float* ctl_magnitudes, void* workspace, size_t workspace_bytes,
                            int B, int F, int M, int N, int window_size, float initial_bias,
                            unsigned flags, uint64_t seed, uint64_t batch_offset,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Backward pass of ddsp_filtered_noise_f32: dL/d magnitudes [B,F,M] from grad_audio [B,N]
 * (the gradient tf.GradientTape forms through ddsp/synths.py:165-196).  `noise` as in the forward
 * call: the tensor that was supplied, or NULL with the same seed / batch_offset so that the noise is
 * regenerated.  Any band count / window (M = 65 with the full window has its own fast kernel).
 * workspace: ddsp_filtered_noise_backward_workspace_bytes.
 */
size_t ddsp_filtered_noise_backward_workspace_bytes(int B, int F, int M, int N);
int ddsp_filtered_noise_backward_f32(const float* magnitudes, const float* noise,
                                     const float* grad_audio, float* grad_magnitudes,
                                     void* workspace, size_t workspace_bytes, int B, int F, int M,
                                     int N, int window_size, float initial_bias, unsigned flags,
                                     uint64_t seed, uint64_t batch_offset, void* stream);

/* ------------------------------------------------------------------------------------
 * core.fft_convolve(audio, impulse_response, padding='same', delay_compensation)
 * (ddsp/core.py:1382-1473) evaluated as the equivalent direct time-varying FIR.
 *   audio [B,N], impulse_response [Bir,F,L] with Bir == B or Bir == 1 (broadcast,
 *   core.py:1433-1434), out [B,N].  delay_compensation < 0 selects (L-1)/2 - 1.
 */
int ddsp_fft_convolve_same_f32(const float* audio, const float* impulse_response, float* out,
                               int B, int Bir, int F, int L, int N, int delay_compensation,
                               void* stream);

/* ------------------------------------------------------------------------------------
 * Long single-frame convolution: effects.Reverb.get_signal (ddsp/effects.py:100-117) and
 * core.fft_convolve (ddsp/core.py:1382-1473) with a 2-D impulse response (one frame,
 * core.py:1428-1430), padding='same':
 *     out[b][n] = sum_k ir[b][k] * audio[b][n + delay - k]   (+ audio[b][n] with ADD_DRY),
 * i.e. crop_and_compensate_delay (core.py:1338-1379) with start = delay >= 0
 * (Reverb passes delay_compensation=0).  ir [Bir,L], Bir == B or 1 (tiled over the batch,
 * effects.py:62-69 / core.py:1433-1434).  DDSP_CONV_MASK_TAP0 zeroes tap 0 as
 * Reverb._mask_dry_ir does (effects.py:50-60).  Evaluated as a partitioned overlap-save FFT
 * convolution with LDS-resident 8192-point FFTs; any L < 2^28 (up to 16 partitions of 4096 taps stay in a
 * register window of the multiply-add pass; beyond - vst_48k.gin's 72 000 taps - a plain pass re-reads them).
 * workspace: ddsp_fft_convolve_long_workspace_bytes(...) bytes, 16-byte aligned.
 */
#define DDSP_CONV_ADD_DRY 1u
#define DDSP_CONV_MASK_TAP0 2u
#define DDSP_CONV_REVERSE_AUDIO 4u   /* _ex only: logical audio sample g is stored at N-1-g   */
#define DDSP_CONV_REVERSE_IR 8u      /* _ex only: logical tap k is stored at L-1-k            */
#define DDSP_CONV_REVERSE_OUT 16u    /* _ex only: logical output n is written to n_out-1-n    */
#define DDSP_CONV_ZERO_OUT0 32u      /* _ex only: logical output 0 is written as 0 (dL/d ir: the masked dry tap has no gradient) */
size_t ddsp_fft_convolve_long_workspace_bytes(int B, int Bir, int N, int L, int delay);
int ddsp_fft_convolve_long_f32(const float* audio, const float* impulse_response, float* out,
                               void* workspace, size_t workspace_bytes, int B, int Bir, int N,
                               int L, int delay, unsigned flags, void* stream);
/* The same convolution with n_out outputs per row (out [B,n_out], out[b][n] = y[n + delay] of the
 * full linear convolution y) and optional index reversal of either input or of the output: the two
 * correlations of the Reverb backward pass are
 *   dL/d audio = reverse( conv(reverse(g), ir)[0:N] )           (REVERSE_AUDIO | REVERSE_OUT)
 *   dL/d ir[k] = conv(g, reverse(audio))[N-1+k], k < L          (REVERSE_IR on the audio passed as "ir",
 *                                                                 n_out = L, delay = N-1)
 * The "impulse response" of a call may be any length below 2^28.  ADD_DRY needs n_out == N. */
size_t ddsp_fft_convolve_long_ex_workspace_bytes(int B, int Bir, int N, int L, int n_out, int delay);
int ddsp_fft_convolve_long_ex_f32(const float* audio, const float* impulse_response, float* out,
                                  void* workspace, size_t workspace_bytes, int B, int Bir, int N,
                                  int L, int n_out, int delay, unsigned flags, void* stream);
/* out[l] = sum_b x[b][l] (row 0 first: a fixed order), out[0] = 0 when zero_first: what collects dL/d ir of a Reverb whose one
 * impulse response serves the whole batch (ddsp/effects.py:62-80; the dry tap is masked, :50-60) from the rows' correlations. */
int ddsp_sum_rows_f32(const float* x, float* out, int B, int L, int zero_first, void* stream);

/* ------------------------------------------------------------------------------------
 * losses.SpectralLoss.call, forward pass (ddsp/losses.py:189-243), loss_type 'L1', magnitude and
 * log-magnitude terms:
 *     *loss = sum over fft_sizes of  mag_weight * mean|mag_t - mag_a| + logmag_weight * mean|safe_log mag_t - safe_log mag_a|
 * with mag = |tf.signal.stft(frame_length=S, frame_step=S/4, pad_end=True)| (spectral_ops.py:34-47,
 * 67-70), safe_log per core.py:213-216.  target_audio, audio [B,N] and loss (one float) are device
 * pointers; fft_sizes is a HOST array of n_sizes (<= 16) frame sizes: powers of two in [16, 8192] or (round 6) 3 * 2^k in [48, 6144] -
 * gin/models/vst/vst_48k.gin:56 -, frames of 4 hops zero-padded to the 4 F / 3 points tf.signal.stft transforms (S / 2 + 1 bins).
 * workspace: ddsp_spectral_loss_workspace_bytes(...) bytes (per-block fp64 partial sums; the
 * result does not depend on scheduling).
 */
size_t ddsp_spectral_loss_workspace_bytes(int B, int N, const int* fft_sizes, int n_sizes);
int ddsp_spectral_loss_f32(const float* target_audio, const float* audio, float* loss,
                           void* workspace, size_t workspace_bytes, int B, int N,
                           const int* fft_sizes, int n_sizes, float mag_weight,
                           float logmag_weight, void* stream);

/* Backward pass of ddsp_spectral_loss_f32 with respect to `audio`: grad_audio [B,N] (overwritten)
 * = grad_loss[0] * dL/d audio, grad_loss a device pointer to one float.  |z| has gradient z/|z|
 * (0 at 0), safe_log passes a gradient only where its argument is positive.  Overlapping frames
 * add with fp32 atomics: the last bit may differ between runs. */
int ddsp_spectral_loss_backward_f32(const float* target_audio, const float* audio,
                                    const float* grad_loss, float* grad_audio, int B, int N,
                                    const int* fft_sizes, int n_sizes, float mag_weight,
                                    float logmag_weight, void* stream);

/* Loss value and dL/d audio (for dL/dloss = 1) in one pass: the frame spectra are computed once
 * for both, which is how a training step calls it (the caller scales grad_audio by the upstream
 * gradient).  loss: one float; workspace as ddsp_spectral_loss_f32. */
int ddsp_spectral_loss_value_and_grad_f32(const float* target_audio, const float* audio, float* loss,
                                          float* grad_audio, void* workspace,
                                          size_t workspace_bytes, int B, int N,
                                          const int* fft_sizes, int n_sizes, float mag_weight,
                                          float logmag_weight, void* stream);

/* ---- the general form of losses.SpectralLoss (ddsp/losses.py:131-243): every term, loss_type and the weights mask ----
 * The 'L1' mag + logmag loss of the shipped configs is ddsp_spectral_loss_f32 above (spectra stay in LDS).  These three
 * entries serve the rest of the reference's argument space on spectrograms materialised in HBM, one FFT size at a time:
 *
 * ddsp_stft_mag_f32: spectral_ops.compute_mag (ddsp/spectral_ops.py:67-70; tf.signal.stft: frames of fft_size every
 *   fft_size/4, zero pad_end, periodic Hann) of two signals at once: target_mag, mag [B, frames, fft_size/2+1],
 *   frames = ceil(N / (fft_size/4)).
 * ddsp_spectral_terms_f32: for one FFT size, adds
 *     mag_weight * D(T, V) + delta_time_weight * D(diff_t T, diff_t V) + delta_freq_weight * D(diff_f T, diff_f V)
 *     + cumsum_freq_weight * D(cumsum_f T, cumsum_f V) + logmag_weight * D(safe_log T, safe_log V)
 *   to *loss_accumulator (fp64, device; `first` != 0 starts it at 0) and rewrites *loss (fp32) with the running total;
 *   D = losses.mean_difference (ddsp/losses.py:102-128) with loss_type DDSP_LOSS_L1 / _L2 / _COSINE and the optional
 *   `weights` mask of shape [weights_b, weights_f, weights_k] (each extent 1 or the term's; NULL: no mask).  Terms
 *   with weight <= 0 are skipped, as in the reference.  grad_value_mag (may be NULL) [B, frames, bins] receives
 *   d(this size's contribution)/dV.  Deterministic (fixed-order fp64 sums).
 * ddsp_stft_mag_backward_f32: grad_audio [B,N] += d|STFT(audio)|^T grad_mag  (|z| has gradient z/|z|, 0 at 0). */
#define DDSP_LOSS_L1 0
#define DDSP_LOSS_L2 1
#define DDSP_LOSS_COSINE 2
int ddsp_stft_mag_f32(const float* target_audio, const float* audio, float* target_mag, float* mag, int B, int N,
                      int fft_size, void* stream);
size_t ddsp_spectral_terms_workspace_bytes(int B, int frames);
int ddsp_spectral_terms_f32(const float* target_mag, const float* value_mag, const float* weights, int weights_b,
                            int weights_f, int weights_k, float* grad_value_mag, double* loss_accumulator, float* loss,
                            void* workspace, size_t workspace_bytes, int B, int frames, int bins, int loss_type,
                            float mag_weight, float delta_time_weight, float delta_freq_weight,
                            float cumsum_freq_weight, float logmag_weight, int first, void* stream);
int ddsp_stft_mag_backward_f32(const float* audio, const float* grad_mag, float* grad_audio, int B, int N, int fft_size,
                               void* stream);
/* fft_size of the two calls above: a power of two in [16, 8192], or any other EVEN frame size in [34, 8190] (gin/models/vst/
 * vst_48k.gin:56 asks for 6144, 3072 .. 192; any since round 6) - tf.signal.stft then transforms the enclosing power of two S,
 * the frame zero-padded, every int(fft_size / 4) samples: bins = S / 2 + 1.
 *
 * The loudness term of SpectralLoss (ddsp/losses.py:238-242 -> spectral_ops.compute_loudness, spectral_ops.py:253-324), in the
 * same materialised form.  ddsp_stft_frames_mag_f32: |STFT| of ONE signal under the caller's frame geometry - frames of
 * fft_size samples (a power of two in [64, 8192]) every `hop`, the first starting pad_left samples before sample 0 (zeros
 * outside the row; compute_loudness: fft_size 2048, hop sample_rate / 250, pad_left 1024, n_frames 1 + N / hop), periodic Hann
 * -> mag [B, n_frames, fft_size/2+1]; ..._backward: grad_audio += its adjoint applied to grad_mag (overlapping frames are
 * added with fp32 atomics: the LAST BITS of this gradient - and of ddsp_stft_mag_backward_f32's for frames of 3 * 2^k samples,
 * and of ddsp_spectral_loss_backward_f32's - depend on the order the frames arrive in; every FORWARD entry point, and the synths'
 * backward passes, are bit-reproducible run to run).
 * ddsp_loudness_from_mag_f32 (weighting: `bins` entries, caller-checked - the Python mirror refuses an n_fft whose spectrogram has
 * another bin count): loudness [B, n_frames] = max(10 log10(max(pmin, mean_k weighting[k] mag[k]^2)) - ref_db, -range_db),
 * pmin = 10^(-range_db/10) (core.power_to_db, core.py:253-267); `weighting` [bins] = 10^(A_weighting/10), made by the caller
 * (librosa's published A-curve: oracle/ddsp_oracle.py::a_weighting_db).  ..._backward: grad_mag [B, n_frames, bins]. */
int ddsp_stft_frames_mag_f32(const float* audio, float* mag, int B, int N, int fft_size, int hop, int pad_left, int n_frames,
                             void* stream);
/* spectral_ops.stft (ddsp/spectral_ops.py:34-47: tf.signal.stft, fft_length=None) - the complex spectrum itself under the same
 * geometry: frames of frame_size samples (even, <= fft_size; periodic Hann of frame_size) zero-padded to fft_size, a power of two
 * in [64, 8192].  spectrum: [B, n_frames, fft_size/2 + 1] pairs (re, im), fp32.  Forward only. */
int ddsp_stft_frames_f32(const float* audio, float* spectrum, int B, int N, int fft_size, int frame_size, int hop, int pad_left,
                         int n_frames, void* stream);
/* ddsp_stft_frames_mag_f32 for frames shorter than the transform (spectral_ops.compute_mag at any even size: frame_size samples
 * under a periodic Hann of frame_size, zero-padded to fft_size).  Forward only. */
int ddsp_stft_frames_mag_ex_f32(const float* audio, float* mag, int B, int N, int fft_size, int frame_size, int hop, int pad_left,
                                int n_frames, void* stream);
int ddsp_stft_frames_mag_backward_f32(const float* audio, const float* grad_mag, float* grad_audio, int B, int N, int fft_size,
                                      int hop, int pad_left, int n_frames, void* stream);
int ddsp_loudness_from_mag_f32(const float* mag, const float* weighting, float* loudness, int B, int n_frames, int bins,
                               float range_db, float ref_db, void* stream);
int ddsp_loudness_from_mag_backward_f32(const float* mag, const float* weighting, const float* grad_loudness, float* grad_mag,
                                        int B, int n_frames, int bins, float range_db, float ref_db, void* stream);

/* Uniform noise exactly as ddsp_filtered_noise_f32 generates it (noise==NULL). out [B,N].
 * _ex: noise_bits = 11 (the flag-clear form, what ddsp_uniform_noise_f32 makes) or 23 (DDSP_NOISE_BITS_23's: the Python
 * mirror's default). */
int ddsp_uniform_noise_f32(float* out, int B, int N, uint64_t seed, uint64_t batch_offset,
                           void* stream);
int ddsp_uniform_noise_ex_f32(float* out, int B, int N, uint64_t seed, uint64_t batch_offset,
                              int noise_bits, void* stream);

/* processors.Add.get_signal (ddsp/processors.py:174-176): out = a + b, n elements. */
int ddsp_add_f32(const float* signal_one, const float* signal_two, float* out, size_t n,
                 void* stream);

/* core.oscillator_bank (ddsp/core.py:912-962) on materialised audio-rate envelopes [B,N,K]:
 * Nyquist mask, phase = inclusive cumsum over time of 2*pi*f/sr, sum_k A_k sin(phase_k).
 * out is [B,N] (sum_sinusoids != 0) or [B,N,K].  The phase scan runs in fp64 revolutions.
 * (synths.Harmonic never materialises these tensors; this entry serves direct callers.) */
size_t ddsp_oscillator_bank_workspace_bytes(int B, int N, int K);
int ddsp_oscillator_bank_f32(const float* frequency_envelopes, const float* amplitude_envelopes,
                             float* out, void* workspace, size_t workspace_bytes, int B, int N,
                             int K, int sample_rate, int sum_sinusoids, void* stream);
/* dL/d amplitude_envelopes [B,N,K] of ddsp_oscillator_bank_f32 given dL/d audio [B,N]: grad_audio[n] where the frequency is
 * below Nyquist, times sin(phase[n,k]) (the output is linear in the amplitudes).  Workspace as the forward's. */
int ddsp_oscillator_bank_grad_amplitudes_f32(const float* frequency_envelopes, const float* grad_audio,
                                             float* grad_amplitude_envelopes, void* workspace, size_t workspace_bytes,
                                             int B, int N, int K, int sample_rate, void* stream);
/* dL/d frequency_envelopes [B,N,K]: phase[n] = (2 pi / sr) sum_{t <= n} f[t] (ddsp/core.py:950-955), so
 * dL/d f[t,k] = (2 pi / sr) sum_{n >= t} grad_audio[n] A[n,k] mask[n,k] cos(phase[n,k]) (the Nyquist mask, tf.where, passes none).
 * Workspace: TWICE ddsp_oscillator_bank_workspace_bytes. */
int ddsp_oscillator_bank_grad_frequencies_f32(const float* frequency_envelopes, const float* amplitude_envelopes,
                                              const float* grad_audio, float* grad_frequency_envelopes, void* workspace,
                                              size_t workspace_bytes, int B, int N, int K, int sample_rate, void* stream);
/* dL/d f0_hz [B,F,1] from dL/d harmonic_frequencies [B,F,K] (= f0 [1..K] (1 + harmonic_shifts), ddsp/core.py:1086-1090;
 * harmonic_shifts may be NULL). */
int ddsp_harmonic_frequencies_backward_f32(const float* grad_harmonic_frequencies, const float* harmonic_shifts,
                                           float* grad_f0_hz, int B, int F, int K, void* stream);

/* core.resample (ddsp/core.py:573-642) for [B,F,C] -> [B,N,C], add_endpoint=True:
 *   window == 0: method='linear' (tf.compat.v1.image.resize BILINEAR, align_corners=False);
 *   window == 1: method='window' (core.upsample_with_windows :645-714); needs N % F == 0.
 * The synth kernels evaluate these envelopes on the fly; this entry is for direct callers. */
int ddsp_resample_f32(const float* x, float* out, int B, int F, int N, int C, int window,
                      void* stream);

/* core.resample (ddsp/core.py:573-642), the whole argument space: [B,F,C] -> [B,N,C] (a 4-D input
 * [B,F,n_freq,C] is the same call with C = n_freq*C: the width axis is resized 1:1), up- or down-sampling.
 *   method: DDSP_RESAMPLE_NEAREST / _LINEAR / _CUBIC = tf.compat.v1.image.resize NEAREST_NEIGHBOR / BILINEAR /
 *           BICUBIC (legacy kernels: no half-pixel centres, align_corners = !add_endpoint; bicubic with the
 *           1024-entry A = -0.75 table and clamped indices); DDSP_RESAMPLE_WINDOW = core.upsample_with_windows
 *           (:645-714; upsampling only, N divisible by F, or by F-1 when add_endpoint == 0, else
 *           DDSP_ERR_BAD_SHAPE - the reference's ValueErrors are raised by the host layer first).
 * (ddsp/csrc/general.hip) */
#define DDSP_RESAMPLE_NEAREST 0
#define DDSP_RESAMPLE_LINEAR 1
#define DDSP_RESAMPLE_CUBIC 2
#define DDSP_RESAMPLE_WINDOW 3
int ddsp_resample_ex_f32(const float* x, float* out, int B, int F, int N, int C, int method,
                         int add_endpoint, void* stream);
/* Its adjoint - what tf.GradientTape forms through core.resample (ddsp/training/trainers.py:162-171): grad_out [B,N,C] ->
 * grad_in [B,F,C], grad_in[j] = sum_t W[t][j] grad_out[t] with the forward's own fp32 positions, clamped indices and
 * weights; gathered per frame (no atomics: deterministic).  (csrc/general.hip) */
int ddsp_resample_ex_backward_f32(const float* grad_out, float* grad_in, int B, int F, int N, int C, int method,
                                  int add_endpoint, void* stream);

/* core.fft_convolve (ddsp/core.py:1382-1473) with any crop (crop_and_compensate_delay :1338-1379):
 *     out[b][n] = z[b][n + start],  n < n_out,   z[m] = sum_i audio[i] * ir[frame(i)][m - i]
 * (z = the overlap-added framed FFT products = the direct time-varying FIR; zero beyond its support).
 * padding='valid' is n_out = L + N - 1, 'same' is n_out = N; start = delay_compensation, or
 * (L-1)/2 - 1 when that is negative.  audio [B,N], impulse_response [Bir,F,L] with Bir == B or 1,
 * out [B,n_out].  frame_size = ceil(N/F) and ceil(N/frame_size) must equal F (DDSP_ERR_BAD_SHAPE).
 * General shapes, one thread per output; ddsp_fft_convolve_same_f32 is the fast entry for 'same'. */
int ddsp_fft_convolve_f32(const float* audio, const float* impulse_response, float* out, int B, int Bir,
                          int F, int L, int N, int n_out, int start, void* stream);

/* out[i] = x[i] * scale[0] (scale: one float in device memory): the upstream scalar of a loss's backward pass applied
 * to a stored gradient - the chain rule tf.GradientTape applies through SpectralLoss (ddsp/training/trainers.py:162-171). */
int ddsp_scale_f32(const float* x, const float* scale, float* out, size_t n, void* stream);

/* core.harmonic_oscillator_bank (ddsp/core.py:966-1025) on audio-rate inputs: frequency [B,N,1] (one fundamental per clip),
 * amplitude_envelopes [B,N,K], initial_phase [B] radians or NULL -> audio [B,N] = sum_k A[n,k] sin((k+1) phase[n]),
 * phase = cumsum(2 pi f / sr) + initial_phase, and final_phase [B] = phase[N-1] (use_angular_cumsum != 0: the scan's
 * part wrapped to [0, 2 pi) as core.angular_cumsum leaves it, ddsp/core.py:800-866).  No Nyquist mask, as in the
 * reference.  The phase scan runs in fp64 revolutions.  Caller: core.streaming_harmonic_synthesis with envelopes
 * the closed-form kernel does not take, ddsp/training/inference.py:446-472 through it. */
size_t ddsp_harmonic_oscillator_bank_workspace_bytes(int B, int N);
int ddsp_harmonic_oscillator_bank_f32(const float* frequency, const float* amplitude_envelopes,
                                      const float* initial_phase, float* audio, float* final_phase,
                                      void* workspace, size_t workspace_bytes, int B, int N, int K,
                                      int sample_rate, int use_angular_cumsum, void* stream);

/* The frame-rate tensors core.harmonic_synthesis builds before resampling (ddsp/core.py:1080-1098,
 * get_harmonic_frequencies :1028-1045), fp32 in the reference's op order:
 *   harmonic_frequencies[B,F,K] = (f0_hz * (k+1)) * (1 + harmonic_shifts)      (shifts may be NULL)
 *   harmonic_amplitudes [B,F,K] = amplitudes * harmonic_distribution           (distribution NULL: amplitudes)
 * Used for the arguments the fused kernels do not take (harmonic_shifts, amp_resample_method 'nearest' /
 * 'cubic'): the host layer then follows the reference's own chain resample -> oscillator_bank on these. */
int ddsp_harmonic_envelopes_f32(const float* amplitudes, const float* harmonic_distribution,
                                const float* f0_hz, const float* harmonic_shifts,
                                float* harmonic_frequencies, float* harmonic_amplitudes, int B, int F,
                                int K, void* stream);

/* dL/d f0_hz [B,F,1] of Harmonic given grad_audio [B,N]: the gradient tf.GradientTape forms through
 * tf.cumsum and tf.sin (ddsp/core.py:950-960) and the bilinear resize of the frequencies (:1101).  Inputs
 * are the CONTROLS (ddsp_harmonic_controls_f32 / get_controls outputs) and f0_hz; the Nyquist masks have
 * zero gradient.  flags: DDSP_HARM_AMP_LINEAR.  N % F == 0.  workspace:
 * ddsp_harmonic_f0_grad_workspace_bytes(B,F,K,N). */
size_t ddsp_harmonic_f0_grad_workspace_bytes(int B, int F, int K, int N);
int ddsp_harmonic_f0_grad_f32(const float* ctl_amplitudes, const float* ctl_harmonic_distribution,
                              const float* f0_hz, const float* grad_audio, float* grad_f0,
                              void* workspace, size_t workspace_bytes, int B, int F, int K, int N,
                              int sample_rate, unsigned flags, void* stream);

/* effects.ExpDecayReverb._get_ir (ddsp/effects.py:144-151):
 *     ir[b][i] = G(gain[b]) * exp(-(2 + exp(decay[b])) * t_i) * noise[i],   t = linspace(0, 1, L)
 * gain, decay [B] (the reference's [B,1]); noise [L]: ONE burst shared by the batch, as the reference draws
 * tf.random.uniform([1, L], -1, 1) (ddsp_uniform_noise_f32 with B = 1 is the on-chip stand-in); ir [B,L] out.
 * flags: DDSP_DECAY_SCALE_EXP_SIGMOID = scale_fn is core.exp_sigmoid (else gain is used as given).
 * _backward: grad_ir [B,L] in, grad_gain / grad_decay [B] out; workspace:
 * ddsp_exp_decay_ir_backward_workspace_bytes(B, L). */
#define DDSP_DECAY_SCALE_EXP_SIGMOID 0x1u
int ddsp_exp_decay_ir_f32(const float* gain, const float* decay, const float* noise, float* ir, int B, int L,
                          unsigned flags, void* stream);
size_t ddsp_exp_decay_ir_backward_workspace_bytes(int B, int L);
int ddsp_exp_decay_ir_backward_f32(const float* gain, const float* decay, const float* noise,
                                   const float* grad_ir, float* grad_gain, float* grad_decay,
                                   void* workspace, size_t workspace_bytes, int B, int L, unsigned flags,
                                   void* stream);

/* processors.Mix (ddsp/processors.py:180-233), the constant-power crossfade next to processors.Add:
 *   ddsp_sigmoid_f32: tf.nn.sigmoid on n values (get_controls, :207; in may equal out);
 *   ddsp_mix_f32:     out[r][c] = sqrt(|m[r]|) * signal_one[r][c] + (1 - sqrt(|m[r] - 1|)) * signal_two[r][c]
 *                     for rows = batch * n_time time steps of C channels (get_signal, :231-233);
 *   ddsp_sigmoid_backward_f32, ddsp_mix_backward_f32: their adjoints, what tf.GradientTape takes through the two
 *                     (grad_in = grad_out s (1 - s); grad_one = sqrt|m| g, grad_two = (1 - sqrt|m - 1|) g, grad_level [rows] =
 *                     sum_c g (one sign(m) / (2 sqrt|m|) - two sign(m - 1) / (2 sqrt|m - 1|)); any of the three outputs may be
 *                     NULL; no atomics: bit-reproducible). */
int ddsp_sigmoid_f32(const float* in, float* out, size_t n, void* stream);
int ddsp_mix_f32(const float* signal_one, const float* signal_two, const float* mix_level, float* out,
                 size_t rows, int C, void* stream);
int ddsp_sigmoid_backward_f32(const float* in, const float* grad_out, float* grad_in, size_t n, void* stream);
int ddsp_mix_backward_f32(const float* signal_one, const float* signal_two, const float* mix_level, const float* grad_out,
                          float* grad_one, float* grad_two, float* grad_level, size_t rows, int C, void* stream);

/* Small pieces of ddsp/core.py that the synths only use fused inside their kernels, callable on their own (the reference
 * exports them; nothing on the hot path calls these):
 *   ddsp_safe_divide_f32           core.safe_divide (ddsp/core.py:207-210): out = numerator / where(denominator == 0, eps,
 *                                  denominator); numerator, out [rows, C]; denominator [rows, den_cols], den_cols = C or 1;
 *   ddsp_safe_log_f32              core.safe_log (:213-216): log(where(x <= 0, eps, x));
 *   ddsp_harmonic_frequencies_f32  core.get_harmonic_frequencies (:1028-1045): frequencies [rows] -> out [rows, n_harmonics],
 *                                  out[r][k] = fl32(frequencies[r] * (k + 1));
 *   ddsp_remove_above_nyquist_f32  core.remove_above_nyquist (:869-891): out = where(frequency >= sample_rate / 2, 0,
 *                                  amplitude) on n values (same shape); sample_rate is a float as in the reference, which
 *                                  compares with `sample_rate / 2.0` whatever number it is handed;
 *   ddsp_angular_cumsum_f32        core.angular_cumsum (:800-866): angular_frequency [B, T, C] (radians per sample) -> the
 *                                  accumulated phase in [0, 2 pi), [B, T, C].  The scan runs in fp64 revolutions (chunk sums,
 *                                  wrapped prefix, running phase): the reference's `chunk_size` has no counterpart - the
 *                                  result is exact to fp32 rounding for any length, where the reference's fp32 chunks drift.
 *                                  workspace: ddsp_angular_cumsum_workspace_bytes(B, T, C), 8-byte aligned. */
int ddsp_safe_divide_f32(const float* numerator, const float* denominator, float* out, size_t rows, int C,
                         int den_cols, float eps, void* stream);
int ddsp_safe_log_f32(const float* x, float* out, size_t n, float eps, void* stream);
int ddsp_harmonic_frequencies_f32(const float* frequencies, float* out, size_t rows, int n_harmonics, void* stream);
int ddsp_remove_above_nyquist_f32(const float* frequency_envelopes, const float* amplitude_envelopes, float* out,
                                  size_t n, float sample_rate, void* stream);
size_t ddsp_angular_cumsum_workspace_bytes(int B, int T, int C);
int ddsp_angular_cumsum_f32(const float* angular_frequency, float* out, void* workspace, size_t workspace_bytes,
                            int B, int T, int C, void* stream);

/* core.exp_sigmoid (ddsp/core.py:386-404), elementwise on n values (in may equal out). */
int ddsp_exp_sigmoid_f32(const float* in, float* out, size_t n, float exponent,
                         float max_value, float threshold, void* stream);

/* ------------------------------------------------------------------------------------
 * Tracing (the reference has none beyond wall-clock logs): opt-in
 * HIP-event brackets around individual kernels, recorded on the stream they are launched
 * on.  ddsp_profile_begin(mask, max_records) turns it on for the kernels whose bit is set
 * in `mask` (bit i = kernel id i, names via ddsp_profile_kernel_name); ddsp_profile_end
 * synchronises the recorded events, fills total_ms[ddsp_profile_kernel_count()] and
 * counts[...], and turns tracing off.  Not for use under stream capture.
 */
int ddsp_profile_kernel_count(void);
const char* ddsp_profile_kernel_name(int kernel_id);
int ddsp_profile_begin(unsigned kernel_mask, int max_records);
/* As ddsp_profile_begin, but only every `stride`-th launch of each selected kernel is bracketed
 * (the first one always is).  A bracketed launch carries a start marker and a stop event, which
 * costs ~5 us of queue time each; sampling keeps a timed region honest while still measuring the
 * kernel inside it.  stride < 1 -> DDSP_ERR_BAD_SHAPE. */
int ddsp_profile_begin_sampled(unsigned kernel_mask, int max_records, int stride);
int ddsp_profile_end(double* total_ms, int* counts);

#ifdef __cplusplus
}
#endif
#endif /* DDSP_AMD_H_ */
