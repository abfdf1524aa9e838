// Internal interface of the matrix-core kernels for FilteredNoise / fft_convolve shapes the canonical kernel
// (filtered_noise_mfma.hip: 65 bands, frames of 64 c samples) does not take - any number of bands up to 288, any window,
// any frame size (filtered_noise_general.hip).  Used by the dispatch in filtered_noise.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddsp {

// IR design (core.frequency_impulse_response + apply_window_to_impulse_response, ddsp/core.py:1477-1565) as ONE matrix
// product with a constant matrix: taps[row][kappa] = sum_m scaled_mag[row][m] C[m][kappa].
bool noise_ir_gemm_ok(int M, int window_size);
int launch_noise_ir_gemm(const float* mag, float* ctl_out, float* ir, long rows, int M, int window_size, float bias,
                         int scale, hipStream_t st);

// the time-varying FIR (core.fft_convolve, ddsp/core.py:1382-1473, as a time-domain sum) for any tap count up to
// kGfMaxTaps and any frame size
bool tv_fir_mfma_ok(int B, int Bir, int F, int L, int N);
// (bits23: noise generated with 23-bit samples - DDSP_NOISE_BITS_23, common.h - when x is null; taps_bounded: the taps are this
// library's own design of exp_sigmoid magnitudes and need no normalisation ahead of the fp16 split)
int launch_tv_fir_mfma(const float* x, const float* ir, float* out, int B, int Bir, int F, int L, int N, int start,
                       uint64_t seed, uint64_t batch_offset, int bits23, int taps_bounded, hipStream_t st);

// FilteredNoise.__call__ in ONE launch for up to 128 bands and 256 taps: tv_fir_mfma_kernel designing its tiles' taps itself
bool filtered_noise_general_fused_ok(int B, int F, int M, int N, int window_size);
int launch_filtered_noise_general_fused(const float* mag, const float* x, float* out, float* ctl_out, int B, int F, int M, int N,
                                        int window_size, float bias, int scale, uint64_t seed, uint64_t batch_offset,
                                        int bits23, hipStream_t st);

// the backward of FilteredNoise.__call__ for 65 bands / full window / frames of 64, 128, 192, 256 samples in ONE launch: the tap
// gradients as Toeplitz products, dL/d magnitudes as a product with the transposed design matrix
bool noise_bwd_mfma_ok(int B, int F, int M, int N, int window_size);
int launch_noise_bwd_mfma(const float* magnitudes, const float* noise, const float* grad_audio, float* grad_magnitudes, int B, int F,
                          int M, int N, int window_size, float bias, int scale, uint64_t seed, uint64_t batch_offset, int bits23,
                          hipStream_t st);

// the design matrix of (bands, window size) and its transpose as fragments on the current device NOW (ddsp_prepare); 0 on success
int noise_general_prepare(int M, int window_size);

}  // namespace ddsp
