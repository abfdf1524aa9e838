// Internal interface of the matrix-core variant of FilteredNoise.__call__ (filtered_noise_mfma.hip), used by the
// dispatch in filtered_noise.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddsp {

// True when ddsp_filtered_noise_f32 can run on noise_mfma65_kernel: 65 bands, full window (128 taps), frames of
// 64 c samples that tile N exactly as core.fft_convolve frames it (ddsp/core.py:1440-1450).
// (magnitudes squashed by exp_sigmoid only - scale != 0: raw magnitudes of any size take the general kernels, which normalise)
bool noise_mfma65_ok(int F, int M, int N, int padding, const void* noise, int scale);

// bits23: generated noise (noise == null) with 23-bit samples, carried as fp16 hi / lo pairs like supplied noise (DDSP_NOISE_BITS_23);
// row_scratch: B floats of workspace for the rows' max |noise| when noise is supplied (it is normalised before the fp16 split)
int launch_noise_mfma65(const float* magnitudes, const float* noise, float* audio, float* ctl_magnitudes, int B, int F,
                        int N, int start, float initial_bias, int scale, uint64_t seed, uint64_t batch_offset,
                        long long* dbg, int bits23, float* row_scratch, hipStream_t st);

}  // namespace ddsp
