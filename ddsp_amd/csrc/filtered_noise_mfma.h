// Internal interface of the matrix-core variant of FilteredNoise.__call__ (filtered_noise_mfma.hip), used by the
// dispatch in filtered_noise.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddsp {

// True when ddsp_filtered_noise_f32 can run on noise_mfma65_kernel: 65 bands, full window (128 taps), frames of
// 64 c samples that tile N exactly as core.fft_convolve frames it (ddsp/core.py:1440-1450).
bool noise_mfma65_ok(int F, int M, int N, int padding, const void* noise);

int launch_noise_mfma65(const float* magnitudes, const float* noise, float* audio, float* ctl_magnitudes, int B, int F,
                        int N, int start, float initial_bias, int scale, uint64_t seed, uint64_t batch_offset,
                        long long* dbg, hipStream_t st);

}  // namespace ddsp
