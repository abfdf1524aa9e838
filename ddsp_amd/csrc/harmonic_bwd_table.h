// Internal interface of the wavetable backward of Harmonic (harmonic_bwd_table.hip), used by ddsp_harmonic_backward_f32
// (harmonic.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ddsp {

// P / Q (harmonic.hip, "Backward pass") for up to 128 harmonics and any frame size: the adjoint of the wavetable synthesis
bool harm_bwd_table_ok(int F, int K, int N);
int launch_harm_bwd_table(const float* f0_hz, const double* theta0, const float* grad_audio, float* pq, size_t q_offset, int B,
                          int F, int K, int N, int sample_rate, int amp_linear, hipStream_t st);

}  // namespace ddsp
