// Internal interface of the wavetable backward of Harmonic (harmonic_bwd_table.hip), used by ddsp_harmonic_backward_f32
// (harmonic.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ddsp {

// P / Q (harmonic.hip, "Backward pass") for up to 128 harmonics and any frame size: the adjoint of the wavetable synthesis
bool harm_bwd_table_ok(int F, int K, int N);
// It writes P / Q and returns 0.  Under DDSP_EXP_HARM_BWD=fused (and `amplitudes` .. `grad_hd` given) the frame-rate chain rule
// (harm_bwd_chain_kernel's) runs in the same launch and the function returns 1 (P and Q are not written): measured, slower.
int launch_harm_bwd_table(const float* f0_hz, const double* theta0, const float* grad_audio, float* pq, size_t q_offset, int B,
                          int F, int K, int N, int sample_rate, int amp_linear, hipStream_t st, const float* amplitudes = nullptr,
                          const float* hd = nullptr, float* grad_amp = nullptr, float* grad_hd = nullptr, unsigned flags = 0,
                          int inputs_are_controls = 0);

// the transposed constant factor a K-harmonic backward launch needs, on the current device NOW (ddsp_prepare); 0 on success
int harm_bwd_table_prepare(int K);

}  // namespace ddsp
