// Internal interface of the wavetable variant of Harmonic.__call__ (harmonic_table.hip), used by the
// dispatch in harmonic.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace ddsp {

// True when ddsp_harmonic_f32 can run on harm_table_kernel: raw network outputs in (exp_sigmoid +
// Nyquist-normalised distribution, the defaults of ddsp/synths.py:59-66), the controls dict requested in full
// or not at all, N % F == 0 with a frame size that is a multiple of 64, any K <= 200 (129 .. 200: ten taps, csrc/harmonic_table.hip WIDE).
bool harm_table_ok(int F, int K, int N, const void* hd, const void* ctl_amp, const void* ctl_hd, unsigned flags,
                   int inputs_are_controls);

// add_in: null, or [B,N] samples added to the synthesised audio before it is stored (processors.Add fused in; may be
// `audio` itself)
int launch_harm_table(const float* amplitudes, const float* hd, const float* f0, float* audio, float* ctl_amp,
                      float* ctl_hd, const float* add_in, int B, int F, int K, int N, int sample_rate, unsigned flags,
                      hipStream_t st);

// the constant fragment sets a K-harmonic launch needs, made and copied to the current device NOW (ddsp_prepare: the copy is
// synchronous and must not fall inside a HIP-graph capture); 0 on success
int harm_table_prepare(int K);

}  // namespace ddsp
