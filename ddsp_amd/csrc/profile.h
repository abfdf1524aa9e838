// Per-kernel HIP-event tracing (opt-in; off by default => zero overhead).
// bench.py uses it to time the dominant kernel on the stream it is launched on; the same
// numbers must agree with `rocprofv3 --kernel-trace --stats` (profiles/).
#pragma once
#include <hip/hip_runtime.h>

namespace ddsp {

enum KernelId {
  kHarmControls = 0,
  kHarmSynth,
  kNoiseControls,
  kNoiseIr,
  kTvFir,
  kUniformNoise,
  kAdd,
  kExpSigmoid,
  kHarmFused,
  kNoiseFused,
  kReverbFft,
  kReverbMac,
  kReverbIfft,
  kStftL1,
  kHarmBwdPq,
  kHarmBwdChain,
  kNoiseBwdTaps,
  kNoiseBwdMags,
  kStftL1Bwd,
  kHarmTable,
  kNoiseMfma,
  kHarmBwdTable,
  kNoiseBwdMfma,
  kTvFirMfma,
  kNoiseIrGemm,
  kNumKernels
};

// mask bit i set => kernel i is bracketed by events while tracing is on
void profile_record(int kernel_id, hipStream_t st, bool start);

// Events to attach to the kernel's own dispatch packet (hipExtLaunchKernelGGL start/stop events):
// these carry the kernel's begin/end timestamps, which is what rocprofv3 reports, without the
// command-processor gaps a pair of separate hipEventRecord packets would add.  Both are nullptr
// when tracing is off for this kernel.
void profile_kernel_events(int kernel_id, hipEvent_t* start, hipEvent_t* stop);

struct ProfileScope {
  int id; hipStream_t st;
  ProfileScope(int kernel_id, hipStream_t s) : id(kernel_id), st(s) { profile_record(id, st, true); }
  ~ProfileScope() { profile_record(id, st, false); }
};

}  // namespace ddsp
