// Harmonic synthesiser kernels for gfx950 (MI355X).  Replaces, as fused closed forms,
// the TF op chain of ddsp/synths.py:94-146 -> ddsp/core.py:894-907 (normalize_harmonics),
// :1048-1111 (harmonic_synthesis), :573-714 (resample / upsample_with_windows) and
// :912-962 (oscillator_bank).  See DESIGN.md "Harmonic" for the derivation.
//
//   kernel 1  harm_controls_kernel : one wavefront per (batch, frame); lanes = harmonics
//             (K-contiguous coalesced reads of harmonic_distribution), exp_sigmoid, frame-
//             rate Nyquist mask, wave-shuffle sum over K, normalise; writes the per-
//             frame harmonic amplitudes a[b,j,k] = amp*hd_k to the workspace (and the
//             controls dict when asked for).
//   kernel 2  harm_synth_kernel    : one wavefront per 64 consecutive samples of one
//             frame; lanes = samples.  Phase of the fundamental is a closed form
//             (fp64 prefix over frames + quadratic inside the frame, wrapped to [0,1)
//             revolutions); harmonic k is sin(2*pi*k*theta) by v_sin_f32 (revolutions
//             in, no range reduction needed below 256 harmonics).  The two frames'
//             amplitudes are wave-uniform -> scalar (SGPR) loads, zero LDS traffic in
//             the k loop; the Hann / linear interpolation weights are applied once per
//             sample after the loop.
#include "common.h"
#include "profile.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {

constexpr int kFramesPerBlock = 16;   // frames handled by one synth block
constexpr int kSynthThreads = 256;

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------
// kernel 1: controls.  grid = ceil(B*F/4) blocks of 4 wavefronts, wave = one (b,f) row.
// ws_a layout: [B][F+1][Kp], row F duplicates row F-1 (the "hold last frame" endpoint of
// core.resample / upsample_with_windows), columns K..Kp-1 are zero.
// ------------------------------------------------------------------------------------
template <int NCHUNK>   // ceil(K/64) <= NCHUNK
__global__ __launch_bounds__(256) void harm_controls_kernel(
    const float* __restrict__ amplitudes, const float* __restrict__ hd,
    const float* __restrict__ f0_hz, float* __restrict__ ctl_amp, float* __restrict__ ctl_hd,
    float* __restrict__ ws_a, int B, int F, int K, int Kp, float nyquist, unsigned flags,
    int inputs_are_controls) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * F) return;
  const int f = (int)(row % F);
  const long b = row / F;
  const bool scale = (flags & DDSP_HARM_SCALE_EXP_SIGMOID) && !inputs_are_controls;
  const bool normalize = (flags & DDSP_HARM_NORMALIZE_NYQUIST) && !inputs_are_controls;
  const float kLog10 = 2.302585092994046f;   // fl32(log(10.0)), tf.math.log(exponent)

  float amp = amplitudes[row];
  const float f0 = f0_hz[row];
  if (scale) amp = exp_sigmoid(amp, kLog10, 2.0f, 1e-7f);

  float v[NCHUNK];
  float part = 0.0f;
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int k = c * 64 + lane;
    float x = 0.0f;
    if (k < K) {
      x = hd[row * K + k];
      if (scale) x = exp_sigmoid(x, kLog10, 2.0f, 1e-7f);
      // core.remove_above_nyquist on f0 * [1..K]  (core.py:899-903, 1028-1045)
      if (normalize && (f0 * (float)(k + 1) >= nyquist)) x = 0.0f;
    }
    v[c] = x;
    part += x;
  }
  float inv = 1.0f;
  if (!inputs_are_controls) {
    // core.safe_divide(hd, reduce_sum(hd))  (core.py:905-907, 207-210)
    float den = wave_sum(part);
    if (den == 0.0f) den = 1e-7f;
    inv = 1.0f / den;
  }
  float* wa = ws_a ? ws_a + ((size_t)b * (F + 1) + f) * Kp : nullptr;
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int k = c * 64 + lane;
    const float h = inputs_are_controls ? v[c] : v[c] * inv;   // == v/den up to 1 ulp
    if (k < K && ctl_hd) ctl_hd[row * K + k] = h;
    if (wa && k < Kp) {
      const float a = (k < K) ? amp * h : 0.0f;   // core.py:1097 amplitudes * distribution
      wa[k] = a;
      if (f == F - 1) wa[Kp + k] = a;
    }
  }
  if (lane == 0 && ctl_amp) ctl_amp[row] = amp;
}

// ------------------------------------------------------------------------------------
// kernel 2: synthesis.  grid = (ceil(F/kFramesPerBlock), B), 256 threads.
// ------------------------------------------------------------------------------------
struct SynthArgs {
  int F, K, Kp, N, hop;
  float sample_rate, nyquist;
  int amp_linear;
};

template <bool UNIFORM, bool FRACT>
__global__ __launch_bounds__(kSynthThreads) void harm_synth_kernel(
    const float* __restrict__ f0_all /*[B,F]*/, const float* __restrict__ ws_a /*[B,F+1,Kp]*/,
    float* __restrict__ audio /*[B,N]*/, SynthArgs p) {
  __shared__ double s_red[kSynthThreads / 64];
  __shared__ double s_theta[kFramesPerBlock + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int j0 = blockIdx.x * kFramesPerBlock;
  const int nfr = min(kFramesPerBlock, p.F - j0);
  const float* __restrict__ f0 = f0_all + (size_t)b * p.F;
  const double hop_d = (double)p.hop;

  // ---- phase prefix over frames, in fp64 (replaces tf.cumsum over time, core.py:955).
  // Samples of frame j carry f[t] = f_j + (f_{j+1}-f_j)*r/hop (legacy bilinear resize,
  // core.py:613-621; f_F = f_{F-1}); their sum is hop*f_j + (f_{j+1}-f_j)*(hop-1)/2.
  auto frame_inc = [&](int j) -> double {
    const double fa = (double)f0[j];
    const double fb = (double)f0[min(j + 1, p.F - 1)];
    return hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
  };
  double part = 0.0;
  for (int j = tid; j < j0; j += kSynthThreads) part += frame_inc(j);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  if (tid == 0) {
    double acc = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const double inv_sr = 1.0 / (double)p.sample_rate;
    for (int q = 0; q <= nfr; ++q) {
      const double cyc = acc * inv_sr;
      s_theta[q] = cyc - floor(cyc);       // revolutions at the start of frame j0+q
      if (q < nfr) acc += frame_inc(j0 + q);
    }
  }
  __syncthreads();

  const int tiles_per_frame = (p.hop + 63) / 64;        // UNIFORM: hop % 64 == 0
  const int n_tiles = UNIFORM ? nfr * tiles_per_frame : (nfr * p.hop + 63) / 64;
  const float inv_hop = 1.0f / (float)p.hop;

  for (int tile = wave; tile < n_tiles; tile += kSynthThreads / 64) {
    // ---- which frame / offset does each lane synthesise -----------------------------
    int q, r;
    if (UNIFORM) {
      const int tu = __builtin_amdgcn_readfirstlane(tile);
      q = tu / tiles_per_frame;                 // wave-uniform
      r = (tu - q * tiles_per_frame) * 64 + lane;
    } else {
      const int s = tile * 64 + lane;           // sample index inside the block's range
      q = min(s / p.hop, nfr - 1);
      r = s - q * p.hop;                        // lanes past the range get r >= hop
    }
    const int j = j0 + q;
    const bool active = UNIFORM ? true : (r < p.hop);
    const float fj = f0[j];
    const float fj1 = f0[min(j + 1, p.F - 1)];

    // fundamental phase in revolutions, inclusive cumsum: sum_{r'<=r} f[r'] / sr
    const double rr = (double)r;
    const double cyc = s_theta[q] + ((rr + 1.0) * (double)fj +
                                     ((double)fj1 - (double)fj) * rr * (rr + 1.0) /
                                         (2.0 * hop_d)) / (double)p.sample_rate;
    const float theta = (float)(cyc - floor(cyc));

    const float lerp = (float)r * inv_hop;      // == TF's pos - floor(pos) (exact for 2^n hops)

    // ---- harmonic ranges: [0,kA) never above Nyquist, [kA,kN) decided per sample ------
    float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
    if (!UNIFORM) { fmx = wave_max(active ? fmx : 0.0f); fmn = wave_min(active ? fmn : 3.0e38f); }
    int kA = p.K, kN = p.K;
    if (fmx > 0.0f) kA = (int)fminf((float)p.K, floorf(p.nyquist * (1.0f - 2e-6f) / fmx));
    if (fmn > 0.0f) kN = (int)fminf((float)p.K, floorf(p.nyquist * (1.0f + 2e-6f) / fmn));
    kA = max(min(kA, kN), 0);

    const float* __restrict__ a0p = ws_a + ((size_t)b * (p.F + 1) + j) * p.Kp;
    const float* __restrict__ a1p = a0p + p.Kp;
    float acc0 = 0.0f, acc1 = 0.0f;
    int k = 0;
    // main loop, 4 harmonics per trip (Kp is a multiple of 16 and zero-padded, kA<=K)
    for (; k + 4 <= kA; k += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float ph = theta * (float)(k + u + 1);
        if (FRACT) ph = __builtin_amdgcn_fractf(ph);
        const float s = sin_rev(ph);
        acc0 = fmaf(a0p[k + u], s, acc0);
        acc1 = fmaf(a1p[k + u], s, acc1);
      }
    }
    for (; k < kA; ++k) {
      float ph = theta * (float)(k + 1);
      if (FRACT) ph = __builtin_amdgcn_fractf(ph);
      const float s = sin_rev(ph);
      acc0 = fmaf(a0p[k], s, acc0);
      acc1 = fmaf(a1p[k], s, acc1);
    }
    // harmonics that cross Nyquist inside this tile: audio-rate mask on the interpolated
    // frequency, same fp32 op order as TF (core.py:942-944 on top + (bottom-top)*lerp).
    for (; k < kN; ++k) {
      const float kf = (float)(k + 1);
      const float top = fj * kf, bot = fj1 * kf;
      const float fk = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), lerp));
      float ph = theta * kf;
      if (FRACT) ph = __builtin_amdgcn_fractf(ph);
      const float s = (fk >= p.nyquist) ? 0.0f : sin_rev(ph);
      acc0 = fmaf(a0p[k], s, acc0);
      acc1 = fmaf(a1p[k], s, acc1);
    }

    // ---- frame-rate -> audio-rate amplitude envelope weights ----------------------------
    float w_next;   // weight of frame j+1
    if (p.amp_linear) {
      w_next = lerp;                               // core.resample 'linear'
    } else {
      w_next = 0.5f - 0.5f * cospif(lerp);         // periodic Hann(2*hop)[r]  (core.py:696-698)
    }
    const float w_cur = 1.0f - w_next;             // Hann(2*hop)[hop + r]
    if (active) {
      const long t = (long)j * p.hop + r;
      audio[(size_t)b * p.N + t] = w_cur * acc0 + w_next * acc1;
    }
  }
}

}  // namespace ddsp

// =====================================================================================
// C ABI
// =====================================================================================
using namespace ddsp;

static inline int check_launch() { return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH; }

extern "C" size_t ddsp_harmonic_workspace_bytes(int B, int F, int K, int N) {
  (void)N;
  if (B <= 0 || F <= 0 || K <= 0) return 0;
  return (size_t)B * (size_t)(F + 1) * (size_t)round_up(K, 16) * sizeof(float);
}

static int launch_controls(const float* amps, const float* hd, const float* f0, float* ctl_amp,
                           float* ctl_hd, float* ws_a, int B, int F, int K, int sample_rate,
                           unsigned flags, int inputs_are_controls, hipStream_t st) {
  const int Kp = round_up(K, 16);
  const long rows = (long)B * F;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const float nyq = (float)(sample_rate / 2.0);
  const int nchunk = (K + 63) / 64;
  ProfileScope prof(kHarmControls, st);
#define DDSP_LAUNCH_CTL(NC)                                                                  \
  hipLaunchKernelGGL((harm_controls_kernel<NC>), grid, block, 0, st, amps, hd, f0, ctl_amp, \
                     ctl_hd, ws_a, B, F, K, Kp, nyq, flags, inputs_are_controls)
  if (nchunk <= 1) DDSP_LAUNCH_CTL(1);
  else if (nchunk <= 2) DDSP_LAUNCH_CTL(2);
  else if (nchunk <= 4) DDSP_LAUNCH_CTL(4);
  else if (nchunk <= 8) DDSP_LAUNCH_CTL(8);
  else if (nchunk <= 16) DDSP_LAUNCH_CTL(16);
  else return DDSP_ERR_UNSUPPORTED;      // K > 1024 harmonics
#undef DDSP_LAUNCH_CTL
  return check_launch();
}

static int launch_synth(const float* f0, const float* ws_a, float* audio, int B, int F, int K,
                        int N, int sample_rate, unsigned flags, hipStream_t st) {
  SynthArgs p;
  p.F = F; p.K = K; p.Kp = round_up(K, 16); p.N = N; p.hop = N / F;
  p.sample_rate = (float)sample_rate;
  p.nyquist = (float)(sample_rate / 2.0);
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  const dim3 grid((unsigned)((F + kFramesPerBlock - 1) / kFramesPerBlock), (unsigned)B);
  const dim3 block(kSynthThreads);
  ProfileScope prof(kHarmSynth, st);
  const bool uniform = (p.hop % 64) == 0;
  const bool fract = K > 255;          // v_sin_f32 is specified for |x| <= 256 revolutions
  if (uniform && !fract) hipLaunchKernelGGL((harm_synth_kernel<true, false>), grid, block, 0, st, f0, ws_a, audio, p);
  else if (uniform) hipLaunchKernelGGL((harm_synth_kernel<true, true>), grid, block, 0, st, f0, ws_a, audio, p);
  else if (!fract) hipLaunchKernelGGL((harm_synth_kernel<false, false>), grid, block, 0, st, f0, ws_a, audio, p);
  else hipLaunchKernelGGL((harm_synth_kernel<false, true>), grid, block, 0, st, f0, ws_a, audio, p);
  return check_launch();
}

static int check_harmonic_shape(int B, int F, int K, int N, int sample_rate) {
  if (B <= 0 || F <= 0 || K <= 0 || N <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (B > 65535) return DDSP_ERR_UNSUPPORTED;   // grid.y limit; shard the batch instead
  if (N % F != 0) return DDSP_ERR_UNSUPPORTED;
  return DDSP_OK;
}

extern "C" int ddsp_harmonic_controls_f32(const float* amplitudes, const float* hd,
                                          const float* f0_hz, float* ctl_amp, float* ctl_hd,
                                          int B, int F, int K, int sample_rate, unsigned flags,
                                          void* stream) {
  if (!amplitudes || !hd || !f0_hz || !ctl_amp || !ctl_hd) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  return launch_controls(amplitudes, hd, f0_hz, ctl_amp, ctl_hd, /*ws_a=*/nullptr, B, F, K,
                         sample_rate, flags, /*inputs_are_controls=*/0, (hipStream_t)stream);
}

extern "C" int ddsp_harmonic_signal_f32(const float* ctl_amp, const float* ctl_hd,
                                        const float* f0_hz, float* audio, void* workspace,
                                        size_t workspace_bytes, int B, int F, int K, int N,
                                        int sample_rate, unsigned flags, void* stream) {
  if (!ctl_amp || !ctl_hd || !f0_hz || !audio || !workspace) return DDSP_ERR_NULL_POINTER;
  int rc = check_harmonic_shape(B, F, K, N, sample_rate);
  if (rc != DDSP_OK) return rc;
  if (workspace_bytes < ddsp_harmonic_workspace_bytes(B, F, K, N) || ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  rc = launch_controls(ctl_amp, ctl_hd, f0_hz, nullptr, nullptr, (float*)workspace, B, F, K,
                       sample_rate, flags, /*inputs_are_controls=*/1, st);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, (const float*)workspace, audio, B, F, K, N, sample_rate, flags, st);
}

extern "C" int ddsp_harmonic_f32(const float* amplitudes, const float* hd, const float* f0_hz,
                                 float* audio, float* ctl_amp, float* ctl_hd, void* workspace,
                                 size_t workspace_bytes, int B, int F, int K, int N,
                                 int sample_rate, unsigned flags, void* stream) {
  if (!amplitudes || !hd || !f0_hz || !audio || !workspace) return DDSP_ERR_NULL_POINTER;
  int rc = check_harmonic_shape(B, F, K, N, sample_rate);
  if (rc != DDSP_OK) return rc;
  if (workspace_bytes < ddsp_harmonic_workspace_bytes(B, F, K, N) || ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  rc = launch_controls(amplitudes, hd, f0_hz, ctl_amp, ctl_hd, (float*)workspace, B, F, K,
                       sample_rate, flags, /*inputs_are_controls=*/0, st);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, (const float*)workspace, audio, B, F, K, N, sample_rate, flags, st);
}
