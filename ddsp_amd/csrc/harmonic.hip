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
constexpr int kRowsPerWave = 4;       // controls kernel: rows per wavefront
constexpr int kCheb = 16;             // harmonics per Chebyshev block (two exact seeds each)

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------
// kernel 1: controls + phase prefix, ONE launch.
//   blocks [0, n_ctl)        : 4 wavefronts x kRowsPerWave (b,f) rows each; lanes = harmonics.
//                              ws_a layout [B][F+1][Kp]: row F duplicates row F-1 (the
//                              "hold last frame" endpoint of core.resample /
//                              upsample_with_windows), columns K..Kp-1 are zero.
//   blocks [n_ctl, n_ctl+B)  : block b scans f0[b,:] in fp64 and writes
//                              theta0[b][j] = frac( sum_{j'<j} sum_{t in frame j'} f[t] / sr ),
//                              the fundamental's phase (revolutions) at the start of frame j.
//                              Replaces tf.cumsum over time (core.py:955): samples of frame
//                              j carry f[t] = f_j + (f_{j+1}-f_j)*r/hop (legacy bilinear
//                              resize, core.py:613-621, f_F = f_{F-1}), whose sum over the
//                              frame is hop*f_j + (f_{j+1}-f_j)*(hop-1)/2.
// ------------------------------------------------------------------------------------
struct ControlsArgs {
  int B, F, K, Kp, hop, n_ctl_blocks;
  float nyquist, sample_rate;
  unsigned flags;
  int inputs_are_controls;
};

template <int NCHUNK>   // ceil(K/64) <= NCHUNK
__global__ __launch_bounds__(256) void harm_controls_kernel(
    const float* __restrict__ amplitudes, const float* __restrict__ hd,
    const float* __restrict__ f0_hz, float* __restrict__ ctl_amp, float* __restrict__ ctl_hd,
    float* __restrict__ ws_a, double* __restrict__ theta0, ControlsArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int F = p.F, K = p.K, Kp = p.Kp;

  if ((int)blockIdx.x >= p.n_ctl_blocks) {
    // ---------------- phase scan for batch row b ----------------
    if (!theta0) return;
    __shared__ double s_wave[4];
    const int b = blockIdx.x - p.n_ctl_blocks;
    const float* __restrict__ f0 = f0_hz + (size_t)b * F;
    const int per = (F + 255) / 256;
    const int jb = threadIdx.x * per, je = min(jb + per, F);
    const double hop_d = (double)p.hop;
    double local = 0.0;
    for (int j = jb; j < je; ++j) {
      const double fa = (double)f0[j], fb = (double)f0[min(j + 1, F - 1)];
      local += hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
    }
    double incl = local;                       // inclusive scan across the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    double run = base + incl - local;          // exclusive prefix at frame jb
    const double inv_sr = 1.0 / (double)p.sample_rate;
    for (int j = jb; j < je; ++j) {
      const double cyc = run * inv_sr;
      theta0[(size_t)b * F + j] = cyc - floor(cyc);
      const double fa = (double)f0[j], fb = (double)f0[min(j + 1, F - 1)];
      run += hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
    }
    return;
  }

  // ---------------- controls for kRowsPerWave rows per wavefront ----------------
  const bool scale = (p.flags & DDSP_HARM_SCALE_EXP_SIGMOID) && !p.inputs_are_controls;
  const bool normalize = (p.flags & DDSP_HARM_NORMALIZE_NYQUIST) && !p.inputs_are_controls;
  const float kLog10 = 2.302585092994046f;   // fl32(log(10.0)), tf.math.log(exponent)
  const long rows = (long)p.B * F;
  const long row0 = ((long)blockIdx.x * 4 + wave) * kRowsPerWave;

  float v[kRowsPerWave][NCHUNK];
  float amp[kRowsPerWave], f0v[kRowsPerWave];
#pragma unroll
  for (int q = 0; q < kRowsPerWave; ++q) {     // all loads first: independent, in flight together
    const long row = min(row0 + q, rows - 1);
    amp[q] = amplitudes[row];
    f0v[q] = f0_hz[row];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const int k = c * 64 + lane;
      v[q][c] = (k < K) ? hd[row * K + k] : 0.0f;
    }
  }
#pragma unroll
  for (int q = 0; q < kRowsPerWave; ++q) {
    const long row = row0 + q;
    if (row >= rows) break;
    const int f = (int)(row % F);
    const long b = row / F;
    float a = amp[q];
    if (scale) a = exp_sigmoid(a, kLog10, 2.0f, 1e-7f);
    float part = 0.0f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const int k = c * 64 + lane;
      float x = v[q][c];
      if (k < K) {
        if (scale) x = exp_sigmoid(x, kLog10, 2.0f, 1e-7f);
        // core.remove_above_nyquist on f0 * [1..K]  (core.py:899-903, 1028-1045)
        if (normalize && (f0v[q] * (float)(k + 1) >= p.nyquist)) x = 0.0f;
      } else {
        x = 0.0f;
      }
      v[q][c] = x;
      part += x;
    }
    float inv = 1.0f;
    if (!p.inputs_are_controls) {
      // core.safe_divide(hd, reduce_sum(hd))  (core.py:905-907, 207-210)
      float den = wave_sum(part);
      if (den == 0.0f) den = 1e-7f;
      inv = 1.0f / den;
    }
    float* wa = ws_a ? ws_a + ((size_t)b * (F + 1) + f) * Kp : nullptr;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const int k = c * 64 + lane;
      const float h = p.inputs_are_controls ? v[q][c] : v[q][c] * inv;   // == v/den up to 1 ulp
      if (k < K && ctl_hd) ctl_hd[row * K + k] = h;
      if (wa && k < Kp) {
        const float av = (k < K) ? a * h : 0.0f;   // core.py:1097 amplitudes * distribution
        wa[k] = av;
        if (f == F - 1) wa[Kp + k] = av;
      }
    }
    if (lane == 0 && ctl_amp) ctl_amp[row] = a;
  }
}

// ------------------------------------------------------------------------------------
// kernel 2: synthesis.  grid = (ceil(F/kFramesPerBlock), B), 256 threads, no LDS, no
// barriers.  One wavefront = 64 consecutive samples; lanes = samples.
//   sin(2 pi k theta), k = 1..K, comes from blocks of kCheb harmonics: two seeds per block
//   by v_sin_f32 on the EXACT fractional phase fma(k, theta, -rint(k*theta)), the rest by the
//   Chebyshev recurrence s_{k+1} = 2cos(2 pi theta) s_k - s_{k-1} (max error 8e-6 per
//   harmonic over a 16-block in fp32; tools/cheb_error.py).  1 FMA (recurrence) + 2 FMA
//   (the two frames' amplitudes, wave-uniform SGPR operands) per harmonic and sample.
// ------------------------------------------------------------------------------------
struct SynthArgs {
  int F, K, Kp, N, hop;
  float sample_rate, nyquist;
  int amp_linear;
};

// exact fractional part of k*theta (theta in [0,1], k < 2^23): in [-0.5, 0.5]
__device__ __forceinline__ float frac_phase(float theta, float kf) {
  return fmaf(theta, kf, -rintf(theta * kf));
}

template <bool GUARD>
__device__ __forceinline__ void cheb_block(const float* __restrict__ a0p,
                                           const float* __restrict__ a1p, int k0, int n_valid,
                                           float theta, float c2, float& acc0, float& acc1) {
  float a0[kCheb], a1[kCheb];
#pragma unroll
  for (int u = 0; u < kCheb; ++u) {        // wave-uniform addresses: scalar loads
    a0[u] = a0p[k0 + u];
    a1[u] = a1p[k0 + u];
    if (GUARD && u >= n_valid) { a0[u] = 0.0f; a1[u] = 0.0f; }
  }
  float s0 = sin_rev(frac_phase(theta, (float)(k0 + 1)));
  float s1 = sin_rev(frac_phase(theta, (float)(k0 + 2)));
  acc0 = fmaf(a0[0], s0, acc0); acc1 = fmaf(a1[0], s0, acc1);
  acc0 = fmaf(a0[1], s1, acc0); acc1 = fmaf(a1[1], s1, acc1);
#pragma unroll
  for (int u = 2; u < kCheb; ++u) {
    const float s2 = fmaf(c2, s1, -s0);
    acc0 = fmaf(a0[u], s2, acc0);
    acc1 = fmaf(a1[u], s2, acc1);
    s0 = s1; s1 = s2;
  }
}

template <bool UNIFORM>
__global__ __launch_bounds__(kSynthThreads) void harm_synth_kernel(
    const float* __restrict__ f0_all /*[B,F]*/, const float* __restrict__ ws_a /*[B,F+1,Kp]*/,
    const double* __restrict__ theta0 /*[B,F]*/, float* __restrict__ audio /*[B,N]*/,
    SynthArgs p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int j0 = blockIdx.x * kFramesPerBlock;
  const int nfr = min(kFramesPerBlock, p.F - j0);
  const float* __restrict__ f0 = f0_all + (size_t)b * p.F;
  const double* __restrict__ th0 = theta0 + (size_t)b * p.F;
  const double inv_sr = 1.0 / (double)p.sample_rate;
  const double inv_2hop = 0.5 / (double)p.hop;

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

    // fundamental phase in revolutions, inclusive cumsum: theta0_j + sum_{r'<=r} f[r'] / sr
    const double rr = (double)r;
    const double wj = (double)fj * inv_sr, dw = ((double)fj1 - (double)fj) * inv_sr * inv_2hop;
    const double cyc = th0[j] + (rr + 1.0) * (wj + dw * rr);
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
    if (UNIFORM) {
      kA = __builtin_amdgcn_readfirstlane(kA);
      const float c2 = 2.0f * __builtin_amdgcn_cosf(theta);   // 2 cos(2 pi theta)
      for (; k + kCheb <= kA; k += kCheb)
        cheb_block<false>(a0p, a1p, k, kCheb, theta, c2, acc0, acc1);
      if (k < kA) {                                  // Kp is a multiple of kCheb: reads stay in-row
        cheb_block<true>(a0p, a1p, k, kA - k, theta, c2, acc0, acc1);
        k = kA;
      }
    } else {
      for (; k < kA; ++k) {
        const float s = sin_rev(frac_phase(theta, (float)(k + 1)));
        acc0 = fmaf(a0p[k], s, acc0);
        acc1 = fmaf(a1p[k], s, acc1);
      }
    }
    // harmonics that cross Nyquist inside this tile: audio-rate mask on the interpolated
    // frequency, same fp32 op order as TF (core.py:942-944 on top + (bottom-top)*lerp).
    for (; k < kN; ++k) {
      const float kf = (float)(k + 1);
      const float top = fj * kf, bot = fj1 * kf;
      const float fk = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), lerp));
      const float s = (fk >= p.nyquist) ? 0.0f : sin_rev(frac_phase(theta, kf));
      acc0 = fmaf(a0p[k], s, acc0);
      acc1 = fmaf(a1p[k], s, acc1);
    }

    // ---- frame-rate -> audio-rate amplitude envelope weights ----------------------------
    float w_next;   // weight of frame j+1
    if (p.amp_linear) {
      w_next = lerp;                                            // core.resample 'linear'
    } else {
      w_next = 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp); // periodic Hann(2*hop)[r]
    }
    const float w_cur = 1.0f - w_next;                          // Hann(2*hop)[hop + r]
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

// workspace = [ theta0: B*F doubles ][ ws_a: B*(F+1)*Kp floats ]
static inline size_t theta_bytes(int B, int F) { return ((size_t)B * F * sizeof(double) + 63) & ~(size_t)63; }
extern "C" size_t ddsp_harmonic_workspace_bytes(int B, int F, int K, int N) {
  (void)N;
  if (B <= 0 || F <= 0 || K <= 0) return 0;
  return theta_bytes(B, F) + (size_t)B * (size_t)(F + 1) * (size_t)round_up(K, 16) * sizeof(float);
}

static int launch_controls(const float* amps, const float* hd, const float* f0, float* ctl_amp,
                           float* ctl_hd, void* workspace, int B, int F, int K, int N,
                           int sample_rate, unsigned flags, int inputs_are_controls,
                           hipStream_t st) {
  ControlsArgs p;
  p.B = B; p.F = F; p.K = K; p.Kp = round_up(K, 16);
  p.hop = workspace ? N / F : 1;
  const long rows = (long)B * F;
  p.n_ctl_blocks = (int)((rows + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave));
  p.nyquist = (float)(sample_rate / 2.0);
  p.sample_rate = (float)sample_rate;
  p.flags = flags;
  p.inputs_are_controls = inputs_are_controls;
  double* theta0 = (double*)workspace;
  float* ws_a = workspace ? (float*)((char*)workspace + theta_bytes(B, F)) : nullptr;
  const dim3 grid((unsigned)(p.n_ctl_blocks + (workspace ? B : 0))), block(256);
  const int nchunk = (K + 63) / 64;
  ProfileScope prof(kHarmControls, st);
#define DDSP_LAUNCH_CTL(NC)                                                                  \
  hipLaunchKernelGGL((harm_controls_kernel<NC>), grid, block, 0, st, amps, hd, f0, ctl_amp, \
                     ctl_hd, ws_a, theta0, p)
  if (nchunk <= 1) DDSP_LAUNCH_CTL(1);
  else if (nchunk <= 2) DDSP_LAUNCH_CTL(2);
  else if (nchunk <= 4) DDSP_LAUNCH_CTL(4);
  else if (nchunk <= 8) DDSP_LAUNCH_CTL(8);
  else return DDSP_ERR_UNSUPPORTED;      // K > 512 harmonics
#undef DDSP_LAUNCH_CTL
  return check_launch();
}

static int launch_synth(const float* f0, const void* workspace, float* audio, int B, int F, int K,
                        int N, int sample_rate, unsigned flags, hipStream_t st) {
  SynthArgs p;
  p.F = F; p.K = K; p.Kp = round_up(K, 16); p.N = N; p.hop = N / F;
  p.sample_rate = (float)sample_rate;
  p.nyquist = (float)(sample_rate / 2.0);
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  const double* theta0 = (const double*)workspace;
  const float* ws_a = (const float*)((const char*)workspace + theta_bytes(B, F));
  const dim3 grid((unsigned)((F + kFramesPerBlock - 1) / kFramesPerBlock), (unsigned)B);
  const dim3 block(kSynthThreads);
  ProfileScope prof(kHarmSynth, st);
  if ((p.hop % 64) == 0)
    hipLaunchKernelGGL((harm_synth_kernel<true>), grid, block, 0, st, f0, ws_a, theta0, audio, p);
  else
    hipLaunchKernelGGL((harm_synth_kernel<false>), grid, block, 0, st, f0, ws_a, theta0, audio, p);
  return check_launch();
}

static int check_harmonic_shape(int B, int F, int K, int N, int sample_rate) {
  if (B <= 0 || F <= 0 || K <= 0 || N <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (B > 65535) return DDSP_ERR_UNSUPPORTED;   // grid.y limit; shard the batch instead
  if (N % F != 0) return DDSP_ERR_UNSUPPORTED;
  if (K > 512) return DDSP_ERR_UNSUPPORTED;
  return DDSP_OK;
}

extern "C" int ddsp_harmonic_controls_f32(const float* amplitudes, const float* hd,
                                          const float* f0_hz, float* ctl_amp, float* ctl_hd,
                                          int B, int F, int K, int sample_rate, unsigned flags,
                                          void* stream) {
  if (!amplitudes || !hd || !f0_hz || !ctl_amp || !ctl_hd) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (K > 512) return DDSP_ERR_UNSUPPORTED;
  return launch_controls(amplitudes, hd, f0_hz, ctl_amp, ctl_hd, /*workspace=*/nullptr, B, F, K,
                         /*N=*/F, sample_rate, flags, /*inputs_are_controls=*/0,
                         (hipStream_t)stream);
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
  rc = launch_controls(ctl_amp, ctl_hd, f0_hz, nullptr, nullptr, workspace, B, F, K, N,
                       sample_rate, flags, /*inputs_are_controls=*/1, st);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, workspace, audio, B, F, K, N, sample_rate, flags, st);
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
  rc = launch_controls(amplitudes, hd, f0_hz, ctl_amp, ctl_hd, workspace, B, F, K, N,
                       sample_rate, flags, /*inputs_are_controls=*/0, st);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, workspace, audio, B, F, K, N, sample_rate, flags, st);
}
