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
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <hip/hip_ext.h>
#include "common.h"
#include "profile.h"
#include "harmonic_bwd_table.h"
#include "harmonic_bwd_chain.h"
#include "harmonic_table.h"
#include "filtered_noise_general.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {

constexpr int kFramesPerBlock = 8;    // frames per block of the generic synthesis kernel
constexpr int kMaxUnitFrames = 16;    // the fused kernel's units are 8 or 16 frames (template parameter FPB)
constexpr int kSynthThreads = 256;
constexpr int kRowsPerWave = 4;       // controls kernel: rows per wavefront
constexpr int kCheb = 16;             // harmonics per Chebyshev block (two exact seeds each)
constexpr int kMaxHarmonics = 2048;   // harm_controls_kernel / harm_controls_bwd_kernel hold a row as ceil(K / 64) values per lane: 32 at most

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
  // streaming synthesis (core.py:966-1025): phase carried in / out, radians, one value per row
  const float* initial_phase;   // [B] or null
  float* final_phase;           // [B] or null
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
    // a thread's frames (and the one behind them) requested together where they fit eight registers - up to 2048 frames per
    // row; taken one at a time the scan was a chain of memory round trips, most of this launch's 6 us
    constexpr int kCache = 8;
    float fc[kCache + 1];
    const bool cached = per <= kCache;
    if (cached) {
#pragma unroll
      for (int u = 0; u <= kCache; ++u) fc[u] = f0[min(jb + u, F - 1)];
#pragma unroll
      for (int u = 0; u < kCache; ++u) {
        if (jb + u < je) {
          const double fa = (double)fc[u], fb = (double)fc[u + 1];
          local += hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
        }
      }
    } else {
      for (int j = jb; j < je; ++j) {
        const double fa = (double)f0[j], fb = (double)f0[min(j + 1, F - 1)];
        local += hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
      }
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
    const double kTwoPi = 6.283185307179586;
    const double init_rev = p.initial_phase ? (double)p.initial_phase[b] / kTwoPi : 0.0;
    if (cached) {
#pragma unroll
      for (int u = 0; u < kCache; ++u) {
        if (jb + u < je) {
          const double cyc = run * inv_sr + init_rev;
          theta0[(size_t)b * F + jb + u] = cyc - floor(cyc);
          const double fa = (double)fc[u], fb = (double)fc[u + 1];
          run += hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
        }
      }
    } else {
      for (int j = jb; j < je; ++j) {
        const double cyc = run * inv_sr + init_rev;
        theta0[(size_t)b * F + j] = cyc - floor(cyc);
        const double fa = (double)f0[j], fb = (double)f0[min(j + 1, F - 1)];
        run += hop_d * fa + (fb - fa) * (hop_d - 1.0) * 0.5;
      }
    }
    // harmonic_oscillator_bank's final_phase (core.py:1008-1012, angular cumsum):
    // (sum of all omega mod 2 pi) + initial_phase - the fundamental's phase at the last sample
    if (p.final_phase && jb < F && je == F) {
      const double cyc = run * inv_sr;
      p.final_phase[b] = (float)((cyc - floor(cyc)) * kTwoPi + (p.initial_phase ? (double)p.initial_phase[b] : 0.0));
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
  int no_audio_mask;       // harmonic_oscillator_bank has no audio-rate Nyquist mask (core.py:966-1025)
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
    if (p.no_audio_mask) kA = kN = p.K;

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
      const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp));
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


// ------------------------------------------------------------------------------------
// kernel 3 (the fast path, hop % 64 == 0): controls + phase prefix + synthesis in ONE launch.
// Persistent blocks; a block takes "units" of kFramesPerBlock frames of one batch row.
//   phase A  the unit's 17 amplitude rows (16 frames + the next frame, which the last frame
//            interpolates towards) are computed as in kernel 1 and written to the BLOCK's
//            private 7.6 KB workspace slot (re-used for every unit, so it lives in this XCD's
//            L2 and never streams to HBM); an fp64 block reduction over the row's earlier
//            frames gives the fundamental's phase at the start of each frame; one lane per
//            frame precomputes everything that is per-frame (phase, slope, live-harmonic counts).
//   phase B  after s_waitcnt vmcnt(0) + barrier the rows sit in L2; the scalar cache is
//            invalidated and every wavefront pulls the two rows of its tile with
//            s_load_dwordx16 (SGPR operands for the 2 FMAs per harmonic, no LDS traffic).
//            The loads are inline asm: the compiler will not select scalar loads for memory
//            written earlier in the same kernel.
//            sin(2 pi h theta): super-blocks of 32 harmonics, four exact seeds, then the
//            stride-2 recurrence s[h] = 2cos(4 pi theta) s[h-2] - s[h-4] - two independent
//            chains (odd / even harmonics), which doubles the dependency distance of the FMA
//            stream (a single chain issues at ~2.9 cycles/instruction, tools/microbench2).
// ------------------------------------------------------------------------------------
typedef float sgpr8 __attribute__((ext_vector_type(8)));

// both loads and their wait in one statement (cdna_hip_programming.md 5.7, form (i))
__device__ __forceinline__ void sload_rows(sgpr8& a0, sgpr8& a1, const float* p0, const float* p1) {
  asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a0), "=&s"(a1) : "s"(p0), "s"(p1) : "memory");
}

typedef float f2 __attribute__((ext_vector_type(2)));

// The synthesis loop keeps PAIRS of adjacent harmonics {h, h+1} in 64-bit VGPR pairs: the two rows'
// amplitudes of a pair are adjacent SGPRs, so the two accumulates of two harmonics (4 FMAs with an
// SGPR operand, ~1.9 ns each) become two v_pk_fma_f32 with an SGPR-pair operand (~2.9 ns each):
// tools/microbench4 measures 1.37 vs 1.66 ns per FMA-equivalent for the loop.  The stride-2
// recurrence s[h] = 2cos(4 pi theta) s[h-2] - s[h-4] maps onto pairs directly:
// P_m = c4 * P_{m-1} - P_{m-2}; accA = {row-0 odd h, row-0 even h}, accB likewise for row 1.
struct PairState { f2 older, newer; };

__device__ __forceinline__ f2 next_pair(PairState& st, float c4) {
  const f2 sn = __builtin_elementwise_fma((f2){c4, c4}, st.newer, -st.older);   // one v_pk_fma_f32
  st.older = st.newer; st.newer = sn;
  return sn;
}
// the second pair of seeds {h+2, h+3} from the exact pair {h, h+1} by two steps of the stride-1
// recurrence s[h+1] = 2cos(2 pi theta) s[h] - s[h-1]: two FMAs instead of two more v_sin_f32
__device__ __forceinline__ f2 seed_pair_next(PairState& st, float c2) {
  f2 sn;
  sn.x = fmaf(c2, st.newer.y, -st.newer.x);
  sn.y = fmaf(c2, sn.x, -st.newer.y);
  st.older = st.newer; st.newer = sn;
  return sn;
}
__device__ __forceinline__ f2 seed_pair(PairState& st, float theta, int h /* first harmonic, 1-based */) {
  f2 sn;
  sn.x = sin_rev(frac_phase(theta, (float)h));
  sn.y = sin_rev(frac_phase(theta, (float)(h + 1)));
  st.older = st.newer; st.newer = sn;
  return sn;
}

// 8 harmonics k+1 .. k+8 in two groups of 4; only the first n4 groups are live (wave-uniform).
template <bool SEEDS>
__device__ __forceinline__ void harm_oct(const float* p0, const float* p1, int k, int n4,
                                         float theta, float c4, float c2, PairState& st, f2& accA, f2& accB) {
  sgpr8 a0, a1;
  sload_rows(a0, a1, p0 + k, p1 + k);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g < n4) {
#pragma unroll
      for (int i = 4 * g; i < 4 * g + 4; i += 2) {
        const f2 sn = (SEEDS && i == 0) ? seed_pair(st, theta, k + 1)
                    : (SEEDS && i == 2) ? seed_pair_next(st, c2) : next_pair(st, c4);
        accA = __builtin_elementwise_fma((f2){a0[i], a0[i + 1]}, sn, accA);
        accB = __builtin_elementwise_fma((f2){a1[i], a1[i + 1]}, sn, accB);
      }
    }
  }
}

// Work-distribution state of the fused kernel: kSchedSets independent sets (one per launch in
// flight, handed out round-robin by the launcher), each with 8 (pull counter, done counter) pairs
// on separate 128-byte lines.  Zero at module load; every launch leaves its set zeroed again (the
// last block of each residue class resets its pair), so no memset is needed on the launch path.
constexpr int kSchedSets = 32;
__device__ unsigned g_sched[kSchedSets][8][64];

typedef float sgpr16 __attribute__((ext_vector_type(16)));

// 16 harmonics k+1 .. k+16, all live, rows at byte offset OFF from p0/p1 (immediate in the load:
// no scalar address arithmetic, no guards - SALU instructions are not free, they take issue
// slots of the wave and of the CU's scalar unit)
template <bool SEEDS, int OFF>
__device__ __forceinline__ void harm_hex_full(const float* p0, const float* p1, int k, float theta,
                                              float c4, float c2, PairState& st, f2& accA, f2& accB) {
  sgpr16 a0, a1;
  asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a0), "=&s"(a1) : "s"(p0), "s"(p1), "i"(OFF) : "memory");
  // the recurrence runs one pair ahead of the accumulates: the pair an accumulate reads was written
  // three instructions earlier, not by the instruction before it
  f2 sn = SEEDS ? seed_pair(st, theta, k + 1) : next_pair(st, c4);
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    f2 nx = sn;
    if (i + 2 < 16) nx = (SEEDS && i == 0) ? seed_pair_next(st, c2) : next_pair(st, c4);
    accA = __builtin_elementwise_fma((f2){a0[i], a0[i + 1]}, sn, accA);
    accB = __builtin_elementwise_fma((f2){a1[i], a1[i + 1]}, sn, accB);
    sn = nx;
  }
}

typedef float sgpr4 __attribute__((ext_vector_type(4)));

// 4 more harmonics k+1 .. k+4 continuing the recurrence of the preceding super-block (no new seeds)
template <int OFF>
__device__ __forceinline__ void harm_quad_cont(const float* p0, const float* p1, float c4,
                                               PairState& st, f2& accA, f2& accB) {
  sgpr4 a0, a1;
  asm volatile("s_load_dwordx4 %0, %2, %4\n\ts_load_dwordx4 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a0), "=&s"(a1) : "s"(p0), "s"(p1), "i"(OFF) : "memory");
#pragma unroll
  for (int i = 0; i < 4; i += 2) {
    const f2 sn = next_pair(st, c4);
    accA = __builtin_elementwise_fma((f2){a0[i], a0[i + 1]}, sn, accA);
    accB = __builtin_elementwise_fma((f2){a1[i], a1[i + 1]}, sn, accB);
  }
}

struct FusedArgs {
  int B, F, K, Kp, N, hop, units_per_row, n_units;
  float sample_rate, nyquist;
  unsigned flags;
  int inputs_are_controls, amp_linear;
  int f0_vec;                    // f0 rows are 16-byte aligned (F % 4 == 0, aligned base): float4 prefix loads
  float inv_upr;                 // 1 / units_per_row
  int sched_set;                 // which g_sched set this launch uses
  float nyq_lo, nyq_hi;          // nyquist * (1 -+ 4e-6): guard band of the live-harmonic counts
  // host-side constants (no fp64 divisions / hoisted-then-spilled invariants on chip)
  double inv_sr, inv_2hop, hop_d, half_hm1;   // 1/sample_rate, 1/(2*hop), hop, (hop-1)/2
};

constexpr int kMaxUnitRows = kMaxUnitFrames + 1;

// LDS tables of one unit (written by the phase wave, read by every wavefront's tiles)
template <int FPB>
struct UnitTables {
  int next_unit;
  double theta[FPB], w[FPB], dw[FPB];
  float f0[FPB + 2];
  int kA[FPB], kN[FPB];
};

// Sum over the LPR lanes that share a matrix row; every lane of the group gets the sum.  Full EXEC.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  if (LPR == 64) return wave_sum_dpp(v);
  v += dpp_mov0<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov0<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov0<0x141, 0xF>(v);    // row_half_mirror
  v += dpp_mov0<0x140, 0xF>(v);    // row_mirror: every lane holds its 16-lane row's sum
  if (LPR == 32)                   // the neighbouring row of the pair: lane ^ 16 (swizzle, no address VGPR)
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
  return v;
}

// Phase A is pure overhead relative to the synthesis loop (it touches 900 values per unit, the loop
// 51200), but it is paid in issue slots on every wavefront: an earlier layout (flat float4 groups
// staged through LDS, 5 barriers, run-time flags) cost ~370 VALU + ~240 SALU instructions per
// wavefront per unit - a third of the kernel's instruction count (profiles/r01_pmc_sq_counters_b32).
// This layout gives LPR lanes to one matrix row, so a row's sum is a DPP reduction in registers:
// no LDS staging, no barrier inside phase A, and the common flag combination (STD) is compiled in.
template <int LPR, bool ONE_TILE, bool STD, int FPB>   // LPR lanes per row (K <= 4*LPR); ONE_TILE: hop == 64;
                                              // STD: scale + Nyquist-normalise, no controls written;
                                              // FPB: frames per unit - 16 when the whole job is one round of
                                              // units (batch 32: 2016 units on 2048 blocks, phase A paid once
                                              // per 16 frames), 8 otherwise (finer units balance better)
__global__ __launch_bounds__(256, 8) void harm_fused_kernel(
    const float* __restrict__ amplitudes, const float* __restrict__ hd,
    const float* __restrict__ f0_all, float* __restrict__ ctl_amp_arg, float* __restrict__ ctl_hd_arg,
    float* ws /*[gridDim.x][FPB + 1][Kp]*/, float* __restrict__ audio, FusedArgs p) {
  constexpr int kUnitRows = FPB + 1;
  __shared__ UnitTables<FPB> t;
  float* __restrict__ ctl_amp = STD ? nullptr : ctl_amp_arg;
  float* __restrict__ ctl_hd = STD ? nullptr : ctl_hd_arg;
  // debug timeline (flag 0x02000000, generic variant only): ctl_amp is reinterpreted as long long [gridDim.x][16]
  const bool dbg_time = !STD && (p.flags & 0x02000000u) != 0;
  long long* dbg = dbg_time ? reinterpret_cast<long long*>(ctl_amp_arg) + (size_t)blockIdx.x * 16 : nullptr;
  int dbg_n = 0;
#define DDSP_STAMP() do { if (!STD && dbg_time && threadIdx.x == 0 && dbg_n < 16) dbg[dbg_n++] = wall_clock64(); } while (0)
  if (dbg_time) ctl_amp = nullptr;
  DDSP_STAMP();
  constexpr int RPW = 64 / LPR;                          // matrix rows per wavefront per pass
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int F = p.F, K = p.K, Kp = p.Kp;
  const int K4 = K >> 2;                                 // K % 4 == 0 on this path
  const int sub = lane / LPR, kq = lane % LPR;           // row within the wave's pass, float4 within the row
  const bool live = kq < K4;
  float* wsu = ws + (size_t)blockIdx.x * kUnitRows * Kp;
  const bool is_ctl = STD ? false : (p.inputs_are_controls != 0);
  const bool scale = STD ? true : ((p.flags & DDSP_HARM_SCALE_EXP_SIGMOID) && !is_ctl);
  const bool normalize = STD ? true : ((p.flags & DDSP_HARM_NORMALIZE_NYQUIST) && !is_ctl);
  const float kLog10 = 2.302585092994046f;      // tf.math.log(exponent), ddsp/core.py:403
  const float4* __restrict__ hd4 = reinterpret_cast<const float4*>(hd);
  float4* __restrict__ ctl_hd4 = reinterpret_cast<float4*>(ctl_hd);

  // The first unit of a block is its blockIdx; further units are handed out dynamically (one atomic
  // per unit): resident blocks keep pulling work, so the kernel balances itself whatever share of
  // the chip it gets - it is meant to run next to the FilteredNoise kernel on another stream.
  // Eight counters (one per XCD-aligned residue class of the unit index, 128 B apart) keep the pulls
  // off a single word (one word serves ~88 atomics/us): class x owns units x, x+8, x+16, ...
  const int xcls = blockIdx.x & 7;
  unsigned* counter = &g_sched[p.sched_set][xcls][0];
  unsigned* done = &g_sched[p.sched_set][xcls][32];
  const int first_pull = (((int)gridDim.x - xcls + 7) >> 3);   // blocks (= static units) of this class
  for (int unit = blockIdx.x; unit < p.n_units;) {
    const int b = __builtin_amdgcn_readfirstlane((int)(((float)unit + 0.5f) * p.inv_upr));   // unit / units_per_row
    const int c = unit - b * p.units_per_row;
    const int j0 = c * FPB;
    const int nfr = min(FPB, F - j0);
    const int row0 = b * F + j0;                                          // first (batch*frame) row

    // ---------------- phase A, rows: LPR lanes per row, RPW rows per wavefront per pass -----------
    // rows 0..nfr-1 are the unit's frames, row nfr is the halo (the next frame, clamped at F-1)
    // that the last frame interpolates towards.  K = 100: 8 rows in pass 0, the halo in pass 1.
    for (int r0 = wave * RPW; r0 <= nfr; r0 += 4 * RPW) {                  // wave-uniform trip count
      const int r = r0 + sub;
      const int rowi = b * F + min(j0 + min(r, nfr), F - 1);             // rows past the halo alias it (never stored)
      const float4 xv = live ? hd4[(size_t)rowi * K4 + kq] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float f0r = f0_all[rowi];
      const float ampr = amplitudes[rowi];
      float x[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (scale) x[u] = exp_sigmoid_fast(x[u], kLog10, 2.0f, 1e-7f);
        // core.remove_above_nyquist on f0 * [1..K]  (core.py:899-903, 1028-1045)
        if (normalize && (f0r * (float)(4 * kq + u + 1) >= p.nyquist)) x[u] = 0.0f;
      }
      float inv = 1.0f;
      if (!is_ctl) {                                  // core.safe_divide(hd, reduce_sum(hd))  (core.py:905-907, 207-210)
        const float part = group_sum<LPR>(live ? (x[0] + x[1]) + (x[2] + x[3]) : 0.0f);
        inv = __builtin_amdgcn_rcpf(part == 0.0f ? 1e-7f : part);
      }
      const float a = scale ? exp_sigmoid_fast(ampr, kLog10, 2.0f, 1e-7f) : ampr;
      const float4 h = make_float4(x[0] * inv, x[1] * inv, x[2] * inv, x[3] * inv);
      if (!STD) {                                     // the controls dict (the halo row belongs to the next unit)
        if (ctl_hd && live && r < nfr) ctl_hd4[(size_t)rowi * K4 + kq] = h;
        if (ctl_amp && kq == 0 && r < nfr) ctl_amp[rowi] = a;
      }
      // core.py:1097 amplitudes * distribution -> the block's slot
      if (live && r <= nfr)
        *reinterpret_cast<float4*>(&wsu[r * Kp + 4 * kq]) = make_float4(a * h.x, a * h.y, a * h.z, a * h.w);
    }
    DDSP_STAMP();                                      // 1: rows issued
    // ---------------- phase A, phase wave: fp64 prefix and everything that is per frame ------------
    // Frame j carries f[t] = f_j + (f_{j+1}-f_j) r/hop (legacy bilinear resize of f0), whose sum over
    // the frame is hop*f_j + (f_{j+1}-f_j)(hop-1)/2; summed over j < J this telescopes to
    // hop*sum_{j<J} f_j + (f_J - f_0)(hop-1)/2: only sum f_j is needed.
    if (wave == 3) {
      const float* __restrict__ f0 = f0_all + (size_t)b * F;
      double part = 0.0;
      if (p.f0_vec) {                                  // j0 % 4 == 0: whole float4s lie before j0
        const float4* __restrict__ f4 = reinterpret_cast<const float4*>(f0);
        const int n4 = j0 >> 2;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (lane + 64 * i < n4) ? f4[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) part += ((double)v[i].x + (double)v[i].y) + ((double)v[i].z + (double)v[i].w);
        for (int m = lane + 256; m < n4; m += 64) {     // F > 1024: rare
          const float4 q4 = f4[m];
          part += ((double)q4.x + (double)q4.y) + ((double)q4.z + (double)q4.w);
        }
      } else {
        for (int j = lane; j < j0; j += 64) part += (double)f0[j];
      }
      const double before = wave_sum_dpp(part);          // sum_{j < j0} f_j
      const float fj = f0[min(j0 + min(lane, nfr), F - 1)], fj1 = f0[min(j0 + min(lane + 1, nfr), F - 1)];
      const double fa = (double)fj, fb = (double)fj1;
      const double mine = (lane < nfr) ? fa : 0.0;
      double incl = mine;                                 // inclusive scan of f_j over the unit's frames
      // (every lane executes the DPP moves: a source lane masked off by EXEC would read as 0;
      //  sources shifted in from outside the 16-lane row read 0 by bound_ctrl=0 / old=0)
      incl += dpp_mov0<0x111, 0xF>(incl);   // row_shr:1
      incl += dpp_mov0<0x112, 0xF>(incl);   // row_shr:2
      incl += dpp_mov0<0x114, 0xF>(incl);   // row_shr:4
      incl += dpp_mov0<0x118, 0xF>(incl);   // row_shr:8
      const double s_excl = before + (incl - mine);       // sum_{j < j0+lane} f_j
      const double run = p.hop_d * s_excl + (fa - (double)f0[0]) * p.half_hm1;
      const double cyc = run * p.inv_sr;
      // [0,kA): below Nyquist for every sample of the frame; [kA,kN): decided per sample.
      // v_rcp_f32 (1 ulp) is well inside the 4e-6 guard band.
      const float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
      int kA = K, kN = K;
      if (fmx > 0.0f) kA = (int)fminf((float)K, floorf(p.nyq_lo * __builtin_amdgcn_rcpf(fmx)));
      if (fmn > 0.0f) kN = (int)fminf((float)K, floorf(p.nyq_hi * __builtin_amdgcn_rcpf(fmn)));
      kA = max(min(kA, kN), 0);
      if (lane <= FPB) t.f0[lane] = fj;
      if (lane < nfr) {
        t.theta[lane] = cyc - floor(cyc);                  // revolutions at the start of the frame
        t.w[lane] = fa * p.inv_sr;                         // revolutions per sample at r = 0
        t.dw[lane] = (fb - fa) * p.inv_sr * p.inv_2hop;    // half the per-sample slope
        t.kA[lane] = kA;
        t.kN[lane] = kN;
      }
    }
    DDSP_STAMP();                                      // 2: tables issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows have reached L2
    __syncthreads();
    if (STD || !(p.flags & 0x08000000u)) __builtin_amdgcn_s_dcache_inv();   // drop stale scalar-cache lines of the slot
    DDSP_STAMP();                                      // 3: slot + tables visible

    // ---------------- phase B: tiles of 64 samples ---------------------------------------------
    const int hop = p.hop;
    const float inv_hop = 1.0f / (float)hop;
    const int tiles_per_frame = hop >> 6;
    const int n_tiles = (!STD && (p.flags & 0x20000000u)) ? 0 : nfr * tiles_per_frame;   // experiment: phase A only
    for (int tile = wave; tile < n_tiles; tile += 4) {
      // ONE_TILE (hop == 64): r == lane for every tile, so everything that depends only on r (its
      // fp64 image, the interpolation weights) is loop invariant and hoisted by the compiler
      const int q = ONE_TILE ? tile : tile / tiles_per_frame;
      const int r = ONE_TILE ? lane : (tile - q * tiles_per_frame) * 64 + lane;
      const double rr = (double)r;
      // inclusive cumsum of f[t]/sr inside the frame: (r+1)*w + r(r+1)*dw
      const double cyc = t.theta[q] + (rr + 1.0) * (t.w[q] + t.dw[q] * rr);
      const float theta = (float)(cyc - floor(cyc));
      const int kA = __builtin_amdgcn_readfirstlane(t.kA[q]);
      const int kN = __builtin_amdgcn_readfirstlane(t.kN[q]);

      const float* a0p = wsu + q * Kp;
      const float* a1p = a0p + Kp;
      const float c4 = 2.0f * __builtin_amdgcn_cosf(theta + theta);   // 2 cos(4 pi theta)
      const float c2 = 2.0f * __builtin_amdgcn_cosf(theta);           // 2 cos(2 pi theta): seeds only
      PairState st = {{0.f, 0.f}, {0.f, 0.f}};
      f2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
      // every harmonic live (the common case): the whole row (zero padded to a multiple of 4);
      // otherwise only the groups of 4 that lie entirely below kA
      const int kend = (kA == K) ? ((K + 3) & ~3) : (kA & ~3);
      int k = 0;
      {
        const float* q0 = a0p;
        const float* q1 = a1p;
        for (; k + 64 <= kend; k += 64, q0 += 64, q1 += 64) {      // full super-blocks of 64: 4 seeds
          harm_hex_full<true, 0>(q0, q1, k, theta, c4, c2, st, accA, accB);
          harm_hex_full<false, 64>(q0, q1, k + 16, theta, c4, c2, st, accA, accB);
          harm_hex_full<false, 128>(q0, q1, k + 32, theta, c4, c2, st, accA, accB);
          harm_hex_full<false, 192>(q0, q1, k + 48, theta, c4, c2, st, accA, accB);
        }
        for (; k + 32 <= kend; k += 32, q0 += 32, q1 += 32) {      // full super-blocks of 32
          harm_hex_full<true, 0>(q0, q1, k, theta, c4, c2, st, accA, accB);
          harm_hex_full<false, 64>(q0, q1, k + 16, theta, c4, c2, st, accA, accB);
        }
        // up to three trailing groups of 4 simply continue the last super-block's recurrence
        // (K = 100: 64 + 32 + 4); only possible when at least one super-block came before
        if (k > 0 && kend - k <= 12)
          for (; k < kend; k += 4, q0 += 4, q1 += 4) harm_quad_cont<0>(q0, q1, c4, st, accA, accB);
      }
      for (; k < kend; k += 32) {           // the tail: octets with group guards, seeds in the first
        const int rem4 = (kend - k) >> 2;
        harm_oct<true>(a0p, a1p, k, min(rem4, 2), theta, c4, c2, st, accA, accB);
        if (rem4 > 2) harm_oct<false>(a0p, a1p, k + 8, min(rem4 - 2, 2), theta, c4, c2, st, accA, accB);
        if (rem4 > 4) harm_oct<false>(a0p, a1p, k + 16, min(rem4 - 4, 2), theta, c4, c2, st, accA, accB);
        if (rem4 > 6) harm_oct<false>(a0p, a1p, k + 24, min(rem4 - 6, 2), theta, c4, c2, st, accA, accB);
      }
      float acc0 = accA.x + accA.y, acc1 = accB.x + accB.y;
      k = min(kend, K);
      const float lerp = (float)r * inv_hop;
      if (k < kN) {                // remaining live harmonics, and those crossing Nyquist
        const float fj = t.f0[q], fj1 = t.f0[q + 1];
        for (; k < kN; ++k) {
          const float kf = (float)(k + 1);
          const float top = fj * kf, bot = fj1 * kf;
          // audio-rate mask on the interpolated frequency, TF's fp32 op order (core.py:942-944)
          const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp));
          const float sv = (fk >= p.nyquist) ? 0.0f : sin_rev(frac_phase(theta, kf));
          const float x0 = __builtin_nontemporal_load(a0p + k), x1 = __builtin_nontemporal_load(a1p + k);
          acc0 = fmaf(x0, sv, acc0);
          acc1 = fmaf(x1, sv, acc1);
        }
      }
      // frame-rate -> audio-rate amplitude envelope: weight of frame j+1 is lerp ('linear',
      // core.resample) or the periodic Hann(2*hop)[r] ('window', core.py:696-698)
      const float w_next = p.amp_linear ? lerp : 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp);
      const float w_cur = 1.0f - w_next;
      audio[(size_t)(row0 + q) * hop + r] = w_cur * acc0 + w_next * acc1;      // N == F * hop
    }
    DDSP_STAMP();                                      // 4: tiles done
    if (p.n_units <= (int)gridDim.x) return;           // one unit per block: nothing to pull or reset
    if (tid == 0) t.next_unit = xcls + 8 * (first_pull + (int)atomicAdd(counter, 1u));
    __syncthreads();              // also: the slot and the LDS tables are rewritten by the next unit
    unit = __builtin_amdgcn_readfirstlane(t.next_unit);
  }
  // the last block of this residue class to finish leaves the class's counters zeroed
  if (p.n_units > (int)gridDim.x && tid == 0) {
    if (atomicAdd(done, 1u) == (unsigned)first_pull - 1u) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#undef DDSP_STAMP
}

}  // namespace ddsp

// =====================================================================================
// C ABI
// =====================================================================================
using namespace ddsp;

static inline int check_launch() { return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH; }

constexpr int kFusedMaxBlocks = 256 * 8;     // persistent grid: 8 blocks of 256 threads per CU
// workspace = [ theta0: B*F doubles ][ ws_a: B*(F+1)*Kp floats ]  (two-kernel path), or one
// 17-row slot per persistent block (fused path)
static inline size_t theta_bytes(int B, int F) { return ((size_t)B * F * sizeof(double) + 63) & ~(size_t)63; }
extern "C" size_t ddsp_harmonic_workspace_bytes(int B, int F, int K, int N) {
  (void)N;
  if (B <= 0 || F <= 0 || K <= 0) return 0;
  const size_t two_kernel = theta_bytes(B, F) + (size_t)B * (size_t)(F + 1) * (size_t)round_up(K, 16) * sizeof(float);
  const size_t fused = (size_t)kFusedMaxBlocks * kMaxUnitRows * (size_t)round_up(K, 16) * sizeof(float);
  return two_kernel > fused ? two_kernel : fused;
}

static int launch_fused(const float* amps, const float* hd, const float* f0, float* audio,
                        float* ctl_amp, float* ctl_hd, void* workspace, int B, int F, int K, int N,
                        int sample_rate, unsigned flags, int inputs_are_controls, hipStream_t st) {
  FusedArgs p;
  p.B = B; p.F = F; p.K = K; p.Kp = round_up(K, 16); p.N = N; p.hop = N / F;
  // one round of 16-frame units if they all fit on the persistent grid, else 8-frame units
  const int fpb = ((long)B * ((F + 15) / 16) <= (long)kFusedMaxBlocks) ? 16 : 8;
  p.units_per_row = (F + fpb - 1) / fpb;
  p.n_units = B * p.units_per_row;
  p.sample_rate = (float)sample_rate;
  p.nyquist = (float)(sample_rate / 2.0);
  p.flags = flags;
  p.inputs_are_controls = inputs_are_controls;
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  p.f0_vec = ((F & 3) == 0 && ((uintptr_t)f0 & 15) == 0) ? 1 : 0;
  p.inv_upr = 1.0f / (float)p.units_per_row;
  p.inv_sr = 1.0 / (double)sample_rate;
  p.inv_2hop = 0.5 / (double)p.hop;
  p.hop_d = (double)p.hop;
  p.half_hm1 = ((double)p.hop - 1.0) * 0.5;
  p.nyq_lo = p.nyquist * (1.0f - 4e-6f);
  p.nyq_hi = p.nyquist * (1.0f + 4e-6f);
  static const int max_blocks = [] {
    const char* e = getenv("DDSP_EXP_HARM_BLOCKS");
    const int v = e ? atoi(e) : kFusedMaxBlocks;
    return v > 0 && v <= kFusedMaxBlocks ? v : kFusedMaxBlocks;
  }();
  const dim3 grid((unsigned)(p.n_units < max_blocks ? p.n_units : max_blocks)), block(256);
  static std::atomic<unsigned> ticket{0};
  p.sched_set = (int)(ticket.fetch_add(1u) % (unsigned)kSchedSets);
  // A set may only be handed out again when the launch that had it last is done: launches on ONE stream are, by stream
  // order; with more than kSchedSets streams in flight the new launch waits for the set's last user (an event per set,
  // recorded behind every launch: ~1 us of host time on a path that is not the default one).
  // (g_sched is a __device__ array: one copy per GPU.  The events are per (device, set) as well - an event belongs to the
  // device it was created on - and made under a lock: two host threads may launch at once.  ADVICE r3)
  constexpr int kMaxDevices = 16;
  struct SetEvents { hipEvent_t done[kSchedSets]; bool made[kSchedSets]; };
  static SetEvents set_events[kMaxDevices] = {};
  static std::mutex set_lock;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return DDSP_ERR_LAUNCH;
  hipEvent_t set_done = nullptr;
  // The lock is held from the wait on the set's last user THROUGH the launch and the record behind it (with the
  // wait and the record locked separately, a second host thread handed the same set could slip in between - wait on the
  // event's OLD record and launch beside this one on the same g_sched set).  ~2 us of host time, on a path that is not the
  // default one.
  std::unique_lock<std::mutex> set_guard(set_lock);
  {
    SetEvents& se = set_events[dev];
    if (se.made[p.sched_set]) {
      (void)hipStreamWaitEvent(st, se.done[p.sched_set], 0);      // (recorded behind the set's last launch, below)
    } else {
      if (hipEventCreateWithFlags(&se.done[p.sched_set], hipEventDisableTiming) != hipSuccess) return DDSP_ERR_LAUNCH;
      se.made[p.sched_set] = true;
    }
    set_done = se.done[p.sched_set];
  }
  // STD: the flag combination of Harmonic.__call__ with default arguments, compiled in
  const bool std_flags = (flags & DDSP_HARM_SCALE_EXP_SIGMOID) && (flags & DDSP_HARM_NORMALIZE_NYQUIST) &&
                         !inputs_are_controls && !ctl_amp && !ctl_hd && (flags >> 24) == 0;
  hipEvent_t ev0, ev1;
  profile_kernel_events(kHarmFused, &ev0, &ev1);
#define DDSP_LAUNCH_FUSED(LPR, ONE, STD)                                                     \
  do {                                                                                       \
    if (fpb == 16)                                                                           \
      hipExtLaunchKernelGGL((harm_fused_kernel<LPR, ONE, STD, 16>), grid, block, 0, st, ev0, ev1, 0, amps, hd, \
                            f0, ctl_amp, ctl_hd, (float*)workspace, audio, p);               \
    else                                                                                     \
      hipExtLaunchKernelGGL((harm_fused_kernel<LPR, ONE, STD, 8>), grid, block, 0, st, ev0, ev1, 0, amps, hd,  \
                            f0, ctl_amp, ctl_hd, (float*)workspace, audio, p);               \
  } while (0)
#define DDSP_LAUNCH_FUSED_LPR(LPR)                                                           \
  do {                                                                                       \
    if (p.hop == 64) { if (std_flags) DDSP_LAUNCH_FUSED(LPR, true, true); else DDSP_LAUNCH_FUSED(LPR, true, false); } \
    else { if (std_flags) DDSP_LAUNCH_FUSED(LPR, false, true); else DDSP_LAUNCH_FUSED(LPR, false, false); }           \
  } while (0)
  if (K <= 128) DDSP_LAUNCH_FUSED_LPR(32);
  else if (K <= 256) DDSP_LAUNCH_FUSED_LPR(64);
  else return DDSP_ERR_UNSUPPORTED;
#undef DDSP_LAUNCH_FUSED_LPR
#undef DDSP_LAUNCH_FUSED
  if (hipEventRecord(set_done, st) != hipSuccess) return DDSP_ERR_LAUNCH;       // (still under set_guard)
  set_guard.unlock();
  return check_launch();
}
// fused path: hop a multiple of 64, K a multiple of 4 (16-byte rows), caller buffers 16-byte aligned
static inline bool fused_ok(int F, int K, int N, const void* hd, const void* ctl_hd) {
  return (N % F) == 0 && ((N / F) % 64) == 0 && K <= 256 && (K % 4) == 0 &&
         (((uintptr_t)hd | (uintptr_t)ctl_hd) & 15) == 0;
}

static int launch_controls(const float* amps, const float* hd, const float* f0, float* ctl_amp,
                           float* ctl_hd, void* workspace, int B, int F, int K, int N,
                           int sample_rate, unsigned flags, int inputs_are_controls,
                           hipStream_t st, const float* initial_phase = nullptr,
                           float* final_phase = nullptr, bool theta_only = false) {
  ControlsArgs p;
  p.B = B; p.F = F; p.K = K; p.Kp = round_up(K, 16);
  p.hop = workspace ? N / F : 1;
  const long rows = (long)B * F;
  p.n_ctl_blocks = theta_only ? 0 : (int)((rows + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave));   // theta_only: just the phase scan
  p.nyquist = (float)(sample_rate / 2.0);
  p.sample_rate = (float)sample_rate;
  p.flags = flags;
  p.inputs_are_controls = inputs_are_controls;
  p.initial_phase = initial_phase; p.final_phase = final_phase;
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
  else if (nchunk <= 16) DDSP_LAUNCH_CTL(16);          // (round 6: up to 2048 harmonics - 128 registers of row values per lane;
  else if (nchunk <= 32) DDSP_LAUNCH_CTL(32);          //  the reference has no cap, VERDICT r5 "missing" #3)
  else return DDSP_ERR_UNSUPPORTED;      // K > 2048 harmonics
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
  p.no_audio_mask = (flags & DDSP_HARM_NO_AUDIO_RATE_MASK) ? 1 : 0;
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
  if (K > kMaxHarmonics) return DDSP_ERR_UNSUPPORTED;
  return DDSP_OK;
}

extern "C" int ddsp_harmonic_controls_f32(const float* amplitudes, const float* hd,
                                          const float* f0_hz, float* ctl_amp, float* ctl_hd,
                                          int B, int F, int K, int sample_rate, unsigned flags,
                                          void* stream) {
  if (!amplitudes || !hd || !f0_hz || !ctl_amp || !ctl_hd) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (K > kMaxHarmonics) return DDSP_ERR_UNSUPPORTED;
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
  if (fused_ok(F, K, N, ctl_hd, nullptr))
    return launch_fused(ctl_amp, ctl_hd, f0_hz, audio, nullptr, nullptr, workspace, B, F, K, N,
                        sample_rate, flags, /*inputs_are_controls=*/1, st);
  rc = launch_controls(ctl_amp, ctl_hd, f0_hz, nullptr, nullptr, workspace, B, F, K, N,
                       sample_rate, flags, /*inputs_are_controls=*/1, st);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, workspace, audio, B, F, K, N, sample_rate, flags, st);
}

// Harmonic.__call__ with processors.Add fused in (ddsp/processors.py:162-176; the node that follows the two synths in every
// shipped DAG, gin/models/ae.gin:49-56): audio = Harmonic(...) + add_signal in one launch, one [B,N] stream written
// where the three-kernel form moves three more.  Only where the wavetable kernel applies (ddsp_harmonic_f32's default
// path: hop % 64 == 0, K <= 200, default flags); DDSP_ERR_UNSUPPORTED otherwise - the caller then runs
// ddsp_harmonic_f32 and ddsp_add_f32.
extern "C" int ddsp_harmonic_add_f32(const float* amplitudes, const float* hd, const float* f0_hz, const float* add_signal,
                                     float* audio, int B, int F, int K, int N, int sample_rate, unsigned flags,
                                     void* stream) {
  if (!amplitudes || !hd || !f0_hz || !add_signal || !audio) return DDSP_ERR_NULL_POINTER;
  int rc = check_harmonic_shape(B, F, K, N, sample_rate);
  if (rc != DDSP_OK) return rc;
  if (!harm_table_ok(F, K, N, hd, nullptr, nullptr, flags, /*inputs_are_controls=*/0)) return DDSP_ERR_UNSUPPORTED;
  return launch_harm_table(amplitudes, hd, f0_hz, audio, nullptr, nullptr, add_signal, B, F, K, N, sample_rate, flags,
                           (hipStream_t)stream);
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
  if (harm_table_ok(F, K, N, hd, ctl_amp, ctl_hd, flags, /*inputs_are_controls=*/0))
    return launch_harm_table(amplitudes, hd, f0_hz, audio, ctl_amp, ctl_hd, nullptr, B, F, K, N, sample_rate, flags, st);
  flags &= ~DDSP_HARM_DIRECT_SUM;
  if (fused_ok(F, K, N, hd, ctl_hd))
    return launch_fused(amplitudes, hd, f0_hz, audio, ctl_amp, ctl_hd, workspace, B, F, K, N,
                        sample_rate, flags, /*inputs_are_controls=*/0, st);
  rc = launch_controls(amplitudes, hd, f0_hz, ctl_amp, ctl_hd, workspace, B, F, K, N,
                       sample_rate, flags, /*inputs_are_controls=*/0, st);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, workspace, audio, B, F, K, N, sample_rate, flags, st);
}

// core.streaming_harmonic_synthesis (ddsp/core.py:1114-1164): frame-wise controls of a short chunk
// (the VST model passes 2 frames: previous and current, ddsp/training/inference.py:446-472) ->
// audio of the chunk + the fundamental's phase to carry into the next call.  Same closed forms as
// ddsp_harmonic_signal_f32 with the phase offset added and, as harmonic_oscillator_bank
// (core.py:966-1025) does, no audio-rate Nyquist mask; always the two-kernel path (the chunks
// are a few hundred samples of one clip: latency, not throughput).
extern "C" int ddsp_harmonic_streaming_f32(const float* amplitudes, const float* hd, const float* f0_hz,
                                           const float* initial_phase, float* audio,
                                           float* final_phase, void* workspace,
                                           size_t workspace_bytes, int B, int F, int K, int N,
                                           int sample_rate, unsigned flags, void* stream) {
  if (!amplitudes || !hd || !f0_hz || !audio || !workspace) return DDSP_ERR_NULL_POINTER;
  int rc = check_harmonic_shape(B, F, K, N, sample_rate);
  if (rc != DDSP_OK) return rc;
  if (workspace_bytes < ddsp_harmonic_workspace_bytes(B, F, K, N) || ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // normalize_harmonics with the frame-rate Nyquist mask when a distribution is given
  // (core.py:1141-1148); DDSP_HARM_INPUTS_ARE_AMPLITUDES: hd already holds amplitudes * distribution
  const int as_is = (flags & DDSP_HARM_INPUTS_ARE_AMPLITUDES) ? 1 : 0;
  const unsigned cflags = as_is ? 0u : DDSP_HARM_NORMALIZE_NYQUIST;
  rc = launch_controls(amplitudes, hd, f0_hz, nullptr, nullptr, workspace, B, F, K, N, sample_rate,
                       cflags, as_is, st, initial_phase, final_phase);
  if (rc != DDSP_OK) return rc;
  return launch_synth(f0_hz, workspace, audio, B, F, K, N, sample_rate,
                      (flags & DDSP_HARM_AMP_LINEAR) | DDSP_HARM_NO_AUDIO_RATE_MASK, st);
}

// =====================================================================================
// Backward pass of Harmonic.__call__: dL/d(amplitudes), dL/d(harmonic_distribution) from
// dL/d(audio) - what tf.GradientTape computes through ddsp/synths.py:94-146 in
// ddsp/training/trainers.py:162-171.  f0 is a constant here (no gradient is formed for it).
//
//   audio[n] = w_cur(r) sum_k a[j,k] s_k(n) + w_next(r) sum_k a[j+1,k] s_k(n),   n = j hop + r,
//   s_k(n) = sin(2 pi k theta(n)) where the harmonic is below Nyquist at sample n, a = amp * hd_norm.
//
//   harm_bwd_pq_kernel     lanes = harmonics, one block per (row, frame): every lane accumulates
//                          P[j,k] = sum_r w_cur(r) g(n) s_k(n) and Q[j,k] = sum_r w_next(r) g(n) s_k(n)
//                          over the frame's samples - the per-sample values (phase, weighted
//                          gradient) are computed once with lanes = samples and broadcast from
//                          LDS, so there is no cross-lane reduction anywhere.
//   harm_bwd_chain_kernel  one wavefront per (row, frame): dL/da[j] = P[j] + Q[j-1] (+ Q[F-1] for
//                          the held last frame), then the frame-rate chain rule through
//                          amp * hd_norm, safe_divide, the Nyquist mask and exp_sigmoid.
// =====================================================================================
namespace ddsp {

constexpr int kBwdMaxHop = 2048;


// A block takes `fb` consecutive frames of one row (fb * hop <= kBwdMaxHop samples staged at once):
// 32000 two-wavefront blocks of one frame each were bound by the block launch rate.
template <int NW>
__global__ __launch_bounds__(64 * NW) void harm_bwd_pq_kernel(const float* __restrict__ f0_all,
                                                              const double* __restrict__ theta0,
                                                              const float* __restrict__ grad_audio,
                                                              float* __restrict__ pq /*[2][B*F][K]*/,
                                                              size_t q_offset, int fb, BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float4 sm[];          // [fb * hop] {w_cur g, w_next g, theta, lerp}: (x, y) an aligned register pair
  const int tid = threadIdx.x, b = blockIdx.y;
  const int j0 = blockIdx.x * fb, nfr = min(fb, p.F - j0);
  const float* __restrict__ f0 = f0_all + (size_t)b * p.F;
  {
    const double inv_sr = 1.0 / (double)p.sample_rate, inv_2hop = 0.5 / (double)p.hop;
    const float inv_hop = 1.0f / (float)p.hop;
    const float* __restrict__ g = grad_audio + (size_t)b * p.N + (size_t)j0 * p.hop;
    for (int e = tid; e < nfr * p.hop; e += 64 * NW) {
      const int q = e / p.hop, r = e - q * p.hop;
      const int j = j0 + q;
      const float fj = f0[j], fj1 = f0[min(j + 1, p.F - 1)];
      const double wj = (double)fj * inv_sr, dw = ((double)fj1 - (double)fj) * inv_sr * inv_2hop;
      const double rr = (double)r;
      const double cyc = theta0[(size_t)b * p.F + j] + (rr + 1.0) * (wj + dw * rr);
      const float lerp = (float)r * inv_hop;
      const float w_next = p.amp_linear ? lerp : 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp);
      const float gv = g[e];
      sm[e] = make_float4((1.0f - w_next) * gv, w_next * gv, (float)(cyc - floor(cyc)), lerp);
    }
  }
  __syncthreads();
  // every lane owns TWO harmonics, k0 = tid and k1 = tid + 64 NW: one broadcast read of the sample's
  // values feeds both (with one harmonic per lane the LDS pipe, 8 clocks per 16-byte broadcast read,
  // was the bound: 73 us at batch 32)
  const int k0 = tid, k1 = tid + 64 * NW;
  const float kf0 = (float)(k0 + 1), kf1 = (float)(k1 + 1);
  for (int q = 0; q < nfr; ++q) {
    const int j = j0 + q;
    const float fj = f0[j], fj1 = f0[min(j + 1, p.F - 1)];
    // [0,kA): below Nyquist for every sample of the frame; [kA,kN): decided per sample (as harm_synth_kernel)
    const float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
    int kA = p.K, kN = p.K;
    if (fmx > 0.0f) kA = (int)fminf((float)p.K, floorf(p.nyquist * (1.0f - 2e-6f) / fmx));
    if (fmn > 0.0f) kN = (int)fminf((float)p.K, floorf(p.nyquist * (1.0f + 2e-6f) / fmn));
    kA = max(min(kA, kN), 0);
    const float4* __restrict__ sq = sm + q * p.hop;
    float P0 = 0.0f, Q0 = 0.0f, P1 = 0.0f, Q1 = 0.0f;
    const bool second = 64 * NW < kN;                    // block-uniform: any live harmonic in the upper half?
    if (kA == kN) {                                      // block-uniform: no harmonic crosses Nyquist in this frame
      if (second) {
        // (P, Q) of a harmonic as one register pair: (w_cur g, w_next g) sin(..) + (P, Q) is ONE packed FMA, the two
        // arguments k theta one packed multiply - 6 instructions per sample and harmonic pair instead of 9 (two of
        // them the quarter-rate sines either way)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 kf01 = {kf0, kf1};
        f32x2 pq0 = {0.0f, 0.0f}, pq1 = {0.0f, 0.0f};
#pragma unroll 4
        for (int r = 0; r < p.hop; ++r) {
          const float4 v = sq[r];                        // same address for every lane: LDS broadcast
          // v_sin_f32 reduces |x| <= 256 revolutions itself; the rounding of k*theta (<= 2.4e-5 rad at
          // k = 100) is far inside the gradient tolerance, so the exact-fraction step of the forward
          // pass (3 more instructions per sine) is not paid here
          const f32x2 arg = kf01 * v.z;
          const float s0 = sin_rev(arg[0]), s1 = sin_rev(arg[1]);
          const f32x2 wg = {v.x, v.y};
          pq0 = __builtin_elementwise_fma(wg, (f32x2){s0, s0}, pq0);
          pq1 = __builtin_elementwise_fma(wg, (f32x2){s1, s1}, pq1);
        }
        P0 = pq0[0]; Q0 = pq0[1]; P1 = pq1[0]; Q1 = pq1[1];
      } else {
#pragma unroll 4
        for (int r = 0; r < p.hop; ++r) {
          const float4 v = sq[r];
          const float s0 = sin_rev(v.z * kf0);
          P0 = fmaf(v.x, s0, P0); Q0 = fmaf(v.y, s0, Q0);
        }
      }
      if (k0 >= kA) { P0 = 0.0f; Q0 = 0.0f; }
      if (k1 >= kA) { P1 = 0.0f; Q1 = 0.0f; }
    } else {
      const float top0 = fj * kf0, bot0 = fj1 * kf0, top1 = fj * kf1, bot1 = fj1 * kf1;
#pragma unroll 2
      for (int r = 0; r < p.hop; ++r) {
        const float4 v = sq[r];
        // audio-rate mask on the interpolated frequency, TF's fp32 op order (core.py:942-944)
        const float fk0 = rn_add(top0, rn_mul(rn_sub(bot0, top0), v.w));
        const float fk1 = rn_add(top1, rn_mul(rn_sub(bot1, top1), v.w));
        const float s0 = (fk0 >= p.nyquist || k0 >= kN) ? 0.0f : sin_rev(v.z * kf0);
        const float s1 = (fk1 >= p.nyquist || k1 >= kN) ? 0.0f : sin_rev(v.z * kf1);
        P0 = fmaf(v.x, s0, P0); Q0 = fmaf(v.y, s0, Q0);
        P1 = fmaf(v.x, s1, P1); Q1 = fmaf(v.y, s1, Q1);
      }
    }
    const size_t at = ((size_t)b * p.F + j) * p.K;
    if (k0 < p.K) { pq[at + k0] = P0; pq[q_offset + at + k0] = Q0; }
    if (k1 < p.K) { pq[at + k1] = P1; pq[q_offset + at + k1] = Q1; }
  }
}

#ifndef DDSP_CHAIN_ROWS
#define DDSP_CHAIN_ROWS 1
#endif
constexpr int kChainRows = DDSP_CHAIN_ROWS;
// wavefronts (= rows) per block (-DDDSP_CHAIN_WAVES): 19.2 / 19.2 / 19.7 us at batch 32 and 57 / 60 / 63 at batch 128 for 4 / 8 / 16
// (round 5, profiles/r05o: not the rate blocks are launched at either): four it stays
#ifndef DDSP_CHAIN_WAVES
#define DDSP_CHAIN_WAVES 4
#endif
constexpr int kChainWaves = DDSP_CHAIN_WAVES;
template <int NCHUNK>   // ceil(K/64) <= NCHUNK
__global__ __launch_bounds__(64 * kChainWaves) void harm_bwd_chain_kernel(const float* __restrict__ amplitudes,
                                                             const float* __restrict__ hd,
                                                             const float* __restrict__ f0_hz,
                                                             const float* __restrict__ pq, size_t q_offset,
                                                             float* __restrict__ grad_amp,
                                                             float* __restrict__ grad_hd, long rows,
                                                             BwdArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // dL/da[j] = P[j] + Q[j-1]; the last frame also receives Q[F-1] (row F repeats row F-1)
  const int F = p.F;
  // kChainRows rows per wavefront (-DDDSP_CHAIN_ROWS): measured 20.1 / 21.5 / 22.4 us for 1 / 2 / 4 rows at batch 32 (r05y) - the
  // kernel is not waiting for its loads; one row it stays
#pragma unroll
  for (int u = 0; u < kChainRows; ++u) {
    const long row = ((long)blockIdx.x * kChainWaves + wave) * kChainRows + u;
    if (row >= rows) break;
    const int j = (int)(row % F);
    // (no branch around the loads and no arithmetic between them - harm_chain_row issues every load of the row, then waits
    //  once: frame 0 reads its own Q and weighs it 0; the last frame's extra term re-reads an address the row has just read
    //  unless the row IS the last frame)
    const size_t q_back = j > 0 ? (size_t)p.K : 0, x_back = j == F - 1 ? 0 : q_back;
    const float q_keep = j > 0 ? 1.0f : 0.0f, x_keep = j == F - 1 ? 1.0f : 0.0f;
    harm_chain_row<NCHUNK>(lane, row, j, amplitudes, hd, f0_hz, grad_amp, grad_hd, p, [&](int k) {
      const size_t at = (size_t)row * p.K + k;
      return ChainPq{pq[at], pq[q_offset + at - q_back], pq[q_offset + at - x_back]};
    }, [&](ChainPq v) { return fmaf(v.x, x_keep, fmaf(v.q, q_keep, v.p)); });
  }
}

// The frame-rate chain rule on its own: dL/d (amplitudes * harmonic_distribution)[B,F,K] handed in (the materialised chain's
// backward pass forms it with the adjoint of core.resample), dL/d amplitudes and dL/d harmonic_distribution out.
template <int NCHUNK>
__global__ __launch_bounds__(256) void harm_controls_bwd_kernel(const float* __restrict__ amplitudes, const float* __restrict__ hd,
                                                                const float* __restrict__ f0_hz, const float* __restrict__ gha,
                                                                float* __restrict__ grad_amp, float* __restrict__ grad_hd,
                                                                long rows, BwdArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  harm_chain_row<NCHUNK>(lane, row, (int)(row % p.F), amplitudes, hd, f0_hz, grad_amp, grad_hd, p,
                         [&](int k) { return gha[(size_t)row * p.K + k]; }, [](float v) { return v; });
}

static inline size_t bwd_pq_floats(int B, int F, int K) { return ((size_t)B * F * K + 15) & ~(size_t)15; }

}  // namespace ddsp

extern "C" size_t ddsp_harmonic_backward_workspace_bytes(int B, int F, int K, int N) {
  const size_t fwd = ddsp_harmonic_workspace_bytes(B, F, K, N);
  if (fwd == 0) return 0;
  return ((fwd + 63) & ~(size_t)63) + 2 * bwd_pq_floats(B, F, K) * sizeof(float);
}

extern "C" int ddsp_harmonic_backward_f32(const float* amplitudes, const float* hd, const float* f0_hz,
                                          const float* grad_audio, float* grad_amplitudes,
                                          float* grad_hd, void* workspace, size_t workspace_bytes,
                                          int B, int F, int K, int N, int sample_rate,
                                          unsigned flags, int inputs_are_controls, void* stream) {
  if (!amplitudes || !hd || !f0_hz || !grad_audio || !grad_amplitudes || !grad_hd || !workspace)
    return DDSP_ERR_NULL_POINTER;
  int rc = check_harmonic_shape(B, F, K, N, sample_rate);
  if (rc != DDSP_OK) return rc;
  if (K > 256 || N / F > kBwdMaxHop) return DDSP_ERR_UNSUPPORTED;
  if (workspace_bytes < ddsp_harmonic_backward_workspace_bytes(B, F, K, N) || ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // the fp64 phase prefix theta0[B,F] of the generic path (its scan blocks only)
  rc = launch_controls(amplitudes, hd, f0_hz, nullptr, nullptr, workspace, B, F, K, N, sample_rate,
                       flags, inputs_are_controls, st, nullptr, nullptr, /*theta_only=*/true);
  if (rc != DDSP_OK) return rc;
  const size_t fwd = (ddsp_harmonic_workspace_bytes(B, F, K, N) + 63) & ~(size_t)63;
  float* pq = (float*)((char*)workspace + fwd);
  const size_t q_offset = bwd_pq_floats(B, F, K);
  BwdArgs p;
  p.F = F; p.K = K; p.N = N; p.hop = N / F;
  p.sample_rate = (float)sample_rate; p.nyquist = (float)(sample_rate / 2.0);
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  p.flags = flags; p.inputs_are_controls = inputs_are_controls;
  const double* theta0 = (const double*)workspace;
  if (harm_bwd_table_ok(F, K, N)) {
    // up to 128 harmonics: spread the weighted gradient onto the table grid, one product with the transposed sine matrix
    // (harmonic_bwd_table.hip) instead of a sine per sample and harmonic
    ProfileScope prof(kHarmBwdTable, st);
    rc = launch_harm_bwd_table(f0_hz, theta0, grad_audio, pq, q_offset, B, F, K, N, sample_rate, p.amp_linear, st, amplitudes, hd,
                               grad_amplitudes, grad_hd, flags, inputs_are_controls);
    if (rc == 1) return check_launch();            // the chain rule ran in the same launch
    if (rc != DDSP_OK) return rc;
  } else {
    ProfileScope prof(kHarmBwdPq, st);
    const int fb = max(1, min(8, kBwdMaxHop / p.hop));           // frames per block
    const dim3 grid((unsigned)((F + fb - 1) / fb), (unsigned)B);
    const size_t lds = (size_t)fb * p.hop * sizeof(float4);      // 8 KB at hop 64: the occupancy is not LDS-limited
    if (K <= 128) hipLaunchKernelGGL((harm_bwd_pq_kernel<1>), grid, dim3(64), lds, st, f0_hz, theta0, grad_audio, pq, q_offset, fb, p);
    else hipLaunchKernelGGL((harm_bwd_pq_kernel<2>), grid, dim3(128), lds, st, f0_hz, theta0, grad_audio, pq, q_offset, fb, p);
  }
  {
    ProfileScope prof(kHarmBwdChain, st);
    const long rows = (long)B * F;
    const dim3 grid((unsigned)((rows + kChainWaves * kChainRows - 1) / (kChainWaves * kChainRows)));
    const int nchunk = (K + 63) / 64;
#define DDSP_LAUNCH_BWD(NC) hipLaunchKernelGGL((harm_bwd_chain_kernel<NC>), grid, dim3(64 * kChainWaves), 0, st, amplitudes, hd, \
                                               f0_hz, (const float*)pq, q_offset, grad_amplitudes, grad_hd, rows, p)
    if (nchunk <= 1) DDSP_LAUNCH_BWD(1);
    else if (nchunk <= 2) DDSP_LAUNCH_BWD(2);
    else DDSP_LAUNCH_BWD(4);
#undef DDSP_LAUNCH_BWD
  }
  return check_launch();
}

// The backward pass of Harmonic.get_controls + the product amplitudes * harmonic_distribution of core.harmonic_synthesis
// (synths.py:94-121, core.py:1097) on its own: grad_harmonic_amplitudes [B,F,K] = dL/d (amplitudes * distribution) in.
extern "C" int ddsp_harmonic_controls_backward_f32(const float* amplitudes, const float* hd, const float* f0_hz,
                                                   const float* grad_harmonic_amplitudes, float* grad_amplitudes,
                                                   float* grad_hd, int B, int F, int K, int sample_rate, unsigned flags,
                                                   int inputs_are_controls, void* stream) {
  if (!amplitudes || !hd || !f0_hz || !grad_harmonic_amplitudes || !grad_amplitudes || !grad_hd) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (K > kMaxHarmonics) return DDSP_ERR_UNSUPPORTED;
  BwdArgs p;
  p.F = F; p.K = K; p.N = 0; p.hop = 0;
  p.sample_rate = (float)sample_rate; p.nyquist = (float)(sample_rate / 2.0);
  p.amp_linear = 0; p.flags = flags; p.inputs_are_controls = inputs_are_controls;
  const long rows = (long)B * F;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = (K + 63) / 64;
#define DDSP_LAUNCH_CB(NC) hipLaunchKernelGGL((harm_controls_bwd_kernel<NC>), grid, dim3(256), 0, st, amplitudes, hd, f0_hz, \
                                              grad_harmonic_amplitudes, grad_amplitudes, grad_hd, rows, p)
  if (nchunk <= 1) DDSP_LAUNCH_CB(1);
  else if (nchunk <= 2) DDSP_LAUNCH_CB(2);
  else if (nchunk <= 4) DDSP_LAUNCH_CB(4);
  else if (nchunk <= 8) DDSP_LAUNCH_CB(8);
  else if (nchunk <= 16) DDSP_LAUNCH_CB(16);
  else DDSP_LAUNCH_CB(32);
#undef DDSP_LAUNCH_CB
  return check_launch();
}

// =====================================================================================
// Stand-alone core.oscillator_bank (ddsp/core.py:912-962) on materialised audio-rate envelopes
// [B,N,K] (what synths.Sinusoidal and direct callers use).  The Harmonic path never calls this:
// it fuses the same maths with closed-form phases.  Here the phase really is a scan over time:
//   pass 1  per (b, chunk of 256 samples): fp64 sum of f over the chunk, per sinusoid
//   pass 2  per (b, k): exclusive prefix over chunks (in revolutions, wrapped to [0,1))
//   pass 3  per (b, chunk): running phase from the chunk offset, Nyquist mask, sin, amplitude,
//           sum over sinusoids (DPP wave reduction + LDS across wavefronts)
// The fp64 wrapped phase is exact to ~1e-13 revolutions, i.e. closer to exact arithmetic than
// either tf.cumsum in fp32 or angular_cumsum; `use_angular_cumsum` is therefore accepted and ignored.
// =====================================================================================
namespace ddsp {
constexpr int kOscChunk = 256;

__global__ __launch_bounds__(256) void osc_chunk_sums_kernel(const float* __restrict__ freq,
                                                             double* __restrict__ sums, int N, int K,
                                                             int n_chunks) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int t0 = c * kOscChunk, t1 = min(t0 + kOscChunk, N);
  const float* __restrict__ fb = freq + (size_t)b * N * K;
  for (int k = threadIdx.x; k < K; k += 256) {
    double s = 0.0;
    for (int t = t0; t < t1; ++t) s += (double)fb[(size_t)t * K + k];
    sums[((size_t)b * n_chunks + c) * K + k] = s;
  }
}

__global__ __launch_bounds__(256) void osc_chunk_prefix_kernel(double* __restrict__ sums, int K,
                                                               int n_chunks, double inv_sr) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double run = 0.0;                                   // revolutions, wrapped
  for (int c = 0; c < n_chunks; ++c) {
    double* p = sums + ((size_t)b * n_chunks + c) * K + k;
    const double s = *p * inv_sr;
    *p = run;
    run += s;
    run -= floor(run);
  }
}

__global__ __launch_bounds__(256) void osc_apply_kernel(const float* __restrict__ freq,
                                                        const float* __restrict__ amp,
                                                        const double* __restrict__ offs,
                                                        float* __restrict__ out, int N, int K,
                                                        int n_chunks, double inv_sr, float nyquist,
                                                        int sum_sinusoids, int amp_per_sample) {
  __shared__ float s_acc[4][kOscChunk];               // one row of partial sums per wavefront, added in a fixed order at the end
  const int c = blockIdx.x, b = blockIdx.y;             // (an LDS atomic per wavefront and sample made the sum's last bit depend on
                                                        //  which wavefront came first: found by a bit-equality test in round 5)
  const int t0 = c * kOscChunk, t1 = min(t0 + kOscChunk, N);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ fb = freq + (size_t)b * N * K;
  // amp_per_sample: one amplitude per sample, [B,N] - the backward pass hands dL/d audio in here and reads
  // dL/d amplitude_envelopes = dL/d audio[n] mask sin(phase) out of the [B,N,K] output (ddsp_oscillator_bank_grad_amplitudes_f32)
  const float* __restrict__ ab = amp + (amp_per_sample ? (size_t)b * N : (size_t)b * N * K);
  for (int i = tid; i < 4 * kOscChunk; i += 256) (&s_acc[0][0])[i] = 0.0f;
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += 256) {               // every thread takes the trip: full waves for DPP
    const int k = k0 + tid;
    const bool live = k < K;
    double ph = live ? offs[((size_t)b * n_chunks + c) * K + k] : 0.0;
    for (int t = t0; t < t1; ++t) {
      float v = 0.0f;
      if (live) {
        const float f = fb[(size_t)t * K + k];
        ph += (double)f * inv_sr;                     // inclusive cumsum (core.py:955)
        ph -= floor(ph);
        const float a = (f >= nyquist) ? 0.0f : (amp_per_sample ? ab[t] : ab[(size_t)t * K + k]);   // remove_above_nyquist
        v = a * sin_rev((float)ph);
        if (!sum_sinusoids) out[((size_t)b * N + t) * K + k] = v;
      }
      if (sum_sinusoids) {
        const float s = wave_sum_dpp(v);
        if (lane == 0) s_acc[wave][t - t0] += s;        // this wavefront's row: no other writer
      }
    }
  }
  if (sum_sinusoids) {
    __syncthreads();
    for (int i = tid; i < t1 - t0; i += 256) out[(size_t)b * N + t0 + i] = ((s_acc[0][i] + s_acc[1][i]) + s_acc[2][i]) + s_acc[3][i];
  }
}
// dL/d frequency_envelopes of oscillator_bank: phase[n] = (2 pi / sr) sum_{t <= n} f[t] (core.py:950-955), so
//   dL/d f[t,k] = (2 pi / sr) sum_{n >= t} c[n,k],   c[n,k] = dL/d audio[n] A[n,k] mask[n,k] cos(phase[n,k])
// (the mask - tf.where - passes no gradient).  Pass 1 walks a chunk forwards like osc_apply_kernel, leaves c in `out` and the
// chunk's sum of c (fp64) in `csum`; osc_chunk_suffix_kernel turns the chunk sums into the sums over all LATER chunks; pass 2
// walks a chunk backwards and replaces c by its inclusive suffix sum.
__global__ __launch_bounds__(256) void osc_grad_freq_c_kernel(const float* __restrict__ freq, const float* __restrict__ amp,
                                                              const float* __restrict__ g, const double* __restrict__ offs,
                                                              float* __restrict__ out, double* __restrict__ csum, int N, int K,
                                                              int n_chunks, double inv_sr, float nyquist) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int t0 = c * kOscChunk, t1 = min(t0 + kOscChunk, N);
  const float* __restrict__ fb = freq + (size_t)b * N * K;
  const float* __restrict__ ab = amp + (size_t)b * N * K;
  const float* __restrict__ gb = g + (size_t)b * N;
  float* __restrict__ ob = out + (size_t)b * N * K;
  const float two_pi_over_sr = (float)(6.283185307179586 * inv_sr);
  for (int k = threadIdx.x; k < K; k += 256) {
    double ph = offs[((size_t)b * n_chunks + c) * K + k];
    double sum = 0.0;
    for (int t = t0; t < t1; ++t) {
      const float f = fb[(size_t)t * K + k];
      ph += (double)f * inv_sr;
      ph -= floor(ph);
      const float a = (f >= nyquist) ? 0.0f : ab[(size_t)t * K + k];
      const float cv = gb[t] * a * __builtin_amdgcn_cosf((float)ph) * two_pi_over_sr;
      ob[(size_t)t * K + k] = cv;
      sum += (double)cv;
    }
    csum[((size_t)b * n_chunks + c) * K + k] = sum;
  }
}

__global__ __launch_bounds__(256) void osc_chunk_suffix_kernel(double* __restrict__ csum, int K, int n_chunks) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double run = 0.0;                                   // the sum over the chunks behind this one
  for (int c = n_chunks - 1; c >= 0; --c) {
    double* p = csum + ((size_t)b * n_chunks + c) * K + k;
    const double s = *p;
    *p = run;
    run += s;
  }
}

__global__ __launch_bounds__(256) void osc_grad_freq_suffix_kernel(float* __restrict__ out, const double* __restrict__ csuf, int N,
                                                                   int K, int n_chunks) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int t0 = c * kOscChunk, t1 = min(t0 + kOscChunk, N);
  float* __restrict__ ob = out + (size_t)b * N * K;
  for (int k = threadIdx.x; k < K; k += 256) {
    double run = csuf[((size_t)b * n_chunks + c) * K + k];
    for (int t = t1 - 1; t >= t0; --t) {
      run += (double)ob[(size_t)t * K + k];
      ob[(size_t)t * K + k] = (float)run;
    }
  }
}
}  // namespace ddsp

extern "C" size_t ddsp_oscillator_bank_workspace_bytes(int B, int N, int K) {
  if (B <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)B * ((N + kOscChunk - 1) / kOscChunk) * K * sizeof(double);
}

extern "C" int ddsp_oscillator_bank_f32(const float* frequency_envelopes,
                                        const float* amplitude_envelopes, float* out,
                                        void* workspace, size_t workspace_bytes, int B, int N, int K,
                                        int sample_rate, int sum_sinusoids, void* stream) {
  if (!frequency_envelopes || !amplitude_envelopes || !out || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || K <= 0 || sample_rate <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_oscillator_bank_workspace_bytes(B, N, K) || ((uintptr_t)workspace & 7))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (N + kOscChunk - 1) / kOscChunk;
  const double inv_sr = 1.0 / (double)sample_rate;
  double* sums = (double*)workspace;
  hipLaunchKernelGGL(osc_chunk_sums_kernel, dim3(n_chunks, B), dim3(256), 0, st, frequency_envelopes,
                     sums, N, K, n_chunks);
  hipLaunchKernelGGL(osc_chunk_prefix_kernel, dim3((K + 255) / 256, B), dim3(256), 0, st, sums, K,
                     n_chunks, inv_sr);
  hipLaunchKernelGGL(osc_apply_kernel, dim3(n_chunks, B), dim3(256), 0, st, frequency_envelopes,
                     amplitude_envelopes, sums, out, N, K, n_chunks, inv_sr,
                     (float)(sample_rate / 2.0), sum_sinusoids, 0);
  return check_launch();
}

// dL/d amplitude_envelopes [B,N,K] of core.oscillator_bank (core.py:912-962) given dL/d audio [B,N]: the output is linear in
// the amplitudes, d audio[n] / d A[n,k] = mask(f[n,k] < Nyquist) sin(phase[n,k]) - the forward's own passes with dL/d audio as
// a per-sample amplitude and nothing summed.  (What tf.GradientTape forms for the materialised chain of harmonic_synthesis:
// 'nearest' / 'cubic' envelopes, n_samples that is not a multiple of n_frames.)  Workspace: ddsp_oscillator_bank_workspace_bytes.
extern "C" int ddsp_oscillator_bank_grad_amplitudes_f32(const float* frequency_envelopes, const float* grad_audio,
                                                        float* grad_amplitude_envelopes, void* workspace,
                                                        size_t workspace_bytes, int B, int N, int K, int sample_rate,
                                                        void* stream) {
  if (!frequency_envelopes || !grad_audio || !grad_amplitude_envelopes || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || K <= 0 || sample_rate <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_oscillator_bank_workspace_bytes(B, N, K) || ((uintptr_t)workspace & 7))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (N + kOscChunk - 1) / kOscChunk;
  const double inv_sr = 1.0 / (double)sample_rate;
  double* sums = (double*)workspace;
  hipLaunchKernelGGL(osc_chunk_sums_kernel, dim3(n_chunks, B), dim3(256), 0, st, frequency_envelopes, sums, N, K, n_chunks);
  hipLaunchKernelGGL(osc_chunk_prefix_kernel, dim3((K + 255) / 256, B), dim3(256), 0, st, sums, K, n_chunks, inv_sr);
  hipLaunchKernelGGL(osc_apply_kernel, dim3(n_chunks, B), dim3(256), 0, st, frequency_envelopes, grad_audio, sums,
                     grad_amplitude_envelopes, N, K, n_chunks, inv_sr, (float)(sample_rate / 2.0), 0, 1);
  return check_launch();
}

// dL/d frequency_envelopes [B,N,K] of ddsp_oscillator_bank_f32 (kernels above).  Workspace: TWICE ddsp_oscillator_bank_workspace_bytes.
extern "C" int ddsp_oscillator_bank_grad_frequencies_f32(const float* frequency_envelopes, const float* amplitude_envelopes,
                                                         const float* grad_audio, float* grad_frequency_envelopes,
                                                         void* workspace, size_t workspace_bytes, int B, int N, int K,
                                                         int sample_rate, void* stream) {
  if (!frequency_envelopes || !amplitude_envelopes || !grad_audio || !grad_frequency_envelopes || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || K <= 0 || sample_rate <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  const size_t half = ddsp_oscillator_bank_workspace_bytes(B, N, K);
  if (workspace_bytes < 2 * half || ((uintptr_t)workspace & 7)) return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (N + kOscChunk - 1) / kOscChunk;
  const double inv_sr = 1.0 / (double)sample_rate;
  double* sums = (double*)workspace;
  double* csum = (double*)((char*)workspace + half);
  hipLaunchKernelGGL(osc_chunk_sums_kernel, dim3(n_chunks, B), dim3(256), 0, st, frequency_envelopes, sums, N, K, n_chunks);
  hipLaunchKernelGGL(osc_chunk_prefix_kernel, dim3((K + 255) / 256, B), dim3(256), 0, st, sums, K, n_chunks, inv_sr);
  hipLaunchKernelGGL(osc_grad_freq_c_kernel, dim3(n_chunks, B), dim3(256), 0, st, frequency_envelopes, amplitude_envelopes,
                     grad_audio, sums, grad_frequency_envelopes, csum, N, K, n_chunks, inv_sr, (float)(sample_rate / 2.0));
  hipLaunchKernelGGL(osc_chunk_suffix_kernel, dim3((K + 255) / 256, B), dim3(256), 0, st, csum, K, n_chunks);
  hipLaunchKernelGGL(osc_grad_freq_suffix_kernel, dim3(n_chunks, B), dim3(256), 0, st, grad_frequency_envelopes, csum, N, K, n_chunks);
  return check_launch();
}

// =====================================================================================
// Validation-only "TF order" Harmonic.get_signal: reproduces the reference's fp32 op chain
// (core.harmonic_synthesis -> resample -> oscillator_bank, ddsp/core.py:1048-1111, 573-714,
// 912-962, 800-866) operation by operation, INCLUDING the strictly sequential fp32 phase
// accumulation of tf.cumsum (or angular_cumsum's chunked variant), so that full-length clips can be
// compared with the fp32-faithful oracle directly.  One lane per (batch row, harmonic), a serial
// loop over time: latency bound and slow by construction; it is never on the product path
// (flag DDSP_HARM_TF_SEQUENTIAL, used by the parity tests only).
// =====================================================================================
namespace ddsp {
__global__ __launch_bounds__(64) void harm_tf_order_kernel(
    const float* __restrict__ ctl_amp, const float* __restrict__ ctl_hd,
    const float* __restrict__ f0_all, float* __restrict__ audio /* pre-zeroed [B,N] */, int F, int K,
    int N, int hop, float sample_rate, float nyquist, int amp_linear, int angular) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * 64 + threadIdx.x;
  const bool live = k < K;
  const int kk = live ? k : K - 1;
  const float ratio = (float)(kk + 1);                    // tf.linspace(1, K, K)[k]
  const float* __restrict__ f0 = f0_all + (size_t)b * F;
  const float* __restrict__ amp = ctl_amp + (size_t)b * F;
  const float* __restrict__ hd = ctl_hd + (size_t)b * F * K;
  const float scale = (float)F / (float)N;                // legacy resize scale, fp32
  const float two_pi = 6.283185307179586f;                // fl32(2*pi)
  float phase = 0.0f, chunk_phase = 0.0f, offset = 0.0f, offset_sum = 0.0f;
  for (int t = 0; t < N; ++t) {
    // frequency envelope: resample(f0*ratio) 'linear' (core.py:1091,1102,613-621)
    const float pos = rn_mul((float)t, scale);
    const float lo = floorf(pos);
    const float lerp_w = rn_sub(pos, lo);
    const int lo_i = (int)lo, hi_i = min((int)ceilf(pos), F - 1);
    const float top = rn_mul(f0[lo_i], ratio), bottom = rn_mul(f0[hi_i], ratio);
    const float f = rn_add(top, rn_mul(rn_sub(bottom, top), lerp_w));
    // amplitude envelope: resample(amplitudes*distribution) 'window' or 'linear' (core.py:1097,1103)
    float a;
    if (amp_linear) {
      const float atop = rn_mul(amp[lo_i], hd[(size_t)lo_i * K + kk]);
      const float abot = rn_mul(amp[hi_i], hd[(size_t)hi_i * K + kk]);
      a = rn_add(atop, rn_mul(rn_sub(abot, atop), lerp_w));
    } else {
      const int j = t / hop, r = t - j * hop, j1 = min(j + 1, F - 1);
      // periodic Hann(2*hop) in fp32: tf.signal.hann_window (core.py:698)
      const float w_hi = 0.5f - 0.5f * cosf(two_pi * (float)(hop + r) / (float)(2 * hop));
      const float w_lo = 0.5f - 0.5f * cosf(two_pi * (float)r / (float)(2 * hop));
      const float x0 = rn_mul(amp[j], hd[(size_t)j * K + kk]);
      const float x1 = rn_mul(amp[j1], hd[(size_t)j1 * K + kk]);
      a = rn_add(rn_mul(x0, w_hi), rn_mul(x1, w_lo));      // overlap_and_add of 2 frames
    }
    if (f >= nyquist) a = 0.0f;                                        // remove_above_nyquist
    const float omega = rn_div(rn_mul(f, two_pi), sample_rate);   // core.py:947-948
    float ph;
    if (!angular) {
      phase = rn_add(phase, omega);                                 // tf.cumsum, sequential
      ph = phase;
    } else {                                                           // angular_cumsum, chunk 1000
      if (t % 1000 == 0) {
        if (t > 0) {
          // offsets: previous chunks' final phases mod 2pi, cumulatively summed then mod 2pi
          offset_sum = rn_add(offset_sum, fmodf(chunk_phase, two_pi));
          offset = fmodf(offset_sum, two_pi);
        }
        chunk_phase = 0.0f;
      }
      chunk_phase = rn_add(chunk_phase, omega);
      ph = fmodf(rn_add(chunk_phase, offset), two_pi);
    }
    float v = live ? rn_mul(a, sinf(ph)) : 0.0f;
    v = wave_sum_dpp(v);                                               // reduce_sum over harmonics
    if (threadIdx.x == 0) {
      if (gridDim.x == 1) audio[(size_t)b * N + t] = v;
      else atomicAdd(&audio[(size_t)b * N + t], v);
    }
  }
}
}  // namespace ddsp

extern "C" int ddsp_harmonic_signal_tf_order_f32(const float* ctl_amp, const float* ctl_hd,
                                                 const float* f0_hz, float* audio, int B, int F, int K,
                                                 int N, int sample_rate, unsigned flags, void* stream) {
  if (!ctl_amp || !ctl_hd || !f0_hz || !audio) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0 || N <= 0 || sample_rate <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  const int amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  if (!amp_linear && (N % F != 0)) return DDSP_ERR_BAD_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int kblocks = (K + 63) / 64;
  if (kblocks > 1 && hipMemsetAsync(audio, 0, (size_t)B * N * sizeof(float), st) != hipSuccess)
    return DDSP_ERR_LAUNCH;
  hipLaunchKernelGGL(harm_tf_order_kernel, dim3(kblocks, B), dim3(64), 0, st, ctl_amp, ctl_hd, f0_hz,
                     audio, F, K, N, amp_linear ? 1 : N / F, (float)sample_rate,
                     (float)(sample_rate / 2.0), amp_linear, (flags & DDSP_HARM_ANGULAR_CUMSUM) ? 1 : 0);
  return check_launch();
}

// The constant tables of the matrix-core kernels are made on the host and copied to the device - synchronously, with hipMalloc -
// the first time a shape needs them (harmonic_table.hip: wt_upload_fragments; harmonic_bwd_table.hip: bt_fragments;
// filtered_noise_general.hip: gi_matrix).  That is illegal inside a HIP-graph stream capture and synchronises the device
// (ADVICE r4): a caller that captures graphs, or that must not stall, calls this once per device and shape first.
extern "C" int ddsp_prepare(int n_harmonics, int n_noise_bands, int window_size) {
  if (n_harmonics > 0 && n_harmonics <= 200) {
    if (ddsp::harm_table_prepare(n_harmonics) != 0 || ddsp::harm_bwd_table_prepare(n_harmonics) != 0) return DDSP_ERR_LAUNCH;
  }
  if (n_noise_bands >= 2 && ddsp::noise_general_prepare(n_noise_bands, window_size) != 0) return DDSP_ERR_LAUNCH;
  return DDSP_OK;
}
