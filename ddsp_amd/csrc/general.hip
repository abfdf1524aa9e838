// General-shape kernels for gfx950 that complete the argument space of the hot path's own functions
// where the fast kernels are specialised (SURVEY.md section 8, rows a5-a7, a13 and the backward row):
//
//   ddsp_resample_ex_f32          core.resample, every method ('nearest', 'linear', 'cubic', 'window'),
//                                 add_endpoint True / False, up- and down-sampling (ddsp/core.py:573-714)
//   ddsp_fft_convolve_f32         core.fft_convolve for any crop: padding 'valid' or 'same', any
//                                 delay_compensation (ddsp/core.py:1382-1473, 1338-1379)
//   ddsp_harmonic_envelopes_f32   the frame-rate tensors core.harmonic_synthesis builds before it resamples:
//                                 k f0 (1 + harmonic_shifts) and amplitudes * harmonic_distribution
//                                 (ddsp/core.py:1080-1098, 1028-1045)
//   ddsp_harmonic_f0_grad_f32     dL/d f0_hz of Harmonic (what tf.GradientTape forms through tf.cumsum and
//                                 tf.sin, ddsp/core.py:950-960; trainers.py:162-171)
//   ddsp_exp_decay_ir_f32 (+ _backward)   effects.ExpDecayReverb's impulse response and its gradient with
//                                 respect to gain and decay (ddsp/effects.py:120-199)
//   ddsp_mix_f32, ddsp_sigmoid_f32        processors.Mix (ddsp/processors.py:180-233)
//
// These are generality paths: one thread per output value, HBM / L2 reads only, no LDS, no cross-lane
// traffic, no inline assembly.  Their cost is irrelevant next to the fused kernels of harmonic*.hip and
// filtered_noise.hip (no shipped gin config reaches them); what matters is that the reference's argument
// space has no holes.  Because they are written in plain HIP, tests/hip_emu compiles this very file for the
// host and the CPU test run checks every entry point against the oracle (tests/test_general_emulated.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ddsp_amd.h"
#include "noise_ir_geom.h"      // hann_denominator: tf.signal.hann_window's denominator

namespace ddsp {
namespace general {

constexpr int kThreads = 256;

// Individually rounded fp32 steps (see csrc/common.h: the HIP header's __fmul_rn is a plain `*` that hipcc
// contracts into FMAs).  g++ (the host build of tests/hip_emu) never contracts on x86-64 without -mfma.
#if defined(__clang__)
#define DDSP_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define DDSP_NO_CONTRACT
#endif
__device__ __forceinline__ float rn_mul(float a, float b) { DDSP_NO_CONTRACT return a * b; }
__device__ __forceinline__ float rn_add(float a, float b) { DDSP_NO_CONTRACT return a + b; }
__device__ __forceinline__ float rn_sub(float a, float b) { DDSP_NO_CONTRACT return a - b; }

__device__ __forceinline__ size_t global_thread() { return (size_t)blockIdx.x * kThreads + threadIdx.x; }
__device__ __forceinline__ size_t grid_threads() { return (size_t)gridDim.x * kThreads; }

// =====================================================================================
// core.resample (ddsp/core.py:573-642): tf.compat.v1.image.resize on [B,F,1,C] -> [B,N,1,C] for
// 'nearest' / 'linear' / 'cubic' (legacy kernels, half_pixel_centers=False, align_corners =
// not add_endpoint; the width axis is 1 -> 1, a copy), core.upsample_with_windows for 'window'.
// =====================================================================================
struct ResampleArgs {
  int F, N, C;
  int method;          // DDSP_RESAMPLE_*
  int align_corners;   // not add_endpoint
  int hop;             // 'window': N / n_intervals
  float scale;         // in fp32 as TF computes it: F/N, or (F-1)/(N-1) with align_corners and N > 1
};

// TF's bicubic coefficient table (A = -0.75, 1024 entries): the two polynomials in double on the
// float abscissa, rounded to float once, exactly as the table is filled.
__device__ __forceinline__ float cubic_near(int i) {      // |x| <= 1:  ((A+2)x - (A+3)) x^2 + 1
  const double A = -0.75;
  const double x = (double)((float)i * (1.0f / 1024.0f));
  return (float)(((A + 2.0) * x - (A + 3.0)) * x * x + 1.0);
}
__device__ __forceinline__ float cubic_far(int i) {       // 1 <= x <= 2:  ((A x - 5A) x + 8A) x - 4A
  const double A = -0.75;
  const double x = (double)((float)i * (1.0f / 1024.0f) + 1.0f);
  return (float)(((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A);
}

__global__ __launch_bounds__(kThreads) void resample_ex_kernel(const float* __restrict__ x,
                                                               float* __restrict__ out, ResampleArgs p) {
  const int b = blockIdx.y;
  const size_t total = (size_t)p.N * p.C;
  const float* __restrict__ xb = x + (size_t)b * p.F * p.C;
  float* __restrict__ ob = out + (size_t)b * total;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const int t = (int)(i / p.C), c = (int)(i - (size_t)t * p.C);
    float v;
    if (p.method == DDSP_RESAMPLE_WINDOW) {
      // upsample_with_windows (core.py:645-714) in closed form: frame j = t / hop fades out with
      // Hann(2 hop)[hop + r], frame j + 1 fades in with Hann(2 hop)[r]; with add_endpoint the
      // appended frame repeats the last one
      const int j = t / p.hop, r = t - j * p.hop;
      const int hi = min(j + 1, p.F - 1);
      const float w = 0.5f - 0.5f * cospif((float)r / (float)p.hop);
      v = xb[(size_t)j * p.C + c] * (1.0f - w) + xb[(size_t)hi * p.C + c] * w;
    } else {
      const float pos = rn_mul((float)t, p.scale);  // legacy scaler: out * scale (rounded product, never an FMA)
      if (p.method == DDSP_RESAMPLE_NEAREST) {
        const int src = min((int)(p.align_corners ? roundf(pos) : floorf(pos)), p.F - 1);
        v = xb[(size_t)src * p.C + c];
      } else if (p.method == DDSP_RESAMPLE_LINEAR) {
        const float lo = floorf(pos);
        const int lo_i = min(max((int)lo, 0), p.F - 1), hi_i = min((int)ceilf(pos), p.F - 1);
        const float top = xb[(size_t)lo_i * p.C + c], bottom = xb[(size_t)hi_i * p.C + c];
        v = rn_add(top, rn_mul(rn_sub(bottom, top), rn_sub(pos, lo)));
      } else {                                          // DDSP_RESAMPLE_CUBIC
        const float lo = floorf(pos);
        const int src = (int)lo;
        const int offset = (int)lrintf(rn_mul(rn_sub(pos, lo), 1024.0f));
        const float w0 = cubic_far(offset), w1 = cubic_near(offset);
        const float w2 = cubic_near(1024 - offset), w3 = cubic_far(1024 - offset);
        const int i0 = min(max(src - 1, 0), p.F - 1), i1 = min(max(src, 0), p.F - 1);
        const int i2 = min(max(src + 1, 0), p.F - 1), i3 = min(max(src + 2, 0), p.F - 1);
        // Interpolate1D: v0 w0 + v1 w1 + v2 w2 + v3 w3, fp32, left to right, no contraction
        float acc = rn_mul(xb[(size_t)i0 * p.C + c], w0);
        acc = rn_add(acc, rn_mul(xb[(size_t)i1 * p.C + c], w1));
        acc = rn_add(acc, rn_mul(xb[(size_t)i2 * p.C + c], w2));
        v = rn_add(acc, rn_mul(xb[(size_t)i3 * p.C + c], w3));
      }
    }
    ob[i] = v;
  }
}

// The adjoint of resample_ex_kernel (what tf.GradientTape forms through core.resample, trainers.py:162-171): resampling is
// linear in its input, grad_in[j] = sum_t W[t][j] grad_out[t].  One thread per (frame j, channel c) GATHERS: it walks the output
// samples whose interpolation can touch frame j (a conservative range: j +- 2 frames' worth and one sample more), recomputes each
// one's taps and weights exactly as the forward kernel does - same fp32 positions, same clamped indices - and adds the weights
// that land on j.  No atomics, a fixed order of additions: deterministic.
__global__ __launch_bounds__(kThreads) void resample_ex_backward_kernel(const float* __restrict__ gout,
                                                                        float* __restrict__ gin, ResampleArgs p) {
  const int b = blockIdx.y;
  const size_t total = (size_t)p.F * p.C;
  const float* __restrict__ gb = gout + (size_t)b * p.N * p.C;
  float* __restrict__ ib = gin + (size_t)b * total;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const int j = (int)(i / p.C), c = (int)(i - (size_t)j * p.C);
    float acc = 0.0f;
    if (p.method == DDSP_RESAMPLE_WINDOW) {
      const int t_lo = max((j - 1) * p.hop, 0);
      const int t_hi = (int)min((long)(j + 1) * p.hop, (long)p.N);
#pragma unroll 4
      for (int t = t_lo; t < t_hi; ++t) {
        const int jj = t / p.hop, r = t - jj * p.hop;
        const int hi = min(jj + 1, p.F - 1);
        const float w = 0.5f - 0.5f * cospif((float)r / (float)p.hop);
        const float g = gb[(size_t)t * p.C + c];
        if (jj == j) acc = fmaf(g, 1.0f - w, acc);
        if (hi == j) acc = fmaf(g, w, acc);
      }
    } else {
      long t_lo = (long)floorf((float)(j - 2) / p.scale) - 1, t_hi = (long)ceilf((float)(j + 2) / p.scale) + 1;
      if (j == 0 || t_lo < 0) t_lo = 0;
      if (j == p.F - 1 || t_hi > p.N - 1) t_hi = p.N - 1;
#pragma unroll 4
      for (long t = t_lo; t <= t_hi; ++t) {
        const float pos = rn_mul((float)t, p.scale);
        const float g = gb[(size_t)t * p.C + c];
        if (p.method == DDSP_RESAMPLE_NEAREST) {
          const int src = min((int)(p.align_corners ? roundf(pos) : floorf(pos)), p.F - 1);
          if (src == j) acc += g;
        } else if (p.method == DDSP_RESAMPLE_LINEAR) {
          const float lo = floorf(pos);
          const int lo_i = min(max((int)lo, 0), p.F - 1), hi_i = min((int)ceilf(pos), p.F - 1);
          const float frac = rn_sub(pos, lo);
          if (lo_i == j) acc = fmaf(g, 1.0f - frac, acc);
          if (hi_i == j) acc = fmaf(g, frac, acc);
        } else {                                          // DDSP_RESAMPLE_CUBIC
          const float lo = floorf(pos);
          const int src = (int)lo;
          const int offset = (int)lrintf(rn_mul(rn_sub(pos, lo), 1024.0f));
          const int i0 = min(max(src - 1, 0), p.F - 1), i1 = min(max(src, 0), p.F - 1);
          const int i2 = min(max(src + 1, 0), p.F - 1), i3 = min(max(src + 2, 0), p.F - 1);
          if (i0 == j) acc = fmaf(g, cubic_far(offset), acc);
          if (i1 == j) acc = fmaf(g, cubic_near(offset), acc);
          if (i2 == j) acc = fmaf(g, cubic_near(1024 - offset), acc);
          if (i3 == j) acc = fmaf(g, cubic_far(1024 - offset), acc);
        }
      }
    }
    ib[i] = acc;
  }
}

// =====================================================================================
// core.fft_convolve (ddsp/core.py:1382-1473) as the direct time-varying FIR it equals:
//   z[m] = sum_i x[i] h_{frame(i)}[m - i]   (framed FFT products, overlap-added)
//   out[n] = z[n + start], n < n_out        (crop_and_compensate_delay, :1338-1379)
// One thread per output sample; taps and samples come from L2.
// =====================================================================================
struct FirArgs {
  int N, F, L, frame_size, start, n_out;
  size_t ir_batch_stride;   // F*L, or 0 when the impulse response is broadcast over the batch
};

__global__ __launch_bounds__(kThreads) void tv_fir_any_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ ir,
                                                              float* __restrict__ out, FirArgs p) {
  const int b = blockIdx.y;
  const float* __restrict__ xb = x + (size_t)b * p.N;
  const float* __restrict__ hb = ir + (size_t)b * p.ir_batch_stride;
  float* __restrict__ ob = out + (size_t)b * p.n_out;
  for (size_t n = global_thread(); n < (size_t)p.n_out; n += grid_threads()) {
    const long m = (long)n + p.start;                       // index into the un-cropped convolution
    const long i_lo = max(m - (long)(p.L - 1), 0L), i_hi = min(m, (long)p.N - 1);
    float acc = 0.0f;
    if (i_lo <= i_hi) {
      const int f_lo = (int)(i_lo / p.frame_size), f_hi = (int)(i_hi / p.frame_size);
      for (int f = f_lo; f <= f_hi; ++f) {
        const long ia = max((long)f * p.frame_size, i_lo);
        const long ib = min((long)(f + 1) * p.frame_size - 1, i_hi);
        const float* __restrict__ h = hb + (size_t)f * p.L;
        for (long i = ia; i <= ib; ++i) acc = fmaf(xb[i], h[m - i], acc);
      }
    }
    ob[n] = acc;
  }
}

// =====================================================================================
// The frame-rate tensors of core.harmonic_synthesis (ddsp/core.py:1080-1098):
//   harmonic_frequencies = (frequencies * [1..K]) * (1 + harmonic_shifts)     fp32, in this order
//   harmonic_amplitudes  = amplitudes * harmonic_distribution   (or amplitudes, broadcast over K)
// =====================================================================================
__global__ __launch_bounds__(kThreads) void harmonic_envelopes_kernel(
    const float* __restrict__ amplitudes /*[R]*/, const float* __restrict__ hd /*[R,K] or null*/,
    const float* __restrict__ f0 /*[R]*/, const float* __restrict__ shifts /*[R,K] or null*/,
    float* __restrict__ freq_out /*[R,K]*/, float* __restrict__ amp_out /*[R,K]*/, size_t rows, int K) {
  const size_t total = rows * (size_t)K;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const size_t row = i / K;
    const int k = (int)(i - row * K);
    float fk = rn_mul(f0[row], (float)(k + 1));
    if (shifts != nullptr) fk = rn_mul(fk, rn_add(1.0f, shifts[i]));
    freq_out[i] = fk;
    amp_out[i] = (hd != nullptr) ? rn_mul(amplitudes[row], hd[i]) : amplitudes[row];
  }
}

// The adjoint of harmonic_frequencies = f0 [1..K] (1 + shifts) with respect to f0: dL/d f0[row] = sum_k dL/d hf[row,k] (k+1) (1 + shift)
__global__ __launch_bounds__(kThreads) void harmonic_frequencies_backward_kernel(const float* __restrict__ ghf /*[R,K]*/,
                                                                                 const float* __restrict__ shifts /*[R,K] or null*/,
                                                                                 float* __restrict__ gf0 /*[R]*/, size_t rows, int K) {
  for (size_t row = global_thread(); row < rows; row += grid_threads()) {
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      float w = (float)(k + 1);
      if (shifts != nullptr) w *= 1.0f + shifts[row * K + k];
      acc += (double)ghf[row * K + k] * (double)w;
    }
    gf0[row] = (float)acc;
  }
}

// out[i] = x[i] * scale[0]: the upstream scalar of a loss's backward pass applied to the stored dL/d audio (the chain
// rule through a scalar; what tf.GradientTape does implicitly, ddsp/training/trainers.py:162-171)
__global__ __launch_bounds__(kThreads) void scale_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         float* __restrict__ out, size_t n) {
  const float s = scale[0];
  for (size_t i = global_thread(); i < n; i += grid_threads()) out[i] = x[i] * s;
}

// =====================================================================================
// core.harmonic_oscillator_bank (ddsp/core.py:966-1025) on AUDIO-RATE inputs: one fundamental per clip,
//   omega = f 2 pi / sr,  phase[n] = cumsum(omega)[n] + initial_phase,  audio[n] = sum_k A[n,k] sin(k phase[n]),
//   final_phase = phase[N-1]  (with use_angular_cumsum, the reference's default here, wrapped to [0, 2 pi) + initial_phase
//   as core.angular_cumsum leaves it; without, the plain sum).
// No Nyquist mask, as in the reference.  The scan runs in fp64 revolutions: chunk sums (one thread per chunk of 256
// samples), a serial exclusive prefix per clip, then one thread per sample (its part of the chunk summed again:
// 256 L2-resident adds - a generality path; the frame-rate closed forms of harmonic.hip are the fast one).
// =====================================================================================
constexpr int kHobChunk = 256;

__global__ __launch_bounds__(kThreads) void hob_chunk_sums_kernel(const float* __restrict__ freq /*[B,N]*/,
                                                                  double* __restrict__ sums /*[B,C]*/, int B, int N, int C) {
  const size_t total = (size_t)B * C;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const size_t b = i / C;
    const int c = (int)(i - b * C);
    const float* __restrict__ f = freq + b * N;
    const int n1 = min((c + 1) * kHobChunk, N);
    double acc = 0.0;
    for (int n = c * kHobChunk; n < n1; ++n) acc += (double)f[n];
    sums[i] = acc;
  }
}

// sums[b][c] <- cycles before chunk c (wrapped to [0,1)); total[b] <- cycles of the whole clip (not wrapped)
__global__ __launch_bounds__(kThreads) void hob_prefix_kernel(double* __restrict__ sums /*[B,C]*/,
                                                              double* __restrict__ total /*[B]*/, int B, int C,
                                                              double inv_sr) {
  for (size_t b = global_thread(); b < (size_t)B; b += grid_threads()) {
    double* __restrict__ s = sums + b * C;
    double acc = 0.0, all = 0.0;
    for (int c = 0; c < C; ++c) {
      const double v = s[c] * inv_sr;
      s[c] = acc;
      acc += v;
      acc -= floor(acc);
      all += v;
    }
    total[b] = all;
  }
}

__global__ __launch_bounds__(kThreads) void hob_synth_kernel(
    const float* __restrict__ freq /*[B,N]*/, const float* __restrict__ amps /*[B,N,K]*/,
    const float* __restrict__ initial_phase /*[B] radians, or null*/, const double* __restrict__ prefix /*[B,C]*/,
    const double* __restrict__ total /*[B]*/, float* __restrict__ audio /*[B,N]*/, float* __restrict__ final_phase /*[B]*/,
    int B, int N, int K, int C, double inv_sr, int angular) {
  const double kInv2Pi = 0.15915494309189533577, k2Pi = 6.283185307179586476925;
  const size_t count = (size_t)B * N;
  for (size_t i = global_thread(); i < count; i += grid_threads()) {
    const size_t b = i / N;
    const int n = (int)(i - b * N);
    const float* __restrict__ f = freq + b * N;
    const int c = n / kHobChunk;
    double acc = 0.0;
    for (int m = c * kHobChunk; m <= n; ++m) acc += (double)f[m];          // inclusive: tf.cumsum starts at omega_0
    const double phi0 = initial_phase ? (double)initial_phase[b] * kInv2Pi : 0.0;
    double theta = prefix[b * C + c] + acc * inv_sr + phi0;                  // revolutions
    theta -= floor(theta);
    const float* __restrict__ a = amps + i * K;
    float out = 0.0f;
    for (int k = 0; k < K; ++k) {
      const double kt = (double)(k + 1) * theta;
      out = fmaf(a[k], sinpif(2.0f * (float)(kt - floor(kt))), out);
    }
    audio[i] = out;
    if (n == N - 1) {
      const double cyc = total[b];
      final_phase[b] = (float)((angular ? (cyc - floor(cyc)) : cyc) * k2Pi + (initial_phase ? (double)initial_phase[b] : 0.0));
    }
  }
}

// =====================================================================================
// dL/d f0_hz of Harmonic.  With phase_k[n] = (2 pi / sr) k cumsum(f_env)[n] and f_env = U f0 (the legacy
// bilinear resize, linear in f0; the Nyquist masks have zero gradient, as tf.where gives them):
//   dL/d f_env[t] = (2 pi / sr) sum_{n >= t} c[n],   c[n] = g[n] sum_k k A_k[n] m_k[n] cos(phase_k[n])
//   dL/d f0[j]    = sum_t U[t, j] dL/d f_env[t]
// U[t, j] is 1 - r/hop on frame j's own samples and r/hop on frame j-1's (r = t % hop; the last frame
// holds), so with per-frame sums T_j = sum c, P_j = sum c W1(r), Q_j = sum c W2(r), where
// W1(r) = sum_{r' <= r} (1 - r'/hop), W2(r) = sum_{r' <= r} r'/hop, and S_j = sum_{j' >= j} T_j':
//   dL/d f0[j] = (2 pi / sr) [ P_j + S_{j+1} (hop + 1)/2  +  (j >= 1: Q_{j-1} + S_j (hop - 1)/2)
//                              +  (j == F-1: Q_{F-1}) ]
// Four launches: the fp64 phase of the fundamental at every frame start (serial per row, as in
// harmonic.hip), c[n] per sample, the three sums per frame, the suffix scan per row.
// =====================================================================================
struct F0GradArgs {
  int B, F, K, N, hop;
  float nyquist;
  double inv_sr;
  int amp_linear;
};

__global__ __launch_bounds__(kThreads) void f0grad_phase_kernel(const float* __restrict__ f0 /*[B,F]*/,
                                                                double* __restrict__ theta0 /*[B,F]*/,
                                                                F0GradArgs p) {
  for (size_t b = global_thread(); b < (size_t)p.B; b += grid_threads()) {
    const float* __restrict__ fr = f0 + b * p.F;
    double* __restrict__ th = theta0 + b * p.F;
    double acc = 0.0;                                   // revolutions, wrapped to [0,1)
    for (int j = 0; j < p.F; ++j) {
      th[j] = acc;
      const double fj = (double)fr[j], fj1 = (double)fr[min(j + 1, p.F - 1)];
      acc += ((double)p.hop * fj + (fj1 - fj) * (0.5 * (double)(p.hop - 1))) * p.inv_sr;
      acc -= floor(acc);
    }
  }
}

__global__ __launch_bounds__(kThreads) void f0grad_c_kernel(
    const float* __restrict__ ctl_amp /*[B,F]*/, const float* __restrict__ ctl_hd /*[B,F,K]*/,
    const float* __restrict__ f0 /*[B,F]*/, const double* __restrict__ theta0 /*[B,F]*/,
    const float* __restrict__ grad_audio /*[B,N]*/, float* __restrict__ c_out /*[B,N]*/, F0GradArgs p) {
  const size_t total = (size_t)p.B * p.N;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const size_t b = i / p.N;
    const int n = (int)(i - b * p.N);
    const int j = n / p.hop, r = n - j * p.hop, j1 = min(j + 1, p.F - 1);
    const float fj = f0[b * p.F + j], fj1 = f0[b * p.F + j1];
    // inclusive cumsum of the interpolated frequency, in revolutions (harm_synth_kernel's closed form)
    const double rr = (double)r;
    const double wj = (double)fj * p.inv_sr;
    const double dw = ((double)fj1 - (double)fj) * p.inv_sr * (0.5 / (double)p.hop);
    double cyc = theta0[b * p.F + j] + (rr + 1.0) * (wj + dw * rr);
    cyc -= floor(cyc);
    const float lerp = (float)r / (float)p.hop;
    const float w_next = p.amp_linear ? lerp : 0.5f - 0.5f * cospif(lerp);
    const float w_cur = 1.0f - w_next;
    const float a_cur = ctl_amp[b * p.F + j] * w_cur, a_next = ctl_amp[b * p.F + j1] * w_next;
    const float* __restrict__ h0 = ctl_hd + (b * p.F + j) * p.K;
    const float* __restrict__ h1 = ctl_hd + (b * p.F + j1) * p.K;
    float acc = 0.0f;
    for (int k = 0; k < p.K; ++k) {
      const float kf = (float)(k + 1);
      // the audio-rate mask of oscillator_bank on the interpolated frequency, TF's fp32 op order
      const float top = rn_mul(fj, kf), bot = rn_mul(fj1, kf);
      const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp));
      if (fk >= p.nyquist) continue;
      double ph = cyc * (double)(k + 1);
      ph -= floor(ph);
      const float amp = fmaf(a_cur, h0[k], a_next * h1[k]);
      acc = fmaf(kf * amp, __builtin_amdgcn_cosf((float)ph), acc);
    }
    c_out[i] = grad_audio[i] * acc;
  }
}

__global__ __launch_bounds__(kThreads) void f0grad_frame_kernel(const float* __restrict__ c /*[B,N]*/,
                                                                double* __restrict__ sums /*[B,F,3]*/,
                                                                F0GradArgs p) {
  const size_t total = (size_t)p.B * p.F;
  const double inv_2hop = 0.5 / (double)p.hop;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const size_t b = i / p.F;
    const int j = (int)(i - b * p.F);
    const float* __restrict__ cf = c + b * p.N + (size_t)j * p.hop;
    double t = 0.0, pw = 0.0, qw = 0.0;
    for (int r = 0; r < p.hop; ++r) {
      const double v = (double)cf[r], rr = (double)r;
      const double w2 = rr * (rr + 1.0) * inv_2hop;      // sum_{r' <= r} r'/hop
      t += v;
      pw += v * ((rr + 1.0) - w2);                        // sum_{r' <= r} (1 - r'/hop)
      qw += v * w2;
    }
    sums[i * 3 + 0] = t; sums[i * 3 + 1] = pw; sums[i * 3 + 2] = qw;
  }
}

__global__ __launch_bounds__(kThreads) void f0grad_scan_kernel(const double* __restrict__ sums /*[B,F,3]*/,
                                                               float* __restrict__ grad_f0 /*[B,F]*/,
                                                               F0GradArgs p) {
  const double scale = 6.283185307179586476925 * p.inv_sr;
  const double w_own = 0.5 * (double)(p.hop + 1), w_prev = 0.5 * (double)(p.hop - 1);
  for (size_t b = global_thread(); b < (size_t)p.B; b += grid_threads()) {
    const double* __restrict__ s = sums + b * p.F * 3;
    float* __restrict__ g = grad_f0 + b * p.F;
    double suffix_next = 0.0;                            // S_{j+1}
    for (int j = p.F - 1; j >= 0; --j) {
      const double suffix = suffix_next + s[j * 3 + 0];  // S_j
      double d = s[j * 3 + 1] + suffix_next * w_own;
      if (j >= 1) d += s[(j - 1) * 3 + 2] + suffix * w_prev;
      if (j == p.F - 1) d += s[j * 3 + 2];
      g[j] = (float)(scale * d);
      suffix_next = suffix;
    }
  }
}

// =====================================================================================
// effects.ExpDecayReverb._get_ir (ddsp/effects.py:144-151): an exponentially decaying burst of noise
//   ir[b, i] = G_b * exp(-(2 + exp(decay_b)) * t_i) * noise[i],  t = linspace(0, 1, L),  G = scale_fn(gain)
// and its gradient with respect to gain and decay (two-stage sums, fp64):
//   dL/d gain_b  = G'(gain_b) * sum_i g[b,i] E_b(t_i) noise[i]
//   dL/d decay_b = -exp(decay_b) * G_b * sum_i g[b,i] E_b(t_i) noise[i] t_i
// =====================================================================================
constexpr int kDecayPartials = 256;

__device__ __forceinline__ float linspace01(int i, int L, float step) {
  return (i == L - 1 && L > 1) ? 1.0f : (float)i * step;      // tf.linspace ends on `stop` exactly
}
// core.exp_sigmoid with its default constants (core.py:386-404): 2 sigmoid(x)^log(10) + 1e-7
__device__ __forceinline__ float exp_sigmoid_default(float x) {
  const float softplus_neg = (x >= 0.0f) ? log1pf(expf(-x)) : (-x + log1pf(expf(x)));   // log(1 + e^-x)
  return 2.0f * expf(-2.302585092994046f * softplus_neg) + 1e-7f;
}

__global__ __launch_bounds__(kThreads) void exp_decay_ir_kernel(const float* __restrict__ gain /*[Bg]*/,
                                                                const float* __restrict__ decay /*[Bg]*/,
                                                                const float* __restrict__ noise /*[L]*/,
                                                                float* __restrict__ ir /*[B,L]*/, int B, int L,
                                                                int scale_exp_sigmoid) {
  const size_t total = (size_t)B * L;
  const float step = L > 1 ? 1.0f / (float)(L - 1) : 0.0f;
  for (size_t idx = global_thread(); idx < total; idx += grid_threads()) {
    const int b = (int)(idx / L), i = (int)(idx - (size_t)b * L);
    const float g = scale_exp_sigmoid ? exp_sigmoid_default(gain[b]) : gain[b];
    const float decay_exponent = 2.0f + expf(decay[b]);
    ir[idx] = g * expf(-decay_exponent * linspace01(i, L, step)) * noise[i];
  }
}

__global__ __launch_bounds__(kThreads) void exp_decay_bwd_partial_kernel(
    const float* __restrict__ decay /*[B]*/, const float* __restrict__ noise /*[L]*/,
    const float* __restrict__ grad_ir /*[B,L]*/, double* __restrict__ partial /*[B,kDecayPartials,2]*/, int B,
    int L) {
  const size_t total = (size_t)B * kDecayPartials;
  const float step = L > 1 ? 1.0f / (float)(L - 1) : 0.0f;
  for (size_t idx = global_thread(); idx < total; idx += grid_threads()) {
    const int b = (int)(idx / kDecayPartials), q = (int)(idx - (size_t)b * kDecayPartials);
    const float decay_exponent = 2.0f + expf(decay[b]);
    double s0 = 0.0, s1 = 0.0;
    for (int i = q; i < L; i += kDecayPartials) {
      const float t = linspace01(i, L, step);
      const double v = (double)(grad_ir[(size_t)b * L + i] * noise[i]) * (double)expf(-decay_exponent * t);
      s0 += v;
      s1 += v * (double)t;
    }
    partial[idx * 2 + 0] = s0;
    partial[idx * 2 + 1] = s1;
  }
}

__global__ __launch_bounds__(kThreads) void exp_decay_bwd_finish_kernel(
    const float* __restrict__ gain, const float* __restrict__ decay, const double* __restrict__ partial,
    float* __restrict__ grad_gain, float* __restrict__ grad_decay, int B, int scale_exp_sigmoid) {
  for (size_t b = global_thread(); b < (size_t)B; b += grid_threads()) {
    double s0 = 0.0, s1 = 0.0;
    for (int q = 0; q < kDecayPartials; ++q) {
      s0 += partial[(b * kDecayPartials + q) * 2 + 0];
      s1 += partial[(b * kDecayPartials + q) * 2 + 1];
    }
    double g = (double)gain[b], dg = 1.0;
    if (scale_exp_sigmoid) {
      // d/dx exp_sigmoid(x) = log(10) (y - threshold) (1 - sigmoid(x))
      const double x = g;
      const double sig = x >= 0.0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x));
      const double y = (double)exp_sigmoid_default((float)x);
      dg = 2.302585092994046 * (y - 1e-7) * (1.0 - sig);
      g = y;
    }
    grad_gain[b] = (float)(dg * s0);
    grad_decay[b] = (float)(-exp((double)decay[b]) * g * s1);
  }
}

// =====================================================================================
// processors.Mix (ddsp/processors.py:180-233): constant-power crossfade.
//   get_controls: mix_level = sigmoid(nn_out_mix_level) (then core.resample to the signals' length)
//   get_signal:   out = sqrt(|m|) * signal_one + (1 - sqrt(|m - 1|)) * signal_two,  m [rows] per time step,
//                 signals [rows, C]
// =====================================================================================
__global__ __launch_bounds__(kThreads) void sigmoid_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           size_t n) {
  for (size_t i = global_thread(); i < n; i += grid_threads()) {
    const float x = in[i];
    const float e = expf(-fabsf(x));
    out[i] = (x >= 0.0f) ? 1.0f / (1.0f + e) : e / (1.0f + e);
  }
}

__global__ __launch_bounds__(kThreads) void mix_kernel(const float* __restrict__ signal_one,
                                                       const float* __restrict__ signal_two,
                                                       const float* __restrict__ mix_level, float* __restrict__ out,
                                                       size_t rows, int C) {
  const size_t total = rows * (size_t)C;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const float m = mix_level[i / C];
    const float level_one = sqrtf(fabsf(m));
    const float level_two = 1.0f - sqrtf(fabsf(m - 1.0f));
    out[i] = rn_add(rn_mul(level_one, signal_one[i]), rn_mul(level_two, signal_two[i]));
  }
}

// The adjoints of the two (round 6: processors.Mix under torch.autograd ran on torch arithmetic; VERDICT r5 "missing" #5).
// grad_in = grad_out s (1 - s), s = sigmoid(in) recomputed
__global__ __launch_bounds__(kThreads) void sigmoid_backward_kernel(const float* __restrict__ in, const float* __restrict__ grad_out,
                                                                    float* __restrict__ grad_in, size_t n) {
  for (size_t i = global_thread(); i < n; i += grid_threads()) {
    const float x = in[i];
    const float e = expf(-fabsf(x));
    const float d = 1.0f + e;
    grad_in[i] = grad_out[i] * (e / (d * d));                   // s (1 - s) = e / (1 + e)^2, e = exp(-|x|): no cancellation in either tail
  }
}
// grad_one = sqrt|m| g, grad_two = (1 - sqrt|m - 1|) g, grad_level[r] = sum_c g (one sign(m) / (2 sqrt|m|) - two sign(m - 1) / (2 sqrt|m - 1|))
// (d sqrt|x| / dx as tf.sqrt and tf.abs differentiate it: sign(x) / (2 sqrt|x|), NaN at x = 0 there as here); one thread per
// row r walks its C channels in order: no atomics, the same bits every run.  Null outputs are skipped.
__global__ __launch_bounds__(kThreads) void mix_backward_kernel(const float* __restrict__ signal_one, const float* __restrict__ signal_two,
                                                                const float* __restrict__ mix_level, const float* __restrict__ grad_out,
                                                                float* __restrict__ grad_one, float* __restrict__ grad_two,
                                                                float* __restrict__ grad_level, size_t rows, int C) {
  for (size_t r = global_thread(); r < rows; r += grid_threads()) {
    const float m = mix_level[r], m1 = m - 1.0f;
    const float s0 = sqrtf(fabsf(m)), s1 = sqrtf(fabsf(m1));
    const float level_one = s0, level_two = 1.0f - s1;
    const float sg0 = (m > 0.0f) ? 1.0f : (m < 0.0f ? -1.0f : 0.0f), sg1 = (m1 > 0.0f) ? 1.0f : (m1 < 0.0f ? -1.0f : 0.0f);
    const float d0 = sg0 * (0.5f / s0), d1 = sg1 * (0.5f / s1);
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
      const size_t i = r * (size_t)C + c;
      const float g = grad_out[i];
      if (grad_one) grad_one[i] = level_one * g;
      if (grad_two) grad_two[i] = level_two * g;
      if (grad_level) acc += g * (signal_one[i] * d0 - signal_two[i] * d1);
    }
    if (grad_level) grad_level[r] = acc;
  }
}

static inline unsigned grid_for(size_t n, unsigned cap = 256 * 32) {
  size_t g = (n + kThreads - 1) / kThreads;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
static inline int check_launch() { return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH; }
static inline size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

}  // namespace general
}  // namespace ddsp

using namespace ddsp::general;

extern "C" int ddsp_resample_ex_f32(const float* x, float* out, int B, int F, int N, int C, int method,
                                    int add_endpoint, void* stream) {
  if (!x || !out) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || N <= 0 || C <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (method < DDSP_RESAMPLE_NEAREST || method > DDSP_RESAMPLE_WINDOW) return DDSP_ERR_BAD_SHAPE;
  ResampleArgs p;
  p.F = F; p.N = N; p.C = C; p.method = method; p.align_corners = add_endpoint ? 0 : 1;
  p.hop = 1;
  if (method == DDSP_RESAMPLE_WINDOW) {
    // core.py:677-693: upsampling only, N divisible by the number of intervals
    const int n_intervals = add_endpoint ? F : F - 1;
    if (n_intervals <= 0 || n_intervals + 1 >= N || N % n_intervals != 0) return DDSP_ERR_BAD_SHAPE;
    p.hop = N / n_intervals;
  }
  p.scale = (p.align_corners && N > 1) ? (float)(F - 1) / (float)(N - 1) : (float)F / (float)N;
  const dim3 grid(grid_for((size_t)N * C, 2048), (unsigned)B);
  hipLaunchKernelGGL(resample_ex_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, x, out, p);
  return check_launch();
}

extern "C" int ddsp_resample_ex_backward_f32(const float* grad_out, float* grad_in, int B, int F, int N, int C, int method,
                                             int add_endpoint, void* stream) {
  if (!grad_out || !grad_in) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || N <= 0 || C <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (method < DDSP_RESAMPLE_NEAREST || method > DDSP_RESAMPLE_WINDOW) return DDSP_ERR_BAD_SHAPE;
  ResampleArgs p;
  p.F = F; p.N = N; p.C = C; p.method = method; p.align_corners = add_endpoint ? 0 : 1;
  p.hop = 1;
  if (method == DDSP_RESAMPLE_WINDOW) {
    const int n_intervals = add_endpoint ? F : F - 1;
    if (n_intervals <= 0 || n_intervals + 1 >= N || N % n_intervals != 0) return DDSP_ERR_BAD_SHAPE;
    p.hop = N / n_intervals;
  }
  p.scale = (p.align_corners && N > 1) ? (float)(F - 1) / (float)(N - 1) : (float)F / (float)N;
  const dim3 grid(grid_for((size_t)F * C, 2048), (unsigned)B);
  hipLaunchKernelGGL(resample_ex_backward_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out, grad_in, p);
  return check_launch();
}

// =====================================================================================
// core.apply_window_to_impulse_response (ddsp/core.py:1477-1531) on its own, for any impulse-response length (the fused IR
// designs of FilteredNoise know it for irfft outputs only): zero-phase (or causal -> fftshift first) responses [rows, L0] ->
// Hann-windowed, causal, cropped responses [rows, L].  Index arithmetic of the reference's concat / fftshift calls, one
// element per thread.
// =====================================================================================
struct WinIrArgs { long rows; int L0, L, ws, padding, half, first_len, causal; };

__global__ __launch_bounds__(kThreads) void window_ir_kernel(const float* __restrict__ ir, float* __restrict__ out, WinIrArgs p) {
  const size_t total = (size_t)p.rows * p.L;
  for (size_t i = global_thread(); i < total; i += grid_threads()) {
    const long row = (long)(i / p.L);
    const int kappa = (int)(i - (size_t)row * p.L);
    int j;                                                       // index into the windowed zero-phase response
    if (p.padding > 0) j = kappa < p.first_len ? p.L0 - p.half + 2 + kappa : kappa - p.first_len;      // concat(w[L0-half+2:], w[:half+1])
    else j = (kappa - p.L0 / 2 + p.L0) % p.L0;                                                       // fftshift
    int wi;                                                      // window index at j, or -1 (the zero padding)
    if (p.padding > 0) wi = j < p.ws - p.half ? p.half + j : (j >= p.L0 - p.half ? j - (p.L0 - p.half) : -1);
    else wi = (j - p.ws / 2 + p.ws) % p.ws;                                                           // fftshift(window)
    const int src = p.causal ? (j - p.L0 / 2 + p.L0) % p.L0 : j;                                      // causal input: fftshift first
    // tf.signal.hann_window(ws): denominator ws for even ws, ws - 1 for odd ws; a window of one sample is [1.0] (noise_ir_geom.h)
    const float w = wi < 0 ? 0.0f : (p.ws == 1 ? 1.0f : 0.5f - 0.5f * cospif(2.0f * (float)wi / (float)ddsp::hann_denominator(p.ws)));
    out[i] = w * ir[(size_t)row * p.L0 + src];
  }
}

extern "C" int ddsp_window_impulse_response_size(int L0, int window_size) {
  if (L0 <= 0) return DDSP_ERR_BAD_SHAPE;
  const int ws = (window_size <= 0 || window_size > L0) ? L0 : window_size;
  if (ws == L0) return L0;
  const int half = (ws + 1) / 2;
  const int first_len = L0 - (L0 - half + 2) > 0 ? L0 - (L0 - half + 2) : 0;
  return first_len + (half + 1 < L0 ? half + 1 : L0);
}

extern "C" int ddsp_apply_window_to_impulse_response_f32(const float* impulse_response, float* out, long rows, int L0,
                                                         int window_size, int causal, void* stream) {
  if (!impulse_response || !out) return DDSP_ERR_NULL_POINTER;
  if (rows <= 0 || L0 <= 0) return DDSP_ERR_BAD_SHAPE;
  WinIrArgs p;
  p.rows = rows; p.L0 = L0; p.causal = causal ? 1 : 0;
  p.ws = (window_size <= 0 || window_size > L0) ? L0 : window_size;
  p.padding = L0 - p.ws;
  p.half = (p.ws + 1) / 2;
  p.first_len = p.half > 2 ? p.half - 2 : 0;
  p.L = ddsp_window_impulse_response_size(L0, window_size);
  hipLaunchKernelGGL(window_ir_kernel, dim3(grid_for((size_t)rows * p.L)), dim3(kThreads), 0, (hipStream_t)stream,
                     impulse_response, out, p);
  return check_launch();
}

extern "C" int ddsp_fft_convolve_f32(const float* audio, const float* impulse_response, float* out, int B,
                                     int Bir, int F, int L, int N, int n_out, int start, void* stream) {
  if (!audio || !impulse_response || !out) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || L <= 0 || N <= 0 || n_out <= 0 || start < 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (Bir != B && Bir != 1) return DDSP_ERR_BAD_SHAPE;
  FirArgs p;
  p.N = N; p.F = F; p.L = L; p.start = start; p.n_out = n_out;
  p.frame_size = (N + F - 1) / F;                                              // core.py:1446
  if ((N + p.frame_size - 1) / p.frame_size != F) return DDSP_ERR_BAD_SHAPE;   // :1451-1457
  p.ir_batch_stride = (Bir == 1) ? 0 : (size_t)F * L;
  const dim3 grid(grid_for((size_t)n_out, 2048), (unsigned)B);
  hipLaunchKernelGGL(tv_fir_any_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, audio, impulse_response,
                     out, p);
  return check_launch();
}

extern "C" int ddsp_harmonic_envelopes_f32(const float* amplitudes, const float* harmonic_distribution,
                                           const float* f0_hz, const float* harmonic_shifts,
                                           float* harmonic_frequencies, float* harmonic_amplitudes, int B,
                                           int F, int K, void* stream) {
  if (!amplitudes || !f0_hz || !harmonic_frequencies || !harmonic_amplitudes) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0) return DDSP_ERR_BAD_SHAPE;
  const size_t rows = (size_t)B * F;
  hipLaunchKernelGGL(harmonic_envelopes_kernel, dim3(grid_for(rows * K)), dim3(kThreads), 0,
                     (hipStream_t)stream, amplitudes, harmonic_distribution, f0_hz, harmonic_shifts,
                     harmonic_frequencies, harmonic_amplitudes, rows, K);
  return check_launch();
}

extern "C" int ddsp_harmonic_frequencies_backward_f32(const float* grad_harmonic_frequencies, const float* harmonic_shifts,
                                                      float* grad_f0_hz, int B, int F, int K, void* stream) {
  if (!grad_harmonic_frequencies || !grad_f0_hz) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0) return DDSP_ERR_BAD_SHAPE;
  const size_t rows = (size_t)B * F;
  hipLaunchKernelGGL(harmonic_frequencies_backward_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, (hipStream_t)stream,
                     grad_harmonic_frequencies, harmonic_shifts, grad_f0_hz, rows, K);
  return check_launch();
}

extern "C" int ddsp_scale_f32(const float* x, const float* scale, float* out, size_t n, void* stream) {
  if (!x || !scale || !out) return DDSP_ERR_NULL_POINTER;
  if (n == 0) return DDSP_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, x, scale, out, n);
  return check_launch();
}

extern "C" size_t ddsp_harmonic_oscillator_bank_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const size_t chunks = ((size_t)N + kHobChunk - 1) / kHobChunk;
  return ((size_t)B * chunks + (size_t)B) * sizeof(double);
}

extern "C" int ddsp_harmonic_oscillator_bank_f32(const float* frequency, const float* amplitude_envelopes,
                                                 const float* initial_phase, float* audio, float* final_phase,
                                                 void* workspace, size_t workspace_bytes, int B, int N, int K,
                                                 int sample_rate, int use_angular_cumsum, void* stream) {
  if (!frequency || !amplitude_envelopes || !audio || !final_phase || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || K <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_harmonic_oscillator_bank_workspace_bytes(B, N) || ((uintptr_t)workspace & 7)) return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int C = (N + kHobChunk - 1) / kHobChunk;
  double* sums = (double*)workspace;
  double* total = sums + (size_t)B * C;
  const double inv_sr = 1.0 / (double)sample_rate;
  hipLaunchKernelGGL(hob_chunk_sums_kernel, dim3(grid_for((size_t)B * C)), dim3(kThreads), 0, st, frequency, sums, B, N, C);
  hipLaunchKernelGGL(hob_prefix_kernel, dim3(grid_for((size_t)B)), dim3(kThreads), 0, st, sums, total, B, C, inv_sr);
  hipLaunchKernelGGL(hob_synth_kernel, dim3(grid_for((size_t)B * N)), dim3(kThreads), 0, st, frequency, amplitude_envelopes,
                     initial_phase, sums, total, audio, final_phase, B, N, K, C, inv_sr, use_angular_cumsum ? 1 : 0);
  return check_launch();
}

extern "C" size_t ddsp_harmonic_f0_grad_workspace_bytes(int B, int F, int K, int N) {
  (void)K;
  if (B <= 0 || F <= 0 || N <= 0) return 0;
  return align_up((size_t)B * F * sizeof(double), 16) + align_up((size_t)B * F * 3 * sizeof(double), 16) +
         align_up((size_t)B * N * sizeof(float), 16);
}

extern "C" int ddsp_harmonic_f0_grad_f32(const float* ctl_amplitudes, const float* ctl_harmonic_distribution,
                                         const float* f0_hz, const float* grad_audio, float* grad_f0,
                                         void* workspace, size_t workspace_bytes, int B, int F, int K, int N,
                                         int sample_rate, unsigned flags, void* stream) {
  if (!ctl_amplitudes || !ctl_harmonic_distribution || !f0_hz || !grad_audio || !grad_f0 || !workspace)
    return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || K <= 0 || N <= 0 || sample_rate <= 0) return DDSP_ERR_BAD_SHAPE;
  if (N % F != 0) return DDSP_ERR_UNSUPPORTED;
  if (workspace_bytes < ddsp_harmonic_f0_grad_workspace_bytes(B, F, K, N) || ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  F0GradArgs p;
  p.B = B; p.F = F; p.K = K; p.N = N; p.hop = N / F;
  p.nyquist = (float)sample_rate / 2.0f;
  p.inv_sr = 1.0 / (double)sample_rate;
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  char* ws = (char*)workspace;
  double* theta0 = (double*)ws;
  ws += align_up((size_t)B * F * sizeof(double), 16);
  double* sums = (double*)ws;
  ws += align_up((size_t)B * F * 3 * sizeof(double), 16);
  float* c = (float*)ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(f0grad_phase_kernel, dim3(grid_for((size_t)B)), dim3(kThreads), 0, st, f0_hz, theta0, p);
  hipLaunchKernelGGL(f0grad_c_kernel, dim3(grid_for((size_t)B * N)), dim3(kThreads), 0, st, ctl_amplitudes,
                     ctl_harmonic_distribution, f0_hz, (const double*)theta0, grad_audio, c, p);
  hipLaunchKernelGGL(f0grad_frame_kernel, dim3(grid_for((size_t)B * F)), dim3(kThreads), 0, st,
                     (const float*)c, sums, p);
  hipLaunchKernelGGL(f0grad_scan_kernel, dim3(grid_for((size_t)B)), dim3(kThreads), 0, st,
                     (const double*)sums, grad_f0, p);
  return check_launch();
}

extern "C" int ddsp_exp_decay_ir_f32(const float* gain, const float* decay, const float* noise, float* ir, int B,
                                     int L, unsigned flags, void* stream) {
  if (!gain || !decay || !noise || !ir) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || L <= 0) return DDSP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(exp_decay_ir_kernel, dim3(grid_for((size_t)B * L)), dim3(kThreads), 0, (hipStream_t)stream,
                     gain, decay, noise, ir, B, L, (flags & DDSP_DECAY_SCALE_EXP_SIGMOID) ? 1 : 0);
  return check_launch();
}

extern "C" size_t ddsp_exp_decay_ir_backward_workspace_bytes(int B, int L) {
  (void)L;
  return B > 0 ? (size_t)B * kDecayPartials * 2 * sizeof(double) : 0;
}

extern "C" int ddsp_exp_decay_ir_backward_f32(const float* gain, const float* decay, const float* noise,
                                              const float* grad_ir, float* grad_gain, float* grad_decay,
                                              void* workspace, size_t workspace_bytes, int B, int L,
                                              unsigned flags, void* stream) {
  if (!gain || !decay || !noise || !grad_ir || !grad_gain || !grad_decay || !workspace)
    return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || L <= 0) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_exp_decay_ir_backward_workspace_bytes(B, L) || ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  double* partial = (double*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const int scale = (flags & DDSP_DECAY_SCALE_EXP_SIGMOID) ? 1 : 0;
  hipLaunchKernelGGL(exp_decay_bwd_partial_kernel, dim3(grid_for((size_t)B * kDecayPartials)), dim3(kThreads), 0,
                     st, decay, noise, grad_ir, partial, B, L);
  hipLaunchKernelGGL(exp_decay_bwd_finish_kernel, dim3(grid_for((size_t)B)), dim3(kThreads), 0, st, gain, decay,
                     (const double*)partial, grad_gain, grad_decay, B, scale);
  return check_launch();
}

// ---- small stand-alone pieces of ddsp/core.py that the synths only use fused: callable on their own for drop-in completeness ----
namespace ddsp {
namespace general {
// core.safe_divide (ddsp/core.py:207-210): numerator [rows, C], denominator [rows, C] or [rows, 1] (den_cols = 1)
__global__ __launch_bounds__(kThreads) void safe_divide_kernel(const float* __restrict__ num, const float* __restrict__ den,
                                                               float* __restrict__ out, size_t n, int C, int den_cols, float eps) {
  for (size_t i = global_thread(); i < n; i += grid_threads()) {
    const float d = den[den_cols == 1 ? i / (size_t)C : i];
    out[i] = num[i] / (d == 0.0f ? eps : d);
  }
}
// core.safe_log (ddsp/core.py:213-216)
__global__ __launch_bounds__(kThreads) void safe_log_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, float eps) {
  for (size_t i = global_thread(); i < n; i += grid_threads()) {
    const float v = x[i];
    out[i] = logf(v <= 0.0f ? eps : v);
  }
}
// core.get_harmonic_frequencies (ddsp/core.py:1028-1045): out[r, k] = fl32(f[r] (k + 1))
__global__ __launch_bounds__(kThreads) void harmonic_frequencies_kernel(const float* __restrict__ f, float* __restrict__ out,
                                                                        size_t rows, int K) {
  const size_t n = rows * (size_t)K;
  for (size_t i = global_thread(); i < n; i += grid_threads()) {
    const size_t r = i / (size_t)K;
    out[i] = rn_mul(f[r], (float)((int)(i - r * (size_t)K) + 1));
  }
}
// core.remove_above_nyquist (ddsp/core.py:869-891)
__global__ __launch_bounds__(kThreads) void remove_above_nyquist_kernel(const float* __restrict__ freq, const float* __restrict__ amp,
                                                                        float* __restrict__ out, size_t n, float nyquist) {
  for (size_t i = global_thread(); i < n; i += grid_threads()) out[i] = freq[i] >= nyquist ? 0.0f : amp[i];
}
// core.angular_cumsum (ddsp/core.py:800-866) on [B, T, C]: the phase in [0, 2 pi).  fp64 revolutions, chunks of 256 samples:
// sums per chunk, an exclusive wrapped prefix over the chunks, then the running phase inside each chunk - exact to ~1e-13
// revolutions for any length (the reference's fp32 chunks of `chunk_size` drift by ~1e-4 rad over a 4 s clip).
constexpr int kAngChunk = 256;
__global__ __launch_bounds__(kThreads) void ang_chunk_sums_kernel(const float* __restrict__ w, double* __restrict__ sums, int T, int C,
                                                                  int n_chunks) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int t0 = c * kAngChunk, t1 = min(t0 + kAngChunk, T);
  const float* __restrict__ wb = w + (size_t)b * T * C;
  for (int k = threadIdx.x; k < C; k += kThreads) {
    double s = 0.0;
    for (int t = t0; t < t1; ++t) s += (double)wb[(size_t)t * C + k];
    sums[((size_t)b * n_chunks + c) * C + k] = s;
  }
}
__global__ __launch_bounds__(kThreads) void ang_prefix_kernel(double* __restrict__ sums, int C, int n_chunks) {
  const int b = blockIdx.y, k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= C) return;
  const double inv_two_pi = 0.15915494309189535;
  double run = 0.0;                                   // revolutions, wrapped
  for (int c = 0; c < n_chunks; ++c) {
    double* p = sums + ((size_t)b * n_chunks + c) * C + k;
    const double s = *p * inv_two_pi;
    *p = run;
    run += s;
    run -= floor(run);
  }
}
__global__ __launch_bounds__(kThreads) void ang_apply_kernel(const float* __restrict__ w, const double* __restrict__ offs,
                                                             float* __restrict__ out, int T, int C, int n_chunks) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int t0 = c * kAngChunk, t1 = min(t0 + kAngChunk, T);
  const double inv_two_pi = 0.15915494309189535, two_pi = 6.283185307179586;
  const float* __restrict__ wb = w + (size_t)b * T * C;
  float* __restrict__ ob = out + (size_t)b * T * C;
  for (int k = threadIdx.x; k < C; k += kThreads) {
    double ph = offs[((size_t)b * n_chunks + c) * C + k];
    for (int t = t0; t < t1; ++t) {
      ph += (double)wb[(size_t)t * C + k] * inv_two_pi;
      ph -= floor(ph);
      ob[(size_t)t * C + k] = (float)(ph * two_pi);
    }
  }
}
}  // namespace general
}  // namespace ddsp

extern "C" int ddsp_safe_divide_f32(const float* numerator, const float* denominator, float* out, size_t rows, int C,
                                    int den_cols, float eps, void* stream) {
  if (!numerator || !denominator || !out) return DDSP_ERR_NULL_POINTER;
  if (C <= 0 || (den_cols != 1 && den_cols != C)) return DDSP_ERR_BAD_SHAPE;
  if (rows == 0) return DDSP_OK;
  hipLaunchKernelGGL(safe_divide_kernel, dim3(grid_for(rows * (size_t)C)), dim3(kThreads), 0, (hipStream_t)stream, numerator,
                     denominator, out, rows * (size_t)C, C, den_cols, eps);
  return check_launch();
}

extern "C" int ddsp_safe_log_f32(const float* x, float* out, size_t n, float eps, void* stream) {
  if (!x || !out) return DDSP_ERR_NULL_POINTER;
  if (n == 0) return DDSP_OK;
  hipLaunchKernelGGL(safe_log_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, x, out, n, eps);
  return check_launch();
}

extern "C" int ddsp_harmonic_frequencies_f32(const float* frequencies, float* out, size_t rows, int n_harmonics, void* stream) {
  if (!frequencies || !out) return DDSP_ERR_NULL_POINTER;
  if (n_harmonics <= 0) return DDSP_ERR_BAD_SHAPE;
  if (rows == 0) return DDSP_OK;
  hipLaunchKernelGGL(harmonic_frequencies_kernel, dim3(grid_for(rows * (size_t)n_harmonics)), dim3(kThreads), 0,
                     (hipStream_t)stream, frequencies, out, rows, n_harmonics);
  return check_launch();
}

extern "C" int ddsp_remove_above_nyquist_f32(const float* frequency_envelopes, const float* amplitude_envelopes, float* out,
                                             size_t n, float sample_rate, void* stream) {
  if (!frequency_envelopes || !amplitude_envelopes || !out) return DDSP_ERR_NULL_POINTER;
  if (!(sample_rate > 0.0f)) return DDSP_ERR_BAD_SHAPE;
  if (n == 0) return DDSP_OK;
  hipLaunchKernelGGL(remove_above_nyquist_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream,
                     frequency_envelopes, amplitude_envelopes, out, n, (float)((double)sample_rate / 2.0));
  return check_launch();
}

extern "C" size_t ddsp_angular_cumsum_workspace_bytes(int B, int T, int C) {
  if (B <= 0 || T <= 0 || C <= 0) return 0;
  return (size_t)B * ((T + kAngChunk - 1) / kAngChunk) * C * sizeof(double);
}

extern "C" int ddsp_angular_cumsum_f32(const float* angular_frequency, float* out, void* workspace, size_t workspace_bytes,
                                       int B, int T, int C, void* stream) {
  if (!angular_frequency || !out || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || T <= 0 || C <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_angular_cumsum_workspace_bytes(B, T, C) || ((uintptr_t)workspace & 7)) return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n_chunks = (T + kAngChunk - 1) / kAngChunk;
  double* sums = (double*)workspace;
  hipLaunchKernelGGL(ang_chunk_sums_kernel, dim3(n_chunks, B), dim3(kThreads), 0, st, angular_frequency, sums, T, C, n_chunks);
  hipLaunchKernelGGL(ang_prefix_kernel, dim3((C + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st, sums, C, n_chunks);
  hipLaunchKernelGGL(ang_apply_kernel, dim3(n_chunks, B), dim3(kThreads), 0, st, angular_frequency, (const double*)sums, out, T, C,
                     n_chunks);
  return check_launch();
}

extern "C" int ddsp_sigmoid_f32(const float* in, float* out, size_t n, void* stream) {
  if (!in || !out) return DDSP_ERR_NULL_POINTER;
  if (n == 0) return DDSP_OK;
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in, out, n);
  return check_launch();
}

extern "C" int ddsp_sigmoid_backward_f32(const float* in, const float* grad_out, float* grad_in, size_t n, void* stream) {
  if (!in || !grad_out || !grad_in) return DDSP_ERR_NULL_POINTER;
  if (n == 0) return DDSP_OK;
  hipLaunchKernelGGL(sigmoid_backward_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in, grad_out, grad_in, n);
  return check_launch();
}

extern "C" int ddsp_mix_backward_f32(const float* signal_one, const float* signal_two, const float* mix_level, const float* grad_out,
                                     float* grad_one, float* grad_two, float* grad_level, size_t rows, int C, void* stream) {
  if (!signal_one || !signal_two || !mix_level || !grad_out) return DDSP_ERR_NULL_POINTER;
  if (C <= 0) return DDSP_ERR_BAD_SHAPE;
  if (rows == 0 || (!grad_one && !grad_two && !grad_level)) return DDSP_OK;
  hipLaunchKernelGGL(mix_backward_kernel, dim3(grid_for(rows)), dim3(kThreads), 0, (hipStream_t)stream, signal_one, signal_two,
                     mix_level, grad_out, grad_one, grad_two, grad_level, rows, C);
  return check_launch();
}

extern "C" int ddsp_mix_f32(const float* signal_one, const float* signal_two, const float* mix_level, float* out,
                            size_t rows, int C, void* stream) {
  if (!signal_one || !signal_two || !mix_level || !out) return DDSP_ERR_NULL_POINTER;
  if (C <= 0) return DDSP_ERR_BAD_SHAPE;
  if (rows == 0) return DDSP_OK;
  hipLaunchKernelGGL(mix_kernel, dim3(grid_for(rows * (size_t)C)), dim3(kThreads), 0, (hipStream_t)stream,
                     signal_one, signal_two, mix_level, out, rows, C);
  return check_launch();
}
