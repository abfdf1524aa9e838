// The general form of losses.SpectralLoss (ddsp/losses.py:131-243) on materialised magnitude spectrograms: every term of
// the reference - magnitudes, their finite differences in time and in frequency (core.diff, ddsp/core.py:171-199), their
// cumulative sum over frequency (tf.math.cumsum), log magnitudes (core.safe_log, ddsp/core.py:213-216) - under every
// loss_type of losses.mean_difference (ddsp/losses.py:102-128: 'L1', 'L2', 'COSINE' = tf.losses.cosine_distance over
// the last axis) with the optional `weights` mask, and its gradient with respect to the value spectrogram.
//
// The shipped gin configs ask for 'L1' with mag_weight / logmag_weight only (gin/models/ae.gin:36-41): that case runs
// the fused kernels of spectral_loss.hip, where the spectra never leave LDS.  Everything else comes here: the
// spectrograms |STFT| of target and value are written to HBM once per FFT size (stft_mag_kernel), one wavefront per
// (batch, frame) row forms the row's part of every requested term (neighbour rows / bins are read again from L2; the
// cumulative sums are wavefront scans with a carry), per-row fp64 partials are added in a fixed order by a one-block
// kernel (deterministic), and the gradient kernel turns the same rows into dL/d|X| for stft_l1_bwd_kernel<S, COT>.
// A generality path: HBM-bound on 2 x [B, frames, bins] floats per size, not tuned further.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {

constexpr int kStTerms = 5;            // mag, delta_time, delta_freq, cumsum_freq, logmag
constexpr int kStRowsPerBlock = 4;     // one wavefront per row
constexpr int kStMaxBins = 4097;       // fft size <= 8192 (frames of 6144 samples: vst_48k.gin)

struct SpecTermArgs {
  int B, FR, NB;
  int loss_type;                        // DDSP_LOSS_L1 / _L2 / _COSINE
  float w[kStTerms];                    // mag, delta_time, delta_freq, cumsum_freq, logmag weights (<= 0: term off)
  int wb, wf, wk;                       // extents of the `weights` mask (1 = broadcast); 0, 0, 0: no mask
  float safe_eps;
};

__device__ __forceinline__ float st_weight(const float* __restrict__ wts, const SpecTermArgs& p, int b, int f, int k) {
  if (wts == nullptr) return 1.0f;
  // (indices clamped to the mask's own extents: a mask made for a difference term has one row / bin less than the
  // magnitudes, and rows / bins that term does not use are still visited here - their weight is discarded, the read
  // must stay inside the buffer)
  const size_t i = ((size_t)min(b, p.wb - 1) * p.wf + min(f, p.wf - 1)) * p.wk + min(k, p.wk - 1);
  return wts[i];
}
__device__ __forceinline__ float st_safe_log(float x, float eps) { return __logf(x <= 0.0f ? eps : x); }
__device__ __forceinline__ float st_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// inclusive scan over the 64 lanes, ascending (up) or descending (down) lane order
__device__ __forceinline__ float st_scan_up(float x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(x, o); if (lane >= o) x += y; }
  return x;
}
__device__ __forceinline__ float st_scan_down(float x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_down(x, o); if (lane + o < 64) x += y; }
  return x;
}

// One element of a term: the pair (a, b) = (target-side, value-side) and its weight.  Returns the element's loss
// contribution for 'L1' / 'L2'; for 'COSINE' the product a b (the row's 1 - sum is formed by the caller).
__device__ __forceinline__ float st_elem(int loss_type, float a, float b, float w) {
  if (loss_type == DDSP_LOSS_L1) return fabsf((a - b) * w);
  if (loss_type == DDSP_LOSS_L2) return (a - b) * (a - b) * w;
  return a * b;
}
// d(row's loss) / d(value-side element b): the "G" of the file header
__device__ __forceinline__ float st_g(int loss_type, float a, float b, float w, float w_row) {
  if (loss_type == DDSP_LOSS_L1) return -st_sign((a - b) * w) * w;
  if (loss_type == DDSP_LOSS_L2) return -2.0f * (a - b) * w;
  return -a * w_row;
}

// partial[row][2 * term + 0] = the row's sum for the term; [2 * term + 1] = 1 where the row's COSINE weight is non-zero
__global__ __launch_bounds__(64 * kStRowsPerBlock) void spec_terms_kernel(const float* __restrict__ tm /*[B,FR,NB]*/,
                                                                        const float* __restrict__ vm,
                                                                        const float* __restrict__ wts,
                                                                        double* __restrict__ partial, SpecTermArgs p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * kStRowsPerBlock + (threadIdx.x >> 6);
  if (row >= (long)p.B * p.FR) return;
  const int b = (int)(row / p.FR), f = (int)(row - (long)b * p.FR);
  const float* __restrict__ t0 = tm + row * p.NB;
  const float* __restrict__ v0 = vm + row * p.NB;
  const bool has_next = f + 1 < p.FR;
  const float* __restrict__ t1 = has_next ? t0 + p.NB : t0;
  const float* __restrict__ v1 = has_next ? v0 + p.NB : v0;
  const bool cosine = p.loss_type == DDSP_LOSS_COSINE;
  float acc[kStTerms] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float carry_t = 0.0f, carry_v = 0.0f;
  for (int k0 = 0; k0 < p.NB; k0 += 64) {
    const int k = k0 + lane;
    const bool in = k < p.NB;
    const float t = in ? t0[k] : 0.0f, v = in ? v0[k] : 0.0f;
    const float w = in ? st_weight(wts, p, b, f, k) : 0.0f;
    if (p.w[0] > 0.0f && in) acc[0] += st_elem(p.loss_type, t, v, w);
    if (p.w[1] > 0.0f && in && has_next) acc[1] += st_elem(p.loss_type, t1[k] - t, v1[k] - v, w);
    if (p.w[2] > 0.0f && k + 1 < p.NB) acc[2] += st_elem(p.loss_type, t0[k + 1] - t, v0[k + 1] - v, w);
    if (p.w[3] > 0.0f) {                                          // every lane takes part in the scans
      const float ct = st_scan_up(t, lane) + carry_t, cv = st_scan_up(v, lane) + carry_v;
      if (in) acc[3] += st_elem(p.loss_type, ct, cv, w);
      carry_t = __shfl(ct, 63);
      carry_v = __shfl(cv, 63);
    }
    if (p.w[4] > 0.0f && in) acc[4] += st_elem(p.loss_type, st_safe_log(t, p.safe_eps), st_safe_log(v, p.safe_eps), w);
  }
  const float w_row = cosine ? st_weight(wts, p, b, f, 0) : 1.0f;
  double* __restrict__ out = partial + row * (2 * kStTerms);
#pragma unroll
  for (int i = 0; i < kStTerms; ++i) {
    const double sum = (double)wave_sum(acc[i]);
    if (lane == 0) {
      const bool row_in_term = (i != 1) || has_next;              // delta_time has FR - 1 rows
      if (cosine) {
        out[2 * i] = row_in_term ? (1.0 - sum) * (double)w_row : 0.0;
        out[2 * i + 1] = (row_in_term && w_row != 0.0f) ? 1.0 : 0.0;
      } else {
        out[2 * i] = sum;
        out[2 * i + 1] = 0.0;
      }
    }
  }
}

// One block: adds the per-row partials of every term in a fixed order, normalises (mean over the term's elements; COSINE:
// over the rows with a non-zero weight, tf.losses.Reduction.SUM_BY_NONZERO_WEIGHTS), adds the weighted terms to the
// running fp64 loss of the call and rewrites the fp32 result; leaves each term's weight / count for the gradient kernel.
__global__ __launch_bounds__(1024) void spec_terms_finish_kernel(const double* __restrict__ partial, double* __restrict__ acc,
                                                               double* __restrict__ coef /*[kStTerms]*/,
                                                               float* __restrict__ loss, SpecTermArgs p, int first) {
  __shared__ double red[2 * kStTerms][16];
  const long rows = (long)p.B * p.FR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s[2 * kStTerms];
#pragma unroll
  for (int i = 0; i < 2 * kStTerms; ++i) s[i] = 0.0;
  for (long r = threadIdx.x; r < rows; r += 1024) {
#pragma unroll
    for (int i = 0; i < 2 * kStTerms; ++i) s[i] += partial[r * (2 * kStTerms) + i];
  }
#pragma unroll
  for (int i = 0; i < 2 * kStTerms; ++i) {
    const double v = wave_sum_dpp(s[i]);
    if (lane == 0) red[i][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double total = first ? 0.0 : acc[0];
    for (int i = 0; i < kStTerms; ++i) {
      double sum = 0.0, cnt = 0.0;
      for (int w = 0; w < 16; ++w) { sum += red[2 * i][w]; cnt += red[2 * i + 1][w]; }
      double inv;
      if (p.loss_type == DDSP_LOSS_COSINE) {
        inv = cnt > 0.0 ? 1.0 / cnt : 0.0;
      } else {
        const double fr = (i == 1) ? (double)(p.FR - 1) : (double)p.FR;
        const double nb = (i == 2) ? (double)(p.NB - 1) : (double)p.NB;
        const double n = (double)p.B * fr * nb;
        inv = n > 0.0 ? 1.0 / n : 0.0;
      }
      const double wt = p.w[i] > 0.0f ? (double)p.w[i] : 0.0;
      total += wt * sum * inv;
      coef[i] = wt * inv;
    }
    acc[0] = total;
    *loss = (float)total;
  }
}

// dL/d value_mag[b, f, k] for an upstream gradient of 1 (one wavefront per row; the cumulative-sum term needs the
// row's G values twice - once to form them from the ascending scans, once for the descending suffix sum - and keeps
// them in LDS in between).
__global__ __launch_bounds__(64 * kStRowsPerBlock) void spec_terms_grad_kernel(const float* __restrict__ tm,
                                                                             const float* __restrict__ vm,
                                                                             const float* __restrict__ wts,
                                                                             const double* __restrict__ coef,
                                                                             float* __restrict__ cot, SpecTermArgs p) {
  extern __shared__ __attribute__((aligned(16))) float s_g_all[];           // [kStRowsPerBlock][NB + 63]: sized by the launch
  float* const s_g_row = s_g_all + (size_t)(threadIdx.x >> 6) * (p.NB + 63);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long rows = (long)p.B * p.FR;
  const long row_raw = (long)blockIdx.x * kStRowsPerBlock + wv;
  const bool row_ok = row_raw < rows;
  const long row = row_ok ? row_raw : rows - 1;                    // surplus wavefronts repeat the last row, store nothing
  const int b = (int)(row / p.FR), f = (int)(row - (long)b * p.FR);
  const float* __restrict__ t0 = tm + row * p.NB;
  const float* __restrict__ v0 = vm + row * p.NB;
  const bool has_next = f + 1 < p.FR, has_prev = f >= 1;
  const float* __restrict__ t1 = has_next ? t0 + p.NB : t0;
  const float* __restrict__ v1 = has_next ? v0 + p.NB : v0;
  const float* __restrict__ tp = has_prev ? t0 - p.NB : t0;
  const float* __restrict__ vp = has_prev ? v0 - p.NB : v0;
  const int lt = p.loss_type;
  const float c_mag = p.w[0] > 0.0f ? (float)coef[0] : 0.0f, c_dt = p.w[1] > 0.0f ? (float)coef[1] : 0.0f;
  const float c_df = p.w[2] > 0.0f ? (float)coef[2] : 0.0f, c_cs = p.w[3] > 0.0f ? (float)coef[3] : 0.0f;
  const float c_log = p.w[4] > 0.0f ? (float)coef[4] : 0.0f;
  const float w_row = st_weight(wts, p, b, f, 0);                  // COSINE: one weight per row
  const float w_prow = has_prev ? st_weight(wts, p, b, f - 1, 0) : 0.0f;
  // ---- cumulative-sum term, pass 1: G_cs(k) from the ascending scans -> LDS ------------------------------------
  if (c_cs != 0.0f) {
    float carry_t = 0.0f, carry_v = 0.0f;
    for (int k0 = 0; k0 < p.NB; k0 += 64) {
      const int k = k0 + lane;
      const bool in = k < p.NB;
      const float t = in ? t0[k] : 0.0f, v = in ? v0[k] : 0.0f;
      const float ct = st_scan_up(t, lane) + carry_t, cv = st_scan_up(v, lane) + carry_v;
      carry_t = __shfl(ct, 63);
      carry_v = __shfl(cv, 63);
      s_g_row[k] = in ? st_g(lt, ct, cv, st_weight(wts, p, b, f, k), w_row) : 0.0f;
    }
  }
  __syncthreads();
  // ---- descending over the chunks: suffix sums of G_cs, and every other term gathered per element ---------------
  float carry_s = 0.0f;
  const int n_chunks = (p.NB + 63) / 64;
  for (int c = n_chunks - 1; c >= 0; --c) {
    const int k = c * 64 + lane;
    const bool in = k < p.NB;
    float g_total = 0.0f;
    if (c_cs != 0.0f) {
      const float sfx = st_scan_down(in ? s_g_row[k] : 0.0f, lane) + carry_s;       // sum over k' >= k
      carry_s = __shfl(sfx, 0);
      g_total = c_cs * sfx;
    }
    if (in) {
      const float t = t0[k], v = v0[k];
      const float w = st_weight(wts, p, b, f, k);
      if (c_mag != 0.0f) g_total += c_mag * st_g(lt, t, v, w, w_row);
      if (c_log != 0.0f && v > 0.0f)                                                 // safe_log passes a gradient where v > 0
        g_total += c_log * st_g(lt, st_safe_log(t, p.safe_eps), __logf(v), w, w_row) / v;
      if (c_dt != 0.0f) {
        // b_dt[f, k] = v[f+1, k] - v[f, k]:  d/dv[f, k] = -G(f, k) + G(f-1, k)
        if (has_next) g_total -= c_dt * st_g(lt, t1[k] - t, v1[k] - v, w, w_row);
        if (has_prev) g_total += c_dt * st_g(lt, t - tp[k], v - vp[k], st_weight(wts, p, b, f - 1, k), w_prow);
      }
      if (c_df != 0.0f) {
        if (k + 1 < p.NB) g_total -= c_df * st_g(lt, t0[k + 1] - t, v0[k + 1] - v, w, w_row);
        if (k >= 1) g_total += c_df * st_g(lt, t - t0[k - 1], v - v0[k - 1], st_weight(wts, p, b, f, k - 1), w_row);
      }
      if (row_ok) cot[row * p.NB + k] = g_total;
    }
  }
}

}  // namespace ddsp

using namespace ddsp;

extern "C" size_t ddsp_spectral_terms_workspace_bytes(int B, int frames) {
  if (B <= 0 || frames <= 0) return 0;
  return ((size_t)B * frames * 2 * kStTerms + kStTerms + 1) * sizeof(double);
}

extern "C" int ddsp_spectral_terms_f32(const float* target_mag, const float* value_mag, const float* weights,
                                       int weights_b, int weights_f, int weights_k, float* grad_value_mag,
                                       double* loss_accumulator, float* loss, void* workspace, size_t workspace_bytes,
                                       int B, int frames, int bins, int loss_type, float mag_weight,
                                       float delta_time_weight, float delta_freq_weight, float cumsum_freq_weight,
                                       float logmag_weight, int first, void* stream) {
  if (!target_mag || !value_mag || !loss_accumulator || !loss || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || frames <= 0 || bins <= 0 || bins > kStMaxBins) return DDSP_ERR_BAD_SHAPE;
  if (loss_type != DDSP_LOSS_L1 && loss_type != DDSP_LOSS_L2 && loss_type != DDSP_LOSS_COSINE) return DDSP_ERR_BAD_SHAPE;
  if (weights && (weights_b < 1 || weights_f < 1 || weights_k < 1 || (weights_b != 1 && weights_b != B))) return DDSP_ERR_BAD_SHAPE;
  // a mask is broadcast (1), full, or one short along the axis of a difference term (losses.py:102-128)
  if (weights && ((weights_f != 1 && weights_f != frames && weights_f != frames - 1) ||
                  (weights_k != 1 && weights_k != bins && weights_k != bins - 1))) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_spectral_terms_workspace_bytes(B, frames) || ((uintptr_t)workspace & 7)) return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  SpecTermArgs p;
  p.B = B; p.FR = frames; p.NB = bins; p.loss_type = loss_type;
  p.w[0] = mag_weight; p.w[1] = delta_time_weight; p.w[2] = delta_freq_weight; p.w[3] = cumsum_freq_weight; p.w[4] = logmag_weight;
  p.wb = weights ? weights_b : 0; p.wf = weights ? weights_f : 0; p.wk = weights ? weights_k : 0;
  p.safe_eps = 1e-5f;
  double* partial = (double*)workspace;
  double* coef = partial + (size_t)B * frames * 2 * kStTerms;
  const unsigned blocks = (unsigned)(((size_t)B * frames + kStRowsPerBlock - 1) / kStRowsPerBlock);
  hipLaunchKernelGGL(spec_terms_kernel, dim3(blocks), dim3(64 * kStRowsPerBlock), 0, st, target_mag, value_mag, weights, partial, p);
  hipLaunchKernelGGL(spec_terms_finish_kernel, dim3(1), dim3(1024), 0, st, (const double*)partial, loss_accumulator, coef, loss, p,
                     first ? 1 : 0);
  if (grad_value_mag) {
    const size_t lds = (size_t)kStRowsPerBlock * (bins + 63) * sizeof(float);
    if (lds > 48 * 1024)      // (per launch: the attribute belongs to the current device's copy of the kernel)
      (void)hipFuncSetAttribute((const void*)spec_terms_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024);
    hipLaunchKernelGGL(spec_terms_grad_kernel, dim3(blocks), dim3(64 * kStRowsPerBlock), lds, st, target_mag, value_mag, weights,
                       (const double*)coef, grad_value_mag, p);
  }
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}
