// The frame-rate chain rule of Harmonic's backward pass for ONE (row, frame), on one wavefront (lanes = harmonics): from
// dL/da[j,k] = P[j,k] + Q[j-1,k] (+ Q[F-1,k] for the held last frame; `fetch(k)`) through amp * hd_norm, safe_divide, the
// frame-rate Nyquist mask and exp_sigmoid (ddsp/synths.py:94-121, core.py:894-907, 386-404) to dL/d amplitudes and
// dL/d harmonic_distribution.  Shared by harm_bwd_chain_kernel (harmonic.hip: P / Q from the workspace) and
// harm_bwd_table_kernel (harmonic_bwd_table.hip: P / Q still in LDS).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {

struct ChainPq { float p, q, x; };      // what harm_bwd_chain_kernel loads per harmonic: P[j], Q[j-1], Q[j] (the held last frame)

struct BwdArgs {
  int F, K, N, hop;
  float sample_rate, nyquist;
  int amp_linear;
  unsigned flags;
  int inputs_are_controls;
};

// `fetch(k)` returns what it LOADS for harmonic k (no arithmetic), `combine(loaded)` makes dL/da[j,k] of it: the loads of a
// row are all issued before the first of them is waited for.
template <int NCHUNK, class Fetch, class Combine>   // ceil(K/64) <= NCHUNK
__device__ __forceinline__ void harm_chain_row(int lane, long row, int j, const float* __restrict__ amplitudes,
                                               const float* __restrict__ hd, const float* __restrict__ f0_hz,
                                               float* __restrict__ grad_amp, float* __restrict__ grad_hd, const BwdArgs& p,
                                               Fetch fetch, Combine combine) {
  (void)j;
  const int K = p.K;
  const bool is_ctl = p.inputs_are_controls != 0;
  const bool scale = (p.flags & DDSP_HARM_SCALE_EXP_SIGMOID) && !is_ctl;
  const bool normalize = (p.flags & DDSP_HARM_NORMALIZE_NYQUIST) && !is_ctl;
  const float kLog10 = 2.302585092994046f;
  const float f0r = f0_hz[row];
  const float amp_raw = amplitudes[row];
  const float amp_s = scale ? exp_sigmoid(amp_raw, kLog10, 2.0f, 1e-7f) : amp_raw;
  float x[NCHUNK], raw[NCHUNK], ga[NCHUNK];
  bool live[NCHUNK];
  float part = 0.0f;
  // every load of the row first, the arithmetic behind them: with a load, its wait and an exp_sigmoid per chunk in turn a
  // wavefront made six memory round trips one after the other (round 5: harm_bwd_chain_kernel 19 -> 12 us at batch 32)
  decltype(fetch(0)) loaded[NCHUNK] = {};
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int k = c * 64 + lane;
    raw[c] = 0.0f;
    if (k < K) {
      raw[c] = hd[(size_t)row * K + k];
      loaded[c] = fetch(k);
    }
  }
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int k = c * 64 + lane;
    x[c] = 0.0f; live[c] = false; ga[c] = 0.0f;
    if (k < K) {
      ga[c] = combine(loaded[c]);
      float v = scale ? exp_sigmoid(raw[c], kLog10, 2.0f, 1e-7f) : raw[c];
      live[c] = !(normalize && (f0r * (float)(k + 1) >= p.nyquist));
      if (!live[c]) v = 0.0f;
      x[c] = v;
    }
    part += x[c];
  }
  float inv = 1.0f;
  bool den_zero = false;
  if (!is_ctl) {
    float den = wave_sum_dpp(part);      // (DPP row reductions: the shuffle form is six dependent ds_bpermute round trips per sum)
    den_zero = den == 0.0f;
    if (den_zero) den = 1e-7f;
    inv = 1.0f / den;
  }
  float dot = 0.0f;                                      // sum_k ga * hd_norm = dL/d(amp_scaled)
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) dot = fmaf(ga[c], x[c] * inv, dot);
  dot = wave_sum_dpp(dot);
  if (lane == 0)
    grad_amp[row] = scale ? dot * kLog10 * (amp_s - 1e-7f) * (1.0f - 1.0f / (1.0f + __expf(-amp_raw))) : dot;
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int k = c * 64 + lane;
    if (k < K) {
      float d;
      if (is_ctl) {
        d = ga[c] * amp_s;
      } else {
        // hd_norm = x / den: d x = (d hd_norm - sum_k d hd_norm hd_norm) / den, d hd_norm = ga * amp
        d = (live[c] && !den_zero) ? amp_s * (ga[c] - dot) * inv : 0.0f;
        if (scale) d *= kLog10 * (x[c] - 1e-7f) * (1.0f - 1.0f / (1.0f + __expf(-raw[c])));
      }
      grad_hd[(size_t)row * K + k] = d;
    }
  }
}

}  // namespace ddsp
