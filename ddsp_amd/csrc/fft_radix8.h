// The radix-8 butterfly shared by the LDS-resident transforms (spectral_loss.hip, reverb.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ddsp {

__device__ __forceinline__ float2 fft_cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 fft_csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 fft_cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 fft_conj(float2 a) { return make_float2(a.x, -a.y); }

// v[m] <- sum_j v[j] exp(-2 pi i j m / 8), in place, natural order
__device__ __forceinline__ void fft_dft8(float2 (&v)[8]) {
  const float kR = 0.70710678118654752f;
  const float2 b0 = fft_cadd(v[0], v[4]), b1 = fft_cadd(v[1], v[5]), b2 = fft_cadd(v[2], v[6]), b3 = fft_cadd(v[3], v[7]);
  const float2 c0 = fft_csub(v[0], v[4]), c1 = fft_csub(v[1], v[5]), c2 = fft_csub(v[2], v[6]), c3 = fft_csub(v[3], v[7]);
  // even outputs: the 4-point transform of b
  const float2 t0 = fft_cadd(b0, b2), t1 = fft_csub(b0, b2), t2 = fft_cadd(b1, b3), bd = fft_csub(b1, b3);
  const float2 t3 = make_float2(bd.y, -bd.x);                            // (b1 - b3) (-i)
  v[0] = fft_cadd(t0, t2); v[4] = fft_csub(t0, t2); v[2] = fft_cadd(t1, t3); v[6] = fft_csub(t1, t3);
  // odd outputs: c_j w^j (w = exp(-2 pi i / 8)), then the 4-point transform
  const float2 d1 = make_float2((c1.x + c1.y) * kR, (c1.y - c1.x) * kR);          // c1 (1 - i) / sqrt 2
  const float2 d2 = make_float2(c2.y, -c2.x);                                     // c2 (-i)
  const float2 d3 = make_float2((c3.y - c3.x) * kR, -(c3.x + c3.y) * kR);         // c3 (-1 - i) / sqrt 2
  const float2 u0 = fft_cadd(c0, d2), u1 = fft_csub(c0, d2), u2 = fft_cadd(d1, d3), ud = fft_csub(d1, d3);
  const float2 u3 = make_float2(ud.y, -ud.x);
  v[1] = fft_cadd(u0, u2); v[5] = fft_csub(u0, u2); v[3] = fft_cadd(u1, u3); v[7] = fft_csub(u1, u3);
}

// the powers 1 .. 7 of w1 (w[0] is unused)
__device__ __forceinline__ void fft_powers8(float2 w1, float2 (&w)[8]) {
  w[1] = w1; w[2] = fft_cmul(w1, w1); w[3] = fft_cmul(w[2], w1); w[4] = fft_cmul(w[2], w[2]);
  w[5] = fft_cmul(w[4], w1); w[6] = fft_cmul(w[3], w[3]); w[7] = fft_cmul(w[4], w[3]);
}

}  // namespace ddsp
