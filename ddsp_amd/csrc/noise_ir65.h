// Compile-time tables of the 65-band noise filter design (ddsp/core.py:1534-1565, 1477-1531 at M = 65, full window):
// the cosine transform of the zero-phase IR and the Hann window, shared by filtered_noise.hip and
// filtered_noise_mfma.hip (each translation unit carries its own __constant__ copy: no relocatable device code).
#pragma once
#include <hip/hip_runtime.h>

namespace ddsp {

// ---- compile-time cosine tables -----------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cos_taylor(double x) {        // |x| <= pi/4
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int i = 1; i <= 12; ++i) { term *= -x2 / ((2 * i - 1) * (2 * i)); sum += term; }
  return sum;
}
constexpr double sin_taylor(double x) {        // |x| <= pi/4
  double x2 = x * x, term = x, sum = x;
  for (int i = 1; i <= 12; ++i) { term *= -x2 / ((2 * i) * (2 * i + 1)); sum += term; }
  return sum;
}
constexpr double cos_q128(int q) {             // cos(2 pi q / 128), exact octant reduction
  q = ((q % 128) + 128) % 128;
  if (q > 64) q = 128 - q;                     // cos(2pi - x) = cos x        -> q in [0,64]
  bool neg = false;
  if (q > 32) { q = 64 - q; neg = true; }      // cos(pi - x) = -cos x        -> q in [0,32]
  const double v = (q <= 16) ? cos_taylor(2.0 * kPi * q / 128.0)
                             : sin_taylor(2.0 * kPi * (32 - q) / 128.0);   // cos x = sin(pi/2 - x)
  return neg ? -v : v;
}
constexpr int kIrRowStride = 80;               // floats per table row (16-dword aligned halves)
struct Ir65Table {
  // row n (0..32): [0..32] = w_m * cos(2 pi (2i) n / 128) for even m = 2i,
  //                [40..71] = w_m * cos(2 pi (2i+1) n / 128) for odd m = 2i+1,
  // w_m = irfft weight: 1/128 for the DC and Nyquist bins, 2/128 otherwise.
  float c[33 * kIrRowStride];
  float win[64];                               // Hann(128)[64 + d] = 0.5 + 0.5 cos(2 pi d / 128)
};
constexpr Ir65Table make_ir65_table() {
  Ir65Table t{};
  for (int n = 0; n <= 32; ++n) {
    for (int i = 0; i <= 32; ++i) {
      const int m = 2 * i;
      const double w = (m == 0 || m == 64) ? 1.0 / 128.0 : 2.0 / 128.0;
      t.c[n * kIrRowStride + i] = (float)(w * cos_q128(m * n));
    }
    for (int i = 0; i < 32; ++i) {
      const int m = 2 * i + 1;
      t.c[n * kIrRowStride + 40 + i] = (float)((2.0 / 128.0) * cos_q128(m * n));
    }
  }
  for (int d = 0; d < 64; ++d) t.win[d] = (float)(0.5 + 0.5 * cos_q128(d));
  return t;
}
static __constant__ Ir65Table kIr65 = make_ir65_table();


}  // namespace ddsp
