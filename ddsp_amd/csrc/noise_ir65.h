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


// ---- the same cosine factor as fp16 hi / lo MFMA A-fragments (noise_mfma65_kernel) -------------------------------
// x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) 2048), both rounded to nearest even at compile time.
constexpr double ir65_pow2(int e) { double v = 1.0; for (int i = 0; i < (e < 0 ? -e : e); ++i) v = e < 0 ? v * 0.5 : v * 2.0; return v; }
constexpr double ir65_round_half_even(double v) {          // v >= 0, < 2^52
  const long long k = (long long)v;
  const double frac = v - (double)k;
  if (frac > 0.5 || (frac == 0.5 && (k & 1))) return (double)(k + 1);
  return (double)k;
}
// value and bit pattern of the nearest fp16 number (normal or subnormal; |x| far below the fp16 maximum)
constexpr double ir65_f16_value(double x, unsigned short* bits) {
  const bool neg = x < 0.0;
  const double a = neg ? -x : x;
  int e = 0;                                               // a = m 2^e, 1 <= m < 2
  if (a != 0.0) { double m = a; while (m >= 2.0) { m *= 0.5; ++e; } while (m < 1.0) { m *= 2.0; --e; } }
  int qe = (a == 0.0 || e < -14) ? -24 : e - 10;           // exponent of the quantum (subnormals: 2^-24)
  double r = ir65_round_half_even(a / ir65_pow2(qe)) * ir65_pow2(qe);
  if (bits) {
    unsigned short b = 0;
    if (r != 0.0) {
      int re = 0; double m = r; while (m >= 2.0) { m *= 0.5; ++re; } while (m < 1.0) { m *= 2.0; --re; }
      if (re < -14) b = (unsigned short)(r / ir65_pow2(-24));                                   // subnormal: mantissa only
      else b = (unsigned short)(((re + 15) << 10) | ((int)((m - 1.0) * 1024.0 + 0.5) & 1023));
    }
    *bits = (unsigned short)(b | (neg ? 0x8000 : 0));
  }
  return neg ? -r : r;
}
struct Ir65Frags {
  // [tap tile mt][parity: even / odd bins][part: hi / lo][lane][dword d]: elements 2 d, 2 d + 1 of the lane's A-fragment
  // (tap n = 16 mt + (lane & 15), bin index k' = 8 (lane >> 4) + e)
  unsigned int v[2][2][2][64][4];
};
constexpr Ir65Frags make_ir65_frags() {
  Ir65Frags f{};
  const Ir65Table t = make_ir65_table();
  for (int mt = 0; mt < 2; ++mt)
    for (int par = 0; par < 2; ++par)
      for (int lane = 0; lane < 64; ++lane)
        for (int d = 0; d < 4; ++d) {
          unsigned int hi2 = 0, lo2 = 0;
          for (int h = 0; h < 2; ++h) {
            const int e = 2 * d + h;
            const double x = (double)t.c[(16 * mt + (lane & 15)) * kIrRowStride + 40 * par + 8 * (lane >> 4) + e];
            unsigned short hb = 0, lb = 0;
            const double hv = ir65_f16_value(x, &hb);
            // (x - hi) * 2048 is exact in fp32 as well as here: x has 24 bits, hi its leading 11
            ir65_f16_value((x - hv) * 2048.0, &lb);
            hi2 |= (unsigned int)hb << (16 * h);
            lo2 |= (unsigned int)lb << (16 * h);
          }
          f.v[mt][par][0][lane][d] = hi2;
          f.v[mt][par][1][lane][d] = lo2;
        }
  return f;
}

}  // namespace ddsp
