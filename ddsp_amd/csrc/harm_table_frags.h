// The constant factor of harm_table_kernel's tabulation, sin(k phi_n) on the half-step grid phi_n = 2 pi (n + 1/2) / 512,
// as fp16 hi / lo MFMA A-fragments made at compile time: x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) 2048),
// x the fp32 value of the sine.  The T-wavefronts used to compute their share at kernel start - 64 v_sin_f32 and splits
// per lane, ~1.5 us before the first row load was even issued; now it is 16 loads that fly with the first rows.
#pragma once
#include "noise_ir65.h"      // the compile-time sine / cosine and fp16 rounding helpers

namespace ddsp {

constexpr double wt_sin_q1024(int q) {             // sin(2 pi q / 1024), exact octant reduction
  q &= 1023;
  bool neg = false;
  if (q >= 512) { q -= 512; neg = true; }          // sin(x + pi) = -sin x          -> q in [0, 512)
  if (q > 256) q = 512 - q;                        // sin(pi - x) = sin x           -> q in [0, 256]
  const double v = (q <= 128) ? sin_taylor(2.0 * kPi * q / 1024.0)
                              : cos_taylor(2.0 * kPi * (256 - q) / 1024.0);      // sin x = cos(pi/2 - x)
  return neg ? -v : v;
}

struct WtSinSplit { unsigned short hi[1024], lo[1024]; };
constexpr WtSinSplit make_wt_sin_split() {
  WtSinSplit t{};
  for (int q = 0; q < 1024; ++q) {
    const double x = (double)(float)wt_sin_q1024(q);                 // the fp32 value
    unsigned short hb = 0, lb = 0;
    const double hv = ir65_f16_value(x, &hb);
    ir65_f16_value((x - hv) * 2048.0, &lb);                          // exact in fp32 as well as here
    t.hi[q] = hb;
    t.lo[q] = lb;
  }
  return t;
}

struct WtFrags {
  // [T-wavefront rw][part: hi / lo][parity][position tile tt][k-step ks][lane][dword d]: elements 2 d, 2 d + 1 of the
  // lane's A-fragment: n = 16 (2 rw + tt) + (lane & 15), k' = 32 ks + 8 (lane >> 4) + e, k = 2 k' + 1 + parity
  unsigned int v[4][2][2][2][2][64][4];
};
constexpr WtFrags make_wt_frags(const WtSinSplit& t) {
  WtFrags f{};
  for (int rw = 0; rw < 4; ++rw)
    for (int par = 0; par < 2; ++par)
      for (int tt = 0; tt < 2; ++tt)
        for (int ks = 0; ks < 2; ++ks)
          for (int lane = 0; lane < 64; ++lane)
            for (int d = 0; d < 4; ++d) {
              unsigned int hi2 = 0, lo2 = 0;
              for (int h = 0; h < 2; ++h) {
                const int e = 2 * d + h;
                const int n = 16 * (2 * rw + tt) + (lane & 15);
                const int k = 2 * (32 * ks + 8 * (lane >> 4) + e) + 1 + par;
                const int q = (k * (2 * n + 1)) & 1023;              // the angle k (2n+1) / 1024 revolutions, exact
                hi2 |= (unsigned int)t.hi[q] << (16 * h);
                lo2 |= (unsigned int)t.lo[q] << (16 * h);
              }
              f.v[rw][0][par][tt][ks][lane][d] = hi2;
              f.v[rw][1][par][tt][ks][lane][d] = lo2;
            }
  return f;
}

// 129 .. 200 harmonics (harm_table_kernel<10, 4, ..>): four k-steps per parity.  A plane row ends at k' = 104, so the
// fourth step reads k' = 72 .. 103 and its fragment is zero where the third already went (k' < 96): k <= 208.
constexpr int wt_wide_kstep_base(int ks) { return ks < 3 ? 32 * ks : 72; }
struct WtFragsWide {
  // [T-wavefront rw][parity][position tile tt][part: hi / lo][k-step ks][lane][dword d]: what a tabulator fetches for one
  // parity is one block of 16 KB - a scalar base, four lane offsets, the k-step in the load's 12-bit immediate offset
  unsigned int v[4][2][2][2][4][64][4];
};
constexpr WtFragsWide make_wt_frags_wide(const WtSinSplit& t) {
  WtFragsWide f{};
  for (int rw = 0; rw < 4; ++rw)
    for (int par = 0; par < 2; ++par)
      for (int tt = 0; tt < 2; ++tt)
        for (int ks = 0; ks < 4; ++ks)
          for (int lane = 0; lane < 64; ++lane)
            for (int d = 0; d < 4; ++d) {
              unsigned int hi2 = 0, lo2 = 0;
              for (int h = 0; h < 2; ++h) {
                const int kp = wt_wide_kstep_base(ks) + 8 * (lane >> 4) + 2 * d + h;
                if (ks == 3 && kp < 96) continue;
                const int n = 16 * (2 * rw + tt) + (lane & 15);
                const int q = ((2 * kp + 1 + par) * (2 * n + 1)) & 1023;
                hi2 |= (unsigned int)t.hi[q] << (16 * h);
                lo2 |= (unsigned int)t.lo[q] << (16 * h);
              }
              f.v[rw][par][tt][0][ks][lane][d] = hi2;
              f.v[rw][par][tt][1][ks][lane][d] = lo2;
            }
  return f;
}

}  // namespace ddsp
