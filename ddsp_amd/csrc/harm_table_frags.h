// The constant factor of harm_table_kernel's tabulation as fp16 hi / lo MFMA A-fragments:
//     A[n][k] = sin(2 pi k (n + 1/2) / T) / psi_hat(k / T)        (n < T / 4, the half-step grid; k = 2 k' + 1 + parity)
// x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) 2048), x the fp32 value of the product.
//
// Round 4: (1) the reciprocal of the interpolation window's transform, which rounds 2-3 multiplied into the amplitude rows, is
// part of the constant factor - the row makers' planes hold the plain normalised distribution, whatever the table size of the
// chunk they belong to will turn out to be; (2) there is a fragment set per TABLE SIZE T = 512 / 256 / 128 / 64: a segment of
// frames whose harmonics below Nyquist number at most K_max T / 512 is tabulated on T points only (harmonic_table.hip, "table
// size"), so that the table positions of neighbouring samples stay 1.3 .. 5 entries apart at any f0.  psi_hat depends on k / T
// alone: 1 / psi_hat_T(k) = 1 / psi_hat_512(k 512 / T), the tables of csrc/wavetable_coeffs.h.
//
// The sets are made ONCE PER DEVICE on the host (a few hundred thousand sines and splits: a millisecond) and copied into
// __device__ arrays the first time a Harmonic kernel of the window is launched there (harmonic_table.hip, wt_upload_fragments);
// rounds 2-3 had the T = 512 set evaluated by the compiler, which five sets would have turned into minutes of constant evaluation.
#pragma once
#include <cmath>
#include <cstring>

namespace ddsp {

// [T-wavefront rw][part: hi / lo][parity][position tile tt][k-step ks][lane][dword d]: elements 2 d, 2 d + 1 of the lane's
// A-fragment: n = 16 (2 rw + tt) + (lane & 15), k' = 32 ks + 8 (lane >> 4) + e, k = 2 k' + 1 + parity
struct alignas(16) WtFrags512 { unsigned int v[4][2][2][2][2][64][4]; };      // (fetched 16 bytes per lane)
// the smaller tables have T / 64 position tiles - one per tabulator, the tabulators beyond that idle - and one k-step (their
// segments have at most 64 live harmonics): [position tile][part][parity][lane][dword]
struct alignas(16) WtFragsSmall { unsigned int v[4][2][2][64][4]; };
struct alignas(16) WtFragSet {            // everything the instances of one window (K <= 128) need
  WtFrags512 t512;            // 64 KB
  WtFragsSmall t256;          // 16 KB, four tiles
  WtFragsSmall t128;          // two tiles used
  WtFragsSmall t64;           // one tile used
};

// 129 .. 200 harmonics (harm_table_kernel<10, 4, ..>): four k-steps per parity, T = 512 only.  A plane row ends at k' = 104, so
// the fourth step reads k' = 72 .. 103 and its fragment is zero where the third already went (k' < 96): k <= 208.
constexpr int wt_wide_kstep_base(int ks) { return ks < 3 ? 32 * ks : 72; }
struct alignas(16) WtFragsWide {
  // [T-wavefront rw][parity][position tile tt][part: hi / lo][k-step ks][lane][dword d]: what a tabulator fetches for one
  // parity is one block of 16 KB - a scalar base, four lane offsets, the k-step in the load's 12-bit immediate offset
  unsigned int v[4][2][2][2][4][64][4];
};

// ---- host side: the values ------------------------------------------------------------------------------------------------
inline unsigned short wt_f16_bits(float x) {
  const _Float16 h = (_Float16)x;                              // round to nearest even, as v_cvt_f16_f32
  unsigned short b;
  memcpy(&b, &h, 2);
  return b;
}
inline float wt_f16_value(unsigned short b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
// the two fp16 halves of A[n][k] on T points; invpsi512[i] = 1 / psi_hat(i / 512), n_invpsi entries
inline void wt_frag_element(int T, int n, int k, const float* invpsi512, int n_invpsi, unsigned short* hi, unsigned short* lo) {
  const int i = k * (512 / T);
  if (k < 1 || i >= n_invpsi) { *hi = 0; *lo = 0; return; }    // beyond what a table of T points may carry: never live there
  // the angle k (2 n + 1) / (2 T) revolutions, reduced exactly
  const long long q = ((long long)k * (2 * n + 1)) % (2 * T);
  const double s = sin(2.0 * 3.14159265358979323846264338327950288 * (double)q / (double)(2 * T));
  const float x = (float)((double)(float)s * (double)invpsi512[i]);
  *hi = wt_f16_bits(x);
  *lo = wt_f16_bits((x - wt_f16_value(*hi)) * 2048.0f);        // (exact in fp32: x has 24 bits, hi its leading 11)
}
inline void wt_fill_frag_set(WtFragSet* f, const float* invpsi512, int n_invpsi) {
  memset(f, 0, sizeof(*f));
  for (int rw = 0; rw < 4; ++rw)
    for (int par = 0; par < 2; ++par)
      for (int tt = 0; tt < 2; ++tt)
        for (int ks = 0; ks < 2; ++ks)
          for (int lane = 0; lane < 64; ++lane)
            for (int d = 0; d < 4; ++d) {
              unsigned int hi2 = 0, lo2 = 0;
              for (int h = 0; h < 2; ++h) {
                unsigned short hb, lb;
                wt_frag_element(512, 16 * (2 * rw + tt) + (lane & 15), 2 * (32 * ks + 8 * (lane >> 4) + 2 * d + h) + 1 + par,
                                invpsi512, n_invpsi, &hb, &lb);
                hi2 |= (unsigned int)hb << (16 * h);
                lo2 |= (unsigned int)lb << (16 * h);
              }
              f->t512.v[rw][0][par][tt][ks][lane][d] = hi2;
              f->t512.v[rw][1][par][tt][ks][lane][d] = lo2;
            }
  WtFragsSmall* small[3] = {&f->t256, &f->t128, &f->t64};
  for (int s = 0; s < 3; ++s) {
    const int T = 256 >> s;
    for (int tile = 0; tile < T / 64; ++tile)
      for (int par = 0; par < 2; ++par)
        for (int lane = 0; lane < 64; ++lane)
          for (int d = 0; d < 4; ++d) {
            unsigned int hi2 = 0, lo2 = 0;
            for (int h = 0; h < 2; ++h) {
              unsigned short hb, lb;
              wt_frag_element(T, 16 * tile + (lane & 15), 2 * (8 * (lane >> 4) + 2 * d + h) + 1 + par, invpsi512, n_invpsi, &hb, &lb);
              hi2 |= (unsigned int)hb << (16 * h);
              lo2 |= (unsigned int)lb << (16 * h);
            }
            small[s]->v[tile][0][par][lane][d] = hi2;
            small[s]->v[tile][1][par][lane][d] = lo2;
          }
  }
}
inline void wt_fill_frags_wide(WtFragsWide* f, const float* invpsi512, int n_invpsi) {
  memset(f, 0, sizeof(*f));
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
                unsigned short hb, lb;
                wt_frag_element(512, 16 * (2 * rw + tt) + (lane & 15), 2 * kp + 1 + par, invpsi512, n_invpsi, &hb, &lb);
                hi2 |= (unsigned int)hb << (16 * h);
                lo2 |= (unsigned int)lb << (16 * h);
              }
              f->v[rw][par][tt][0][ks][lane][d] = hi2;
              f->v[rw][par][tt][1][ks][lane][d] = lo2;
            }
}

}  // namespace ddsp
