// Harmonic.__call__ (ddsp/synths.py:94-146) for gfx950 as wavetables built on the matrix cores.
//
// Within frame j every output sample is
//     audio[t] = w_cur(r) S_j(theta_t) + w_next(r) S_{j+1}(theta_t),   S_j(theta) = sum_k a_j[k] sin(2 pi k theta)
// (harmonic.hip derives this form from core.upsample_with_windows / core.oscillator_bank): a frame's
// amplitude row enters only through the periodic, band-limited function S_j.  harm_fused_kernel evaluates
// S_j and S_{j+1} harmonic by harmonic at every sample (3 FMAs per harmonic and sample, VALU bound).
// Here each S_j is tabulated once on T = 512 uniform phases and every sample reads the two tables
// through a W-tap Kaiser-Bessel window (the interpolation step of a type-2 nonuniform FFT, Dutt & Rokhlin
// 1993; coefficients and the error analysis: tools/gen_wavetable_coeffs.py).  The tabulation
//     S_j(phi_n) = sum_k sin(k phi_n) a_j[k] / psi_hat(k)
// is a dense [T x K] . [K x frames] product with a constant left factor - the one place on this path that
// is matrix-core work.  The symmetries of the sine cut it to an eighth: with the table grid offset by
// half a step, phi_n = 2 pi (n + 1/2) / T, and O / E the sums over odd / even harmonics,
//     S(n) = O(n) + E(n),  S(T/2-1-n) = O(n) - E(n),  S(T-1-n) = -S(n)      (n = 0 .. T/4-1)
// so two [128 x K/2] products per row give the whole table.  v_mfma_f32_16x16x4_f32 (exact fp32, 64
// flop/clk/SIMD) does them with the constant factor resident in registers for the life of the block.
//
// Per sample the VALU work drops from ~4 K flop-instructions to ~65 (phase, W polynomial weights,
// 2 W taps), independent of K; error vs exact arithmetic <= 6.3e-6 * sum_k a_k (W = 6, K <= 100),
// 6.5e-6 (W = 8, K <= 128), smaller than the sine recurrence's 3.1e-5.
//
// The audio-rate Nyquist mask of core.oscillator_bank (core.py:942-944) only differs from the frame-rate
// mask of normalize_harmonics inside frames where a harmonic crosses sr/2; for those harmonics the masked
// samples subtract their contribution again, evaluated directly.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "../../include/ddsp_amd.h"
#include "common.h"
#include "profile.h"
#include "harmonic_table.h"
#define DDSP_WT_TABLE __constant__
#include "wavetable_coeffs.h"
#include "harm_table_frags.h"

namespace ddsp {

constexpr WtSinSplit kWtSinSplit = make_wt_sin_split();
static __device__ const WtFrags kWtFrags = make_wt_frags(kWtSinSplit);      // 64 KB of constants, fetched once per T-wavefront

constexpr int kWtT = 512;            // table points per revolution
constexpr int kWtHalf = kWtT / 2;    // the table holds p in [-kWtH, kWtHalf + kWtH); the other half is its mirror image
constexpr int kWtNQ = kWtT / 4;      // positions produced by the matrix product
constexpr int kWtH = 4;              // halo entries on either side (>= W/2)
constexpr int kWtTS = 268;           // table row stride in floats: 4*odd, so 16 rows' b128 writes spread over the banks
constexpr int kWtRowTiles = 2;       // MFMA N-tiles of 16 amplitude rows per chunk
constexpr int kWtRows = 16 * kWtRowTiles;    // amplitude rows per chunk
constexpr int kWtFrames = kWtRows - 1;       // frames per chunk (31): row r+1 is the "next" row of frame r
constexpr int kWtNT = 4;             // tiles of 64 samples an S-wavefront carries through phase B together
constexpr int kWtPS = 72;            // row stride of an amplitude plane (fp16 elements; odd / even harmonics apart): 144 B
constexpr float kWtLoScale = 2048.0f; // x = hi + lo / 2048 in two fp16 numbers
constexpr int kWtRS = 132;           // row stride of the raw staging buffer: 128 harmonics, f0, amplitude

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));      // what v_cvt_pkrtz_f16_f32 returns

struct TableArgs {
  int B, F, K, N, hop, chunks_per_row, n_chunks;
  float nyquist, nyq_lo, nyq_hi;
  int amp_linear, f0_vec;
  double inv_sr, inv_2hop, hop_d, half_hm1;
  long long* dbg;      // timeline of block 0 (tools/exp_table.py --timeline), or null
};

struct ChunkTables {
  double theta[kWtRows], w[kWtRows], dw[kWtRows];
  float f0[kWtRows + 2];
  int kA[kWtRows], kN[kWtRows];
  int cross;           // any frame of the chunk with a harmonic crossing Nyquist inside it (kA < kN)
};

template <int W> struct WtPoly;
template <> struct WtPoly<6> {
  static constexpr int DE = kWtDegE6, DO = kWtDegO6;
  static constexpr float e(int p, int d) { return kWtE6[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO6[p * (DO + 1) + d]; }
  __device__ static float invpsi(int k) { return kWtInvPsi6_T512[k]; }
  __device__ static float psi(int k) { return kWtPsi6_T512[k]; }
};
template <> struct WtPoly<8> {
  static constexpr int DE = kWtDegE8, DO = kWtDegO8;
  static constexpr float e(int p, int d) { return kWtE8[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO8[p * (DO + 1) + d]; }
  __device__ static float invpsi(int k) { return kWtInvPsi8_T512[k]; }
  __device__ static float psi(int k) { return kWtPsi8_T512[k]; }
};

template <int W, int P, int D> struct WtE { static constexpr float v = WtPoly<W>::e(P, D); };
template <int W, int P, int D> struct WtO { static constexpr float v = WtPoly<W>::o(P, D); };

// window weights of the tap pair at distance -+(P + 1/2) from the centre: E(z^2) +- z O(z^2)
template <int W, int P>
__device__ __forceinline__ void wt_pair(float z, float z2, float& w_lo, float& w_hi) {
  float e, o;
  if constexpr (WtPoly<W>::DE == 3)
    e = fmaf(fmaf(fmaf(WtE<W, P, 3>::v, z2, WtE<W, P, 2>::v), z2, WtE<W, P, 1>::v), z2, WtE<W, P, 0>::v);
  else
    e = fmaf(fmaf(WtE<W, P, 2>::v, z2, WtE<W, P, 1>::v), z2, WtE<W, P, 0>::v);
  if constexpr (WtPoly<W>::DO == 3)
    o = fmaf(fmaf(fmaf(WtO<W, P, 3>::v, z2, WtO<W, P, 2>::v), z2, WtO<W, P, 1>::v), z2, WtO<W, P, 0>::v);
  else
    o = fmaf(fmaf(WtO<W, P, 2>::v, z2, WtO<W, P, 1>::v), z2, WtO<W, P, 0>::v);
  w_lo = fmaf(z, o, e);
  w_hi = fmaf(-z, o, e);
}

// NT tiles at once (NT independent chains per instruction slot): every stage of a tile is a chain of dependent
// instructions - fp64 phase, LDS reads, the window polynomials - and a wavefront with one or two chains leaves most of
// its issue slots empty
template <int W, int P, int NT>
__device__ __forceinline__ void wt_taps(const float* const (&t)[kWtNT], const float (&z)[kWtNT], const float (&z2)[kWtNT],
                                        float (&acc0)[kWtNT], float (&acc1)[kWtNT]) {
  float lo[NT], hi[NT], a0[NT], a1[NT], a2[NT], a3[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    wt_pair<W, P>(z[u], z2[u], lo[u], hi[u]);
    a0[u] = t[u][-P]; a1[u] = t[u][1 + P]; a2[u] = t[u][kWtTS - P]; a3[u] = t[u][kWtTS + 1 + P];
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) { acc0[u] = fmaf(lo[u], a0[u], acc0[u]); acc1[u] = fmaf(lo[u], a2[u], acc1[u]); }
#pragma unroll
  for (int u = 0; u < NT; ++u) { acc0[u] = fmaf(hi[u], a1[u], acc0[u]); acc1[u] = fmaf(hi[u], a3[u], acc1[u]); }
  if constexpr (P + 1 < W / 2) wt_taps<W, P + 1, NT>(t, z, z2, acc0, acc1);
}

// The per-frame phase tables of a chunk (one wavefront, lanes = frames).  (Handing this block of fp64 work to a
// T-wavefront, between the issue of its MFMAs and the use of their results, was measured in round 2 and lost:
// 21.3 against 20.0 us at batch 32, profiles/r02a_harm_table_vs_direct.json - the variant is gone.)
// frame j carries f[t] = f_j + (f_{j+1}-f_j) r/hop; its sum over the frame is hop f_j + (f_{j+1}-f_j)(hop-1)/2,
// which telescopes over j < J to hop sum_{j<J} f_j + (f_J - f_0)(hop-1)/2
__device__ __forceinline__ void wt_phase_tables(const float* __restrict__ raw, ChunkTables& t, int lane, int nfr, int K,
                                                const TableArgs& p) {
  const double* psum = reinterpret_cast<const double*>(raw + kWtRows * kWtRS);
  const double before = (psum[0] + psum[1]) + (psum[2] + psum[3]);                 // sum_{j < j0} f_j
  const float f0_first = raw[kWtRows * kWtRS + 8];
  const float fj = raw[min(lane, nfr) * kWtRS + 128], fj1 = raw[min(lane + 1, nfr) * kWtRS + 128];
  const double fa = (double)fj, fb = (double)fj1;
  const double mine = (lane < nfr) ? fa : 0.0;
  double incl = mine;                                 // inclusive scan over the chunk's frames (lanes 0..31)
  incl += dpp_mov0<0x111, 0xF>(incl);   // row_shr:1
  incl += dpp_mov0<0x112, 0xF>(incl);   // row_shr:2
  incl += dpp_mov0<0x114, 0xF>(incl);   // row_shr:4
  incl += dpp_mov0<0x118, 0xF>(incl);   // row_shr:8
  {                                                   // lanes 16..31: + the total of lanes 0..15
    const long long bits = __builtin_bit_cast(long long, incl);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits & 0xffffffffll), 15);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), 15);
    const double first16 = __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
    if (lane >= 16) incl += first16;
  }
  const double s_excl = before + (incl - mine);
  const double run = p.hop_d * s_excl + (fa - (double)f0_first) * p.half_hm1;
  const double cyc = run * p.inv_sr;
  const float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
  int kA = K, kN = K;
  if (fmx > 0.0f) kA = (int)fminf((float)K, floorf(p.nyq_lo * __builtin_amdgcn_rcpf(fmx)));
  if (fmn > 0.0f) kN = (int)fminf((float)K, floorf(p.nyq_hi * __builtin_amdgcn_rcpf(fmn)));
  kA = max(min(kA, kN), 0);
  // harmonics [0,kA) are below Nyquist at every sample of the frame, [kN,K) at none: both rows carry zeros
  // there; [kA,kN) is decided per sample.  v_rcp_f32 (1 ulp) is well inside the 4e-6 guard band.  Phase B
  // looks at the per-frame bounds only when some frame of the chunk has a crossing at all.
  const bool crossing = lane < nfr && kA < kN;
  const unsigned long long any = __builtin_amdgcn_ballot_w64(crossing);
  if (lane == 0) t.cross = any != 0ull ? 1 : 0;
  if (lane <= kWtRows) t.f0[lane] = fj;
  if (lane < kWtRows) {
    t.theta[lane] = cyc - floor(cyc);
    t.w[lane] = fa * p.inv_sr;
    t.dw[lane] = (fb - fa) * p.inv_sr * p.inv_2hop;
    t.kA[lane] = kA;
    t.kN[lane] = kN;
  }
}

// One block = 12 wavefronts in two roles, one block per CU.  T-wavefronts (0..3, one per SIMD) own the
// matrix cores and the loads: each holds its share of the constant sine matrix in registers (two position
// tiles x two parities) and turns the amplitude planes of a chunk into its table; around that it fetches
// the raw rows of a later chunk from HBM into an LDS staging buffer (issued before the MFMAs, stored after
// them, so the latency hides behind the matrix pipe).  S-wavefronts (4..11) own the vector ALUs and never
// wait on HBM: the controls prologue of a chunk (phase A: exp_sigmoid, Nyquist mask, normalisation ->
// amplitude planes; the fp64 phase tables) and the per-sample interpolation (phase B).  Chunks move through
// a four-stage pipeline, one stage per tick and one barrier per tick:
//     tick tau:   T: rows of chunk tau+3 -> staging;  MFMAs and table of chunk tau+1
//                 S: phase B of chunk tau, then phase A of chunk tau+2
// tables and staging double-buffered, planes / frame tables triple-buffered in LDS.
//
// Precision of the product.  The exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the vector FMA rate and,
// measured (tools/exp_table_timeline.py), *on* the vector ALUs' time: with it the S-wavefronts sharing the
// SIMD crawled and the kernel was no faster than harm_fused_kernel.  The fp16 matrix cores are separate
// hardware and 16x faster, so both factors are split into two fp16 numbers, x = hi + lo / 2048 with
// hi = fp16(x), lo = fp16((x - hi) 2048) (the scaling keeps lo a normal number), and three products are
// accumulated in fp32: hi.hi, and hi.lo + lo.hi in a second accumulator that is scaled back once.  The
// dropped lo.lo term and the roundings of the lo parts leave an error <= 5e-8 sum_k |a_k| (tests/wavetable_model.py
// reproduces the split), below the fp32 round-off of the sum itself.
// NK: k-steps of 32 per parity (ceil(K/2) <= 32 NK); ONE_TILE: hop == 64
template <int W, int NK, bool ONE_TILE>
__global__ __launch_bounds__(768, 3) void harm_table_kernel(
    const float* __restrict__ amplitudes, const float* __restrict__ hd, const float* __restrict__ f0_all,
    float* __restrict__ audio, float* __restrict__ ctl_amp, float* __restrict__ ctl_hd, TableArgs p) {
  __shared__ __attribute__((aligned(16))) float tab_all[2][kWtRows * kWtTS];
  __shared__ __attribute__((aligned(16))) _Float16 planes_all[3][4 * kWtRows * kWtPS];   // [hi, lo][parity][row][k']: a_k / psi_hat(k)
  __shared__ __attribute__((aligned(16))) float raw_all[2][kWtRows * kWtRS + 12];     // raw rows; then 4 doubles (parts of the sum of f0 before the chunk) and f0 of frame 0
  __shared__ ChunkTables t_all[3];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_t = wave < 4;
  const int rw = is_t ? wave : wave - 4;         // index within the role: 0..3 (T), 0..7 (S)
  const int F = p.F, K = p.K;
  const int K4 = K >> 2;
  const float kLog10 = 2.302585092994046f;       // tf.math.log(exponent), ddsp/core.py:403
  const int n_my = ((int)p.n_chunks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // chunks of this block
  // this block's chunks are blockIdx.x + i gridDim.x; (batch row, chunk within the row) advance by a fixed step
  const int cpr = p.chunks_per_row;
  const int step_b = (int)gridDim.x / cpr, step_c = (int)gridDim.x % cpr;
  const int first_b = (int)blockIdx.x / cpr, first_c = (int)blockIdx.x % cpr;
  // 32 lanes per matrix row, lane kq owns harmonics 4 kq + 1 .. 4 kq + 4 (loads and phase A)
  const int sub = lane >> 5, kq = lane & 31;
  const bool live = kq < K4;
  const float4* __restrict__ hd4 = reinterpret_cast<const float4*>(hd);
  const int mi = lane & 15, mg = lane >> 4;      // MFMA fragment coordinates
  // debug timeline: [wavefront 0 / 4 / 11][tick + 3][stamp], shader clocks
  const int dbg_w = (wave == 0) ? 0 : (wave == 4) ? 1 : (wave == 11) ? 2 : -1;
  const bool dbg_on = p.dbg != nullptr && blockIdx.x == 0 && lane == 0 && dbg_w >= 0;
#define DDSP_WT_STAMP(i) do { if (dbg_on && tick + 3 < 64) p.dbg[(dbg_w * 64 + tick + 3) * 8 + (i)] = clock64(); } while (0)
#define DDSP_WT_ADVANCE(b_, c_) do { b_ += step_b; c_ += step_c; if (c_ >= cpr) { c_ -= cpr; b_ += 1; } } while (0)

  if (is_t) {
    // ---- this wavefront's share of the constant factor, in MFMA A-operand layout -----------------------
    // element e of lane (i = lane & 15, g = lane >> 4) of k-step ks, parity par, position tile pt: sin(k phi_n),
    // n = 16 pt + i, k' = 32 ks + 8 g + e, k = 2 k' + 1 + par (the B fragments use the same k'(g, e));
    // Rows k > K meet zero amplitudes.  Made at compile time (harm_table_frags.h): 16-byte loads, in flight together
    // with the first chunk's rows.
    f16x8 ahi[2][2][NK], alo[2][2][NK];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 vh = *reinterpret_cast<const u32x4*>(kWtFrags.v[rw][0][par][tt][ks][lane]);
          const u32x4 vl = *reinterpret_cast<const u32x4*>(kWtFrags.v[rw][1][par][tt][ks][lane]);
          ahi[par][tt][ks] = __builtin_bit_cast(f16x8, vh);
          alo[par][tt][ks] = __builtin_bit_cast(f16x8, vl);
        }
    int lb = first_b, lc = first_c;               // position of the chunk whose rows are fetched next
    int qb = first_b, qc = first_c;               // position of the chunk whose phase tables are made next

    for (int tick = -3; tick < n_my; ++tick) {
      DDSP_WT_STAMP(0);
      // ---------------- rows of chunk tick+3: issue the loads -----------------------------------------------
      // Nothing below may depend on the loaded values until the MFMAs have been issued, so every load is
      // unconditional (indices clamped to something valid, the result masked after the MFMAs): past the
      // block's last chunk the last one is simply fetched again into a staging slot nobody reads.
      const int lj0 = lc * kWtFrames;
      const int kqc = min(kq, K4 - 1);
      // this wavefront's eight rows of the chunk, two (sub = 0, 1) per load instruction
      ddsp_f32x4 lx[4];
      float lf0[4], lamp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lrow = lb * F + min(lj0 + rw * 8 + 2 * i + sub, F - 1);
        load_issue(lx[i], hd4 + ((size_t)lrow * K4 + kqc));
        lf0[i] = f0_all[lrow];
        lamp[i] = amplitudes[lrow];
      }
      // f0 of the frames before the chunk, for the fp64 phase prefix: float4 number lane + 64 rw of the row
      // (rows of up to 1024 frames in one go), the <= 3 frames past the last whole float4 on wavefront 0
      const float* __restrict__ f0row = f0_all + (size_t)lb * F;
      const int n4 = p.f0_vec ? min(lj0 >> 2, 256) : 0;
      const int m4 = lane + 64 * rw;
      const float4 pf = reinterpret_cast<const float4*>(p.f0_vec ? f0row : hd)[min(m4, max(n4 - 1, 0))];
      const int jt = (n4 << 2) + lane;
      const float ptail = f0row[min(jt, F - 1)];
      const float f0_first = f0row[0];
      DDSP_WT_STAMP(1);
      // ---------------- table of chunk tick+1: O and E on the quarter range -----------------------------------
      if (tick + 1 >= 0 && tick + 1 < n_my) {
#pragma unroll
       for (int rt = 0; rt < kWtRowTiles; ++rt) {
        // B: element e of lane (j = lane & 15, g = lane >> 4): plane[part][par][row 16 rt + j][32 ks + 8 g + e]
        const _Float16* bsrc = planes_all[(tick + 1) % 3] + (16 * rt + mi) * kWtPS + 8 * mg;
        f32x4 acc[2][2], accx[2][2];
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            acc[par][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx[par][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            const f16x8 bhi = *reinterpret_cast<const f16x8*>(bsrc + (0 * 2 + par) * kWtRows * kWtPS + 32 * ks);
            const f16x8 blo = *reinterpret_cast<const f16x8*>(bsrc + (1 * 2 + par) * kWtRows * kWtPS + 32 * ks);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              acc[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][tt][ks], bhi, acc[par][tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              accx[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][tt][ks], blo, accx[par][tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              accx[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[par][tt][ks], bhi, accx[par][tt], 0, 0, 0);
          }
        // D[row = 4 (lane >> 4) + reg][col = lane & 15]: this lane holds positions n0 .. n0+3 of table row mi
        float* trow = tab_all[(tick + 1) & 1] + (16 * rt + mi) * kWtTS + kWtH;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int n0 = 16 * (2 * rw + tt) + 4 * mg;
          const f32x4 so = acc[0][tt] + accx[0][tt] * (1.0f / kWtLoScale);     // odd harmonics
          const f32x4 se = acc[1][tt] + accx[1][tt] * (1.0f / kWtLoScale);     // even harmonics
          const f32x4 sp = so + se;                         // S(n)         = O + E
          const f32x4 sm = so - se;                         // S(T/2-1-n)   = O - E
          *reinterpret_cast<f32x4*>(trow + n0) = sp;
          *reinterpret_cast<f32x4*>(trow + (kWtHalf - 4 - n0)) = (f32x4){sm.w, sm.z, sm.y, sm.x};
          if (n0 == 0) {                                     // halos: S(-1-m) = -S(m), S(T/2+m) = -S(T/2-1-m)
            *reinterpret_cast<f32x4*>(trow - kWtH) = (f32x4){-sp.w, -sp.z, -sp.y, -sp.x};
            *reinterpret_cast<f32x4*>(trow + kWtHalf) = (f32x4){-sm.x, -sm.y, -sm.z, -sm.w};
          }
        }
       }
      }
      DDSP_WT_STAMP(2);
      // ---------------- rows of chunk tick+3: into the staging buffer --------------------------------------------
      {
        float* raw = raw_all[(tick + 3) & 1];
        const int r0 = rw * 8 + sub;
        load_settle(lx[0], lx[1], lx[2], lx[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *reinterpret_cast<ddsp_f32x4*>(raw + (r0 + 2 * i) * kWtRS + 4 * kq) = lx[i];
          if (kq == 0) *reinterpret_cast<float2*>(raw + (r0 + 2 * i) * kWtRS + 128) = make_float2(lf0[i], lamp[i]);
        }
        double part = (m4 < n4) ? ((double)pf.x + (double)pf.y) + ((double)pf.z + (double)pf.w) : 0.0;
        if (rw == 0 && jt < lj0 && (jt >> 2) == n4 && p.f0_vec) part += (double)ptail;
        // what the one-shot loads do not cover (rows longer than 1024 frames, unaligned rows): serially
        const int covered = p.f0_vec ? min(lj0, 1024) : 0;
        for (int j = covered + tid; j < lj0; j += 256) part += (double)f0row[j];
        const double psum = wave_sum_dpp(part);                // this wavefront's part of sum_{j < j0} f_j
        if (lane == 0) {
          reinterpret_cast<double*>(raw + kWtRows * kWtRS)[rw] = psum;
          if (rw == 0) raw[kWtRows * kWtRS + 8] = f0_first;
        }
      }
      if (tick + 3 < n_my - 1) DDSP_WT_ADVANCE(lb, lc);
      // ---------------- the per-frame phase tables of chunk tick+2 (its rows were staged a tick ago) ------------
      // one wavefront, lanes = frames; here, in the slack a T-wavefront has before the barrier, and not on the
      // S-wavefront that finishes last (profiles/r02o_*: that one set the length of the tick)
      if (tick + 2 >= 0 && tick + 2 < n_my) {
        if (rw == 3) wt_phase_tables(raw_all[(tick + 2) & 1], t_all[(tick + 2) % 3], lane, min(kWtFrames, F - qc * kWtFrames), K, p);
        DDSP_WT_ADVANCE(qb, qc);
      }
      DDSP_WT_STAMP(3);
      __syncthreads();
      DDSP_WT_STAMP(4);
    }
  } else {
    float ipsi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ipsi[u] = live ? WtPoly<W>::invpsi(4 * kq + u + 1) : 0.0f;
    const int arow0 = rw * 2 + sub;                // the chunk rows this lane works on in phase A: arow0, arow0 + 16
    int bb = first_b, bc = first_c;                // position of the chunk of the next phase B
    int ab = first_b, ac = first_c;                // position of the chunk of the next phase A

    for (int tick = -3; tick < n_my; ++tick) {
      DDSP_WT_STAMP(0);
      if (tick >= 0) {
        // ---------------- phase B of chunk tick: tiles of 64 samples, lanes = samples ----------------------
        const int j0 = bc * kWtFrames;
        const int nfr = min(kWtFrames, F - j0);
        const int row0 = bb * F + j0;
        DDSP_WT_ADVANCE(bb, bc);
        const float* tab = tab_all[tick & 1];
        const _Float16* planes = planes_all[tick % 3];
        const ChunkTables& t = t_all[tick % 3];
        const int hop = p.hop;
        const float inv_hop = 1.0f / (float)hop;
        const bool chunk_cross = __builtin_amdgcn_readfirstlane(t.cross) != 0;      // one look per tick, not per tile
        const int tiles_per_frame = hop >> 6;
        const int n_tiles = nfr * tiles_per_frame;
        // up to four tiles per wavefront move through the stages together (u = 0 .. NT-1), see wt_taps
        auto tiles = [&](int tile, auto nt_tag) {
          constexpr int NT = decltype(nt_tag)::value;          // tiles tile, tile + 8, .. tile + 8 (NT - 1)
          int q[kWtNT], r[kWtNT];
          double cyc[kWtNT];
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            const int tl = tile + 8 * u;
            q[u] = ONE_TILE ? tl : tl / tiles_per_frame;
            r[u] = ONE_TILE ? lane : (tl - q[u] * tiles_per_frame) * 64 + lane;
            const double rr = (double)r[u];
            // inclusive cumsum of f[t]/sr inside the frame: (r+1) w + r (r+1) dw, in revolutions
            cyc[u] = t.theta[q[u]] + (rr + 1.0) * (t.w[q[u]] + t.dw[q[u]] * rr);
          }
          float theta[kWtNT], z[kWtNT], z2[kWtNT];
          bool neg[kWtNT];
          const float* t0[kWtNT];
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            theta[u] = (float)__builtin_amdgcn_fract(cyc[u]);               // v_fract_f64: [0, 1]
            neg[u] = theta[u] >= 0.5f;                                    // S(1 - theta) = -S(theta)
            const float th = neg[u] ? 1.0f - theta[u] : theta[u];         // [0, 0.5]
            const float pos = fmaf(th, (float)kWtT, -0.5f);               // table coordinate, [-0.5, 255.5]
            const float fl = floorf(pos);
            z[u] = (pos - fl) - 0.5f;
            z2[u] = z[u] * z[u];
            t0[u] = tab + q[u] * kWtTS + kWtH + (int)fl;                  // (int)fl in [-1, 255]
          }
#ifdef DDSP_WT_TILE_STAMPS
          if (tile == rw) DDSP_WT_STAMP(5);
#endif                    // phase and table coordinate known
          float acc0[kWtNT] = {0.0f, 0.0f, 0.0f, 0.0f}, acc1[kWtNT] = {0.0f, 0.0f, 0.0f, 0.0f};
          wt_taps<W, 0, NT>(t0, z, z2, acc0, acc1);
#ifdef DDSP_WT_TILE_STAMPS
          if (tile == rw) DDSP_WT_STAMP(6);
#endif                    // taps read and accumulated
          float out[kWtNT], w_cur[kWtNT], w_next[kWtNT], lerp[kWtNT];
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            lerp[u] = (float)r[u] * inv_hop;
            // frame-rate -> audio-rate amplitude envelope: weight of frame j+1 is lerp ('linear', core.resample)
            // or the periodic Hann(2 hop)[r] ('window', core.py:696-698)
            w_next[u] = p.amp_linear ? lerp[u] : 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp[u]);
            w_cur[u] = 1.0f - w_next[u];
            const float v = w_cur[u] * acc0[u] + w_next[u] * acc1[u];
            out[u] = neg[u] ? -v : v;
          }
          if (chunk_cross)
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            const int kA = __builtin_amdgcn_readfirstlane(t.kA[q[u]]);
            const int kN = __builtin_amdgcn_readfirstlane(t.kN[q[u]]);
            if (kA < kN) {         // harmonics crossing Nyquist inside this frame: audio-rate mask, TF's fp32 op order
              const float fj = t.f0[q[u]], fj1 = t.f0[q[u] + 1];
              for (int k = kA; k < kN; ++k) {
                const float kf = (float)(k + 1);
                const float top = fj * kf, bot = fj1 * kf;
                const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp[u]));
                const _Float16* pl = planes + ((k & 1) * kWtRows + q[u]) * kWtPS + (k >> 1);
                const float c0 = (float)pl[0] + (float)pl[2 * kWtRows * kWtPS] * (1.0f / kWtLoScale);
                const float c1 = (float)pl[kWtPS] + (float)pl[2 * kWtRows * kWtPS + kWtPS] * (1.0f / kWtLoScale);
                const float ak = (w_cur[u] * c0 + w_next[u] * c1) * WtPoly<W>::psi(k + 1);
                const float sv = sin_rev(fmaf(theta[u], kf, -rintf(theta[u] * kf)));     // exact fractional part of k theta
                if (fk >= p.nyquist) out[u] = fmaf(-ak, sv, out[u]);
              }
            }
          }
#ifdef DDSP_WT_TILE_STAMPS
          if (tile == rw) DDSP_WT_STAMP(7);
#endif                    // envelope, Nyquist corrections done
#pragma unroll
          for (int u = 0; u < NT; ++u) audio[(size_t)(row0 + q[u]) * hop + r[u]] = out[u];            // N == F * hop
        };
        for (int tile = rw; tile < n_tiles; tile += 8 * kWtNT) {
          const int left = (n_tiles - tile + 7) >> 3;          // tiles tile, tile + 8, ... still inside the chunk
          if (left >= 4) tiles(tile, std::integral_constant<int, 4>{});
          else if (left == 3) tiles(tile, std::integral_constant<int, 3>{});
          else if (left == 2) tiles(tile, std::integral_constant<int, 2>{});
          else tiles(tile, std::integral_constant<int, 1>{});
        }
      }
      DDSP_WT_STAMP(1);
      if (tick + 2 >= 0 && tick + 2 < n_my) {
        // ---------------- phase A of chunk tick+2: controls of rows j0 .. j0+31 (clamped at F-1) -> planes ----
        // core.exp_sigmoid (core.py:386-404), remove_above_nyquist on f0 * [1..K] (core.py:899-903, 1028-1045),
        // safe_divide by the row sum (core.py:905-907, 207-210), amplitudes * distribution (core.py:1097)
        const int j0 = ac * kWtFrames;
        const int nfr = min(kWtFrames, F - j0);
        const int crow0 = ab * F + j0;
        DDSP_WT_ADVANCE(ab, ac);
        const float* raw = raw_all[(tick + 2) & 1];
        _Float16* planes = planes_all[(tick + 2) % 3];
#pragma unroll
        for (int h = 0; h < kWtRowTiles; ++h) {        // independent rows: their chains interleave
          const int arow = arow0 + 16 * h;
          const int crow = crow0 + arow;               // this lane's (batch * frame) row, if arow < nfr
          const float4 xv = *reinterpret_cast<const float4*>(raw + arow * kWtRS + 4 * kq);
          const float2 fa2 = *reinterpret_cast<const float2*>(raw + arow * kWtRS + 128);
          const float f0r = fa2.x;
          float x[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            x[u] = exp_sigmoid_fast(x[u], kLog10, 2.0f, 1e-7f);
            if (!live || f0r * (float)(4 * kq + u + 1) >= p.nyquist) x[u] = 0.0f;
          }
          float part = (x[0] + x[1]) + (x[2] + x[3]);
          part += dpp_mov0<0xB1, 0xF>(part);      // quad_perm [1,0,3,2]
          part += dpp_mov0<0x4E, 0xF>(part);      // quad_perm [2,3,0,1]
          part += dpp_mov0<0x141, 0xF>(part);     // row_half_mirror
          part += dpp_mov0<0x140, 0xF>(part);     // row_mirror: every lane holds its 16-lane row's sum
          {                                        // + the other row of the pair, through SGPRs (no LDS round trip)
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 0));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 16));
            const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 32));
            const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, part), 48));
            part = sub ? s2 + s3 : s0 + s1;
          }
          const float inv = __builtin_amdgcn_rcpf(part == 0.0f ? 1e-7f : part);
          const float a_ctl = exp_sigmoid_fast(fa2.y, kLog10, 2.0f, 1e-7f);
          const float a = a_ctl * inv;
          // the controls dict (return_outputs_dict=True, how dags.py:171-173 calls every processor); the halo row
          // belongs to the next chunk
          if (ctl_hd != nullptr && arow < nfr) {
            if (live)
              reinterpret_cast<float4*>(ctl_hd)[(size_t)crow * K4 + kq] = make_float4(x[0] * inv, x[1] * inv, x[2] * inv, x[3] * inv);
            if (kq == 0) ctl_amp[crow] = a_ctl;
          }
          // c_k = a_k / psi_hat(k) as hi + lo / 2048, two fp16 numbers each
          // (hi rounded toward zero by v_cvt_pkrtz_f16_f32: lo takes up the rest)
          float c[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) c[u] = a * x[u] * ipsi[u];
          _Float16* dst = planes + arow * kWtPS + 2 * kq;
#pragma unroll
          for (int par = 0; par < 2; ++par) {       // k odd: c[0], c[2] (k' = 2 kq, 2 kq + 1); k even: c[1], c[3]
            const h16x2 hi = __builtin_amdgcn_cvt_pkrtz(c[par], c[par + 2]);
            const h16x2 lo = __builtin_amdgcn_cvt_pkrtz((c[par] - (float)hi[0]) * kWtLoScale,
                                                        (c[par + 2] - (float)hi[1]) * kWtLoScale);
            *reinterpret_cast<h16x2*>(dst + (0 * 2 + par) * kWtRows * kWtPS) = hi;
            *reinterpret_cast<h16x2*>(dst + (1 * 2 + par) * kWtRows * kWtPS) = lo;
          }
        }
      }
      DDSP_WT_STAMP(3);
      __syncthreads();
      DDSP_WT_STAMP(4);
    }
  }
#undef DDSP_WT_STAMP
#undef DDSP_WT_ADVANCE
}

bool harm_table_ok(int F, int K, int N, const void* hd, const void* ctl_amp, const void* ctl_hd, unsigned flags,
                   int inputs_are_controls) {
  if (flags & DDSP_HARM_DIRECT_SUM) return false;
  if (!(flags & DDSP_HARM_SCALE_EXP_SIGMOID) || !(flags & DDSP_HARM_NORMALIZE_NYQUIST)) return false;
  if (inputs_are_controls || (ctl_amp == nullptr) != (ctl_hd == nullptr) || (flags >> 24) != 0) return false;
  return (N % F) == 0 && ((N / F) % 64) == 0 && K >= 4 && K <= 128 && (K % 4) == 0 &&
         (((uintptr_t)hd | (uintptr_t)ctl_hd) & 15) == 0;
}

int launch_harm_table(const float* amplitudes, const float* hd, const float* f0, float* audio, float* ctl_amp,
                      float* ctl_hd, int B, int F, int K, int N, int sample_rate, unsigned flags, hipStream_t st) {
  TableArgs p;
  p.B = B; p.F = F; p.K = K; p.N = N; p.hop = N / F;
  p.chunks_per_row = (F + kWtFrames - 1) / kWtFrames;
  p.n_chunks = B * p.chunks_per_row;
  p.nyquist = (float)(sample_rate / 2.0);
  p.nyq_lo = p.nyquist * (1.0f - 4e-6f);
  p.nyq_hi = p.nyquist * (1.0f + 4e-6f);
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  p.f0_vec = ((F & 3) == 0 && ((uintptr_t)f0 & 15) == 0) ? 1 : 0;
  p.inv_sr = 1.0 / (double)sample_rate;
  p.inv_2hop = 0.5 / (double)p.hop;
  p.hop_d = (double)p.hop;
  p.half_hm1 = ((double)p.hop - 1.0) * 0.5;
  // persistent grid: one block of 12 wavefronts per CU
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  const dim3 grid((unsigned)(p.n_chunks < n_cu ? p.n_chunks : n_cu)), block(768);
  // DDSP_EXP_TABLE_TIMELINE=1: block 0 records shader-clock stamps per tick; printed after a synchronisation
  static const bool timeline = getenv("DDSP_EXP_TABLE_TIMELINE") != nullptr;
  static long long* dbg_buf = nullptr;
  p.dbg = nullptr;
  if (timeline) {
    if (!dbg_buf && hipMalloc(&dbg_buf, 3 * 64 * 8 * sizeof(long long)) != hipSuccess) dbg_buf = nullptr;
    if (dbg_buf) (void)hipMemsetAsync(dbg_buf, 0, 3 * 64 * 8 * sizeof(long long), st);
    p.dbg = dbg_buf;
  }
  hipEvent_t ev0, ev1;
  profile_kernel_events(kHarmTable, &ev0, &ev1);
#define DDSP_LAUNCH_TABLE(W, NK)                                                                              \
  do {                                                                                                        \
    if (p.hop == 64)                                                                                          \
      hipExtLaunchKernelGGL((harm_table_kernel<W, NK, true>), grid, block, 0, st, ev0, ev1, 0, amplitudes, hd, f0, \
                            audio, ctl_amp, ctl_hd, p);                                                               \
    else                                                                                                      \
      hipExtLaunchKernelGGL((harm_table_kernel<W, NK, false>), grid, block, 0, st, ev0, ev1, 0, amplitudes, hd, f0, \
                            audio, ctl_amp, ctl_hd, p);                                                               \
  } while (0)
  // the 6-tap window holds its 6.3e-6 up to K = 100 (T / 2K >= 2.56); denser spectra take 8 taps
  if (K <= 64) DDSP_LAUNCH_TABLE(6, 1);
  else if (K <= 100) DDSP_LAUNCH_TABLE(6, 2);
  else DDSP_LAUNCH_TABLE(8, 2);
#undef DDSP_LAUNCH_TABLE
  if (p.dbg) {
    static long long host[3 * 64 * 8];
    if (hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(host, p.dbg, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess) {
      const char* names[3] = {"T0", "S0", "S7"};
      const long long t0 = host[0];
      for (int w = 0; w < 3; ++w)
        for (int i = 0; i < 64 && host[(w * 64 + i) * 8] != 0; ++i) {
          const long long* r = host + (w * 64 + i) * 8;
          fprintf(stderr, "[timeline] %s tick %3d  start %8lld  +%6lld +%6lld +%6lld  barrier +%6lld", names[w], i - 3,
                  r[0] - t0, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3]);
          if (w > 0 && r[5] != 0)      // S-wavefronts: inside phase B's first tile pair (since tick start)
            fprintf(stderr, "   | B: coord +%5lld taps +%5lld env +%5lld store..end +%5lld", r[5] - r[0], r[6] - r[5],
                    r[7] - r[6], r[1] - r[7]);
          fprintf(stderr, "\n");
        }
    }
  }
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

}  // namespace ddsp
