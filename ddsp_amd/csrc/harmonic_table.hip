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
// so two [128 x K/2] products per row give the whole table.  v_mfma_f32_16x16x32_f16 does them on fp16 hi / lo pairs
// (22 bits per operand, fp32 accumulation: see the kernel) with the constant factor resident in registers for the life
// of the block.
//
// Per sample the VALU work drops from ~4 K flop-instructions to ~55 (phase, W polynomial weights, 2 W taps; packed
// FMAs), independent of K; error vs exact arithmetic <= 6.3e-6 * sum_k a_k (W = 6, K <= 100), 6.5e-6 (W = 8,
// K <= 128), 5.3e-6 (W = 10, K <= 200), smaller than the sine recurrence's 3.1e-5.
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
#include "wavetable_coeffs.h"      // (the window's polynomials: compile-time constants; 1 / psi_hat: host tables for the fragments)
#include "harm_table_frags.h"
#include <mutex>

namespace ddsp {

// the constant factors (harm_table_frags.h), filled in once per device by wt_upload_fragments() below
static __device__ WtFragSet kWtFragSet6;      // six taps (K <= 100): table sizes 512 .. 64, 92 KB
static __device__ WtFragSet kWtFragSet8;      // eight taps (K <= 128)
static __device__ WtFragsWide kWtFragsWide;   // ten taps, 129 .. 200 harmonics: 128 KB, streamed from L2 every tick

constexpr int kWtT = 512;            // table points per revolution: the LARGEST table (a segment with few live harmonics takes 256, 128 or 64)
constexpr int kWtHalf = kWtT / 2;    // the table holds p in [-kWtH, T / 2 + kWtH); the other half is its mirror image
constexpr int kWtNQ = kWtT / 4;      // positions produced by the matrix product
// halo entries on either side of a table row (>= W/2, a multiple of 4) and the row stride in floats (4*odd, so 16 rows'
// b128 writes spread over the banks), by window width: 4 and 268 up to eight taps, 8 and 276 for ten
template <int W> struct WtGeom {
  static constexpr int H = W <= 8 ? 4 : 8;
  static constexpr int TS = kWtHalf + 2 * H + 4;
};
constexpr int kWtRowTiles = 2;       // MFMA N-tiles of 16 amplitude rows per chunk
constexpr int kWtRows = 16 * kWtRowTiles;    // amplitude rows per chunk
constexpr int kWtFrames = kWtRows - 1;       // frames per chunk (31): row r+1 is the "next" row of frame r
constexpr int kWtNT = 4;             // tiles of 64 samples an S-wavefront carries through phase B together
// row stride of an amplitude plane (fp16 elements; odd / even harmonics apart): 144 B for two k-steps (K <= 128), 208 B
// for four (K <= 208: three plane buffers of 136 would not fit the LDS next to the tables; harm_table_frags.h)
template <int NK> struct WtPlane { static constexpr int PS = NK <= 2 ? 72 : 104; };
constexpr float kWtLoScale = 2048.0f; // x = hi + lo / 2048 in two fp16 numbers

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));      // what v_cvt_pkrtz_f16_f32 returns

struct ChunkTables {
  double theta[kWtRows], w[kWtRows], dw[kWtRows];
  float f0[kWtRows + 2];
  int kA[kWtRows], kN[kWtRows];
  int cross;           // any frame of the chunk with a harmonic crossing Nyquist inside it (kA < kN)
  // the FIRST crossing harmonic kA of a frame, ready-made by the wavefront that builds these tables (lanes = frames): the two
  // rows' amplitudes d am, the frame's end frequencies of that harmonic as top and (bot - top), and k as a float.  The
  // interpolators used to gather all of that per tile - two index reads, two f0, two amplitudes, four 2-byte plane reads -
  // behind three waits: a frame with a crossing harmonic cost twice a frame without (round 3 measured this at 6 - 9 % per
  // launch at 200 / 250 / 333 Hz and took it out again when two bit-equality tests failed once - the packed-FMA fault of
  // phase B, which had nothing to do with it: profiles/r04_packed_fma_glitch.txt).
  float4 cx[kWtRows];  // {d0 am0, d1 am1, fj k, (fj1 - fj) k}
  float ck[kWtRows];   // k
};

template <int W> struct WtPoly;
template <> struct WtPoly<6> {
  // K512: the harmonics 512 points carry at this window's error bound (T / 2K >= 2.56 for six taps; launch_harm_table);
  // a table of T points carries K512 T / 512
  static constexpr int DE = kWtDegE6, DO = kWtDegO6, K512 = 100;
  static constexpr float e(int p, int d) { return kWtE6[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO6[p * (DO + 1) + d]; }
};
template <> struct WtPoly<8> {
  static constexpr int DE = kWtDegE8, DO = kWtDegO8, K512 = 128;
  static constexpr float e(int p, int d) { return kWtE8[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO8[p * (DO + 1) + d]; }
};

template <> struct WtPoly<10> {      // 129 .. 200 harmonics on the same 512 points (oversampling 1.28; tools/gen_wavetable_coeffs.py)
  static constexpr int DE = kWtDegE10, DO = kWtDegO10, K512 = 200;
  static constexpr float e(int p, int d) { return kWtE10[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO10[p * (DO + 1) + d]; }
};

// =====================================================================================================================
// harm_table_kernel: one block of SIXTEEN wavefronts per CU, three roles.
//
// What the round-3 microbenchmarks say about this chip (profiles/r03a_*, r03b_*): one wavefront issues at most one
// instruction per ~6.5 clocks whatever its kind - vector, scalar or LDS; a SIMD issues one per ~3.4 clocks from two busy
// wavefronts, ~2.5 from three, ~2.15 from four, 1.6 from eight.  The round-2 kernel (12 wavefronts: 4 T + 8 S, 161 KB of
// LDS, rows staged through LDS by the T-wavefronts) had two busy wavefronts per SIMD - every instruction it executed
// cost ~1.6 times its price at full occupancy - and a quarter of its instructions were scalar bookkeeping, which costs
// as much as vector work.  Here (51.4 -> 40.6 us per launch at batch 128, profiles/r03c_* .. r03h_*):
//   * 4 T (tabulators) + 8 interpolators + 4 row makers, 128 VGPRs: four busy wavefronts per SIMD, one of each kind
//     of work (matrix core, LDS reads + FMAs, transcendentals) on every SIMD.  T: the constant factor (64 VGPRs), the
//     MFMAs and the table; T3 also the per-frame fp64 phase tables, T0 the chunk descriptors.  Interpolators: phase B,
//     four tiles at a time, on packed FMAs.  Row makers: phase A, four row pairs at a time (interleaved: a row is one
//     long chain of dependent instructions), the rows fetched straight from HBM into registers a tick ahead - no LDS
//     staging buffer, no staging writes, no second read;
//   * a block owns a CONTIGUOUS run of frames, cut into chunks of equal length (<= 31): the fp64 phase prefix is
//     carried from chunk to chunk instead of re-summed from the start of the row every tick, and a batch of 32 clips
//     is 5 ticks of 25 frames per block instead of 4.1 -> 5 ticks of 31;
//   * one 16-byte chunk descriptor read per wavefront and tick (issued at the top of the tick, taken into scalar
//     registers at its end): the scalar work per tick is a fraction of what it was;
//   * the pipeline is three stages deep (phase A, tabulate, phase B): two fill ticks instead of three;
//   * phase A without v_readlane (v_permlane16_swap for the sum of a row pair) and without compare / select for the
//     Nyquist mask (v_med3_f32); phase B folds theta into [0, 1/2] with |x| and restores the sign with a xor;
//   * processors.Add can ride in phase B (add_in): the other signal's samples are fetched at the top of a tile and
//     added at the store.
// Second half of round 3 (same-session A/B runs, profiles/r03o_* .. r03t_*; 41.0 -> 37.5 us): the row makers fetch their
// rows a whole tick ahead (pinned loads, two register sets; fetched at the end of the previous tick, the block's last
// wavefronts began every tick with the HBM latency); the frame's amplitude is made by tabulator 1 and multiplied into the
// table sums (the planes hold the normalised distribution); phase B's tap pairs are packed as neighbours in the table row
// (no register moves), its stores and fused-Add loads go through a scalar base, the tile slots are dealt by SIMD load;
// any K <= 128 (rows that are not 16 bytes apart: ROWS16 = false).  What did NOT pay: fewer row-maker instructions as such
// (the SIMD serves its wavefronts oldest first; the row makers run in what is left whatever they have to do), the phase
// tables removed from tabulator 3 or threaded between its MFMAs, s_setprio in any arrangement.
// 129 .. 200 harmonics (WIDE = NK > 2; BASELINE configs[4]: 48 kHz, 200 harmonics, frames of 192 samples) run here too since the end
// of round 3 instead of on the direct sum (88 against 212 us at batch 32, profiles/r03u_*): the SAME 512 points read through ten
// taps (oversampling 1.28; the window's transform is down at (T - K) / T just past its cut-off: <= 5.3e-6 per harmonic, 1 / psi_hat
// up to 27 at k = 200 - tools/gen_wavetable_coeffs.py), halo 8, planes of 104 halves per row (three buffers of 136 do not fit), four
// k-steps whose fragments - 128 registers' worth - are fetched from L2 for one parity and row tile at a time and used as they land
// (a frame of three tiles leaves the tabulators the time), rows spread over whole wavefronts (lane = four harmonics, eight rows per
// row maker in two passes).  Their pinned loads carry their own wait states (common.h load_issue_spaced): the first build read
// fragments through stale scalar bases on the chip and nowhere else (profiles/r03u_inline_asm_scalar_hazard.txt).  The instances for
// K <= 128 are instruction for instruction what they were (only prologue scheduling differs in seven of the 24).
// Measured and dropped on the way (timelines under profiles/): twelve identical S-wavefronts (r03d: 45.4 us - a wavefront
// that does three tiles AND a row pair is ~5000 clocks long, the T-wavefronts idle half the tick); a row pair on each
// T-wavefront as well (r03e: their MFMA section stretches from 2600 to 4000 clocks and they become the critical
// path); three row pairs per row maker one after the other (r03f: 1500 clocks each - interleaved, four take 3900).
// Results are independent of how the frames are cut into chunks: every quantity of a frame depends on its own two rows
// and on an fp64 prefix that is exact for any f0 a synthesiser sees; every product-sum is spelled out, so which
// template instance a tile runs in cannot change a bit (tests/test_gpu_contract_shapes.py compares rows run alone, in a
// batch of 32 and in a batch of 128).
#ifndef DDSP_WT_WALKER
#define DDSP_WT_WALKER 2
#endif
#ifndef DDSP_WT_SLOTS
#define DDSP_WT_SLOTS 0x73256104u
#endif
// Which wavefront does what is chosen for the four SIMDs' totals, not the wavefronts': a SIMD issues for its four
// wavefronts in turn, oldest first, and a tick ends when the busiest SIMD is done (r03n: the row maker next to
// tabulator 3, whose SIMD also builds the phase tables, arrives at the barrier 900 clocks after the others).
constexpr int kWtWalker = DDSP_WT_WALKER;        // the tabulator that walks the chunk descriptors
#ifndef DDSP_WT_SIZER
#define DDSP_WT_SIZER 0
#endif
constexpr int kWtSizer = DDSP_WT_SIZER;          // the tabulator that chooses the table sizes (wt_table_size)
constexpr unsigned kWtSlots = DDSP_WT_SLOTS;     // nibble sw: the tile slot of interpolator sw (wavefront 4 + sw, SIMD sw % 4):
                                                 // the slots that come up short (7, 6, then 5, 4) are SIMD 3's and SIMD 0's
struct WtDesc { int b, j0, nfr, fresh; };        // a chunk: frames j0 .. j0 + nfr - 1 of row b; nfr == 0: none

struct WtSizeThresholds { float t64, t128, t256; };      // wt_table_size below

struct TableArgs {
  int B, F, K, N, hop;
  int total_frames, frames_per_block;
  FastDiv f_div, tpf_div, seg_div;        // F; hop / 64; kWtSegment
  float nyquist, nyq_lo, nyq_hi;
  WtSizeThresholds size_thr;      // the table size of a segment from its smallest f0
  int amp_linear;
  int ragged;          // hop % 64 != 0: a frame's last tile is cut short (lanes past the frame's end store nothing)
  int rows16;          // rows of hd (and of the controls out) are 16 bytes apart and aligned: K % 4 == 0, aligned bases
  double inv_sr, inv_2hop, hop_d, half_hm1;
  long long* dbg;      // DDSP_EXP_TABLE_TIMELINE=1: shader-clock stamps of block 0, [wavefront][tick + 2][stamp]; or null
};

struct WtWalk { int pos, pos_first, end, seg_left, base, rem, b, j; };

// TABLE SIZE (round 4).  The table positions of neighbouring samples are T f0 / sr entries apart; on 512 points that is 6.4 at
// 200 Hz, 10.7 at 333 Hz, 16 at 500 Hz (16 kHz) - strides at which the 32 lanes of an LDS read meet in a handful of banks (the
// table reads then cost 2.5 - 4 times their 70 Hz time: the kernel's f0 dependence of rounds 2-3, 37 us at 70 Hz against 60 - 74
// at the divisors of Nyquist).  But a frame whose f0 is high has few harmonics below Nyquist, and a table of T points carries
// K512 T / 512 of them at the same error bound (the window and its oversampling ratio T / 2K are what they were): frames are
// therefore tabulated on T = 512, 256, 128 or 64 points by their f0, which keeps the stride between 1.3 and 5.1 entries at
// any f0, halves (quarters, ..) the matrix product and the table writes - and makes the table reads of a 500 Hz note what they
// are at 70 Hz.  T must be the same for the two rows of a frame, hence for runs of frames - and which frames share a T must not
// depend on how the batch is cut into blocks and chunks, or a row run alone would differ in its bits from the row run in a
// batch.  So: a row is cut into SEGMENTS of kWtSegment frames at fixed positions (j / kWtSegment), T is a function of the
// segment's f0 alone (the smallest over its frames and the one behind them: wt_table_size), and the walker never lets a chunk
// straddle a segment boundary (two chunks of 31 frames per full segment).
#if defined(DDSP_EXP_SEGMENT)
constexpr int kWtSegment = DDSP_EXP_SEGMENT;       // experiment: other segment lengths (a huge one: no cuts)
#else
constexpr int kWtSegment = 2 * kWtFrames;
#endif

// the next chunk of the block's run of frames (one wavefront, wave-uniform arithmetic)
__device__ __forceinline__ WtDesc wt_next_chunk(WtWalk& w, const TableArgs& p) {
  WtDesc d{0, 0, 0, 0};
  if (w.pos >= w.end) return d;
  if (w.seg_left == 0) {                      // a new piece of a segment: cut it into equal chunks
    uint32_t j;
    const int b = (int)fastdiv((uint32_t)w.pos, p.f_div, j);
    d.fresh = (w.pos == w.pos_first || (int)j == 0) ? 1 : 0;      // the block's first chunk, or the first of a row: the prefix is summed afresh
    w.b = b;
    w.j = (int)j;
    uint32_t in_seg;
    (void)fastdiv(j, p.seg_div, in_seg);
    w.seg_left = min(min(w.end - w.pos, p.F - w.j), kWtSegment - (int)in_seg);
    const int n = (w.seg_left + kWtFrames - 1) / kWtFrames;
    w.base = w.seg_left / n;
    w.rem = w.seg_left - w.base * n;
  }
  const int len = w.base + (w.rem > 0 ? 1 : 0);
  if (w.rem > 0) --w.rem;
  d.b = w.b; d.j0 = w.j; d.nfr = len;
  w.pos += len; w.j += len; w.seg_left -= len;
  return d;
}

// The table size of a segment from the smallest f0 of its frames (and of the frame behind them).  floor(nyquist / f0) bounds the
// harmonics any of its rows has below Nyquist (core.remove_above_nyquist on f0 k, core.py:899-903); a table of T points carries
// K512 T / 512.  As three thresholds on f0 made once per launch (wt_size_thresholds): T points suffice iff f0_min >= thr[T]
// (-inf where K itself is small enough; a NaN or an f0 <= 0 with many harmonics compares false: 512 points).
__device__ __forceinline__ int wt_table_size(float f0_min, const WtSizeThresholds& th) {
  return f0_min >= th.t64 ? 64 : f0_min >= th.t128 ? 128 : f0_min >= th.t256 ? 256 : kWtT;
}
inline WtSizeThresholds wt_size_thresholds(int K512, int K, float nyq_hi) {
  auto thr = [&](int T) -> float {
    const int kmax = K512 * T / 512;                  // what T points carry (100 -> 50, 25, 12; 128 -> 64, 32, 16)
    if (K <= kmax) return -__builtin_inff();
    // floor(nyq_hi / f0) <= kmax  <=>  f0 > nyq_hi / (kmax + 1); a hair above it
    return (float)((double)nyq_hi / (double)(kmax + 1) * (1.0 + 1e-6));
  };
  return WtSizeThresholds{thr(64), thr(128), thr(256)};
}

__device__ __forceinline__ WtDesc wt_read_desc(const WtDesc* ring, int slot) {
  const int4 v = *reinterpret_cast<const int4*>(ring + slot);
  WtDesc d;
  d.b = __builtin_amdgcn_readfirstlane(v.x);
  d.j0 = __builtin_amdgcn_readfirstlane(v.y);
  d.nfr = __builtin_amdgcn_readfirstlane(v.z);
  d.fresh = __builtin_amdgcn_readfirstlane(v.w);
  return d;
}

// ---- phase B on pairs of fp32 values: one FMA per half, spelled out ---------------------------------------------------------
// Rounds 2-3 issued these as v_pk_fma_f32 assembly statements (two FMAs per issued instruction, op_sel / op_sel_hi routing the
// halves) and that was the source of the run-to-run differences the round-3 bit-equality tests met once: on the MI355X a
// wavefront running such packed FMAs back to back now and then gets ONE HALF of a result wrong in its LAST SIXTEEN LANES - a
// window weight off in its low bits, the error of both table rows' accumulators in the same half - while a tabulator of the same
// SIMD executes MFMAs: once in 10^6 .. 10^9 tiles for the shapes of the K <= 128 instances, in 5 - 20 % of all launches for the
// 129 .. 200-harmonic instances (tools/stress_determinism.py reproduces it at will; with one extra vector load per tabulator and
// tick, -DDDSP_EXP_T_DUMMY_LOADS=1, every launch of a frame-size-128 shape differs somewhere).  What it is NOT (each tested on
// the chip, profiles/r04_packed_fma_glitch.txt): a missing wait state around the statements (s_nop 1 in front of every one: no
// change), a result pair placed on a source pair whose halves cross (early-clobber / tied constraints: no change), the
// v_permlane32_swap of the wide instances' row makers, a load landing in a register still in use (the same statements without
// the packed instructions: no event in any configuration), an LDS race (theta, z, the table offset and the envelope weight of a
// differing sample are bit-equal between the two launches; only the accumulators' halves differ).  The same arithmetic as
// plain v_fma_f32 - below - never showed an event, in the amplified configurations either, and costs 0.7 us of 36.6 at batch
// 128 (nothing at batch 32): -DDDSP_EXP_PACKED_PHASE_B brings the assembly statements back for whoever wants to look again.
// The compiler's own packed instructions in the row makers (v_pk_mul / v_pk_add / v_pk_fma_f32 of phase A, a dozen per row
// pair between transcendentals) were never hit - not in 10^4 launches in which the interpolators were hit every time.
// tests/test_isa_guards.py keeps assembly-statement packed FMAs out of the instruction stream.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#if defined(__AMDGCN__) && defined(DDSP_EXP_PACKED_PHASE_B)
#define DDSP_WT_PK_ASM 1
#else
#define DDSP_WT_PK_ASM 0
#endif
// (e, o) <- (e, o) * z^2 + (ce, co), zz = (z, z^2)
__device__ __forceinline__ f32x2 wt_pk_horner(f32x2 eo, f32x2 zz, f32x2 coef) {
#if DDSP_WT_PK_ASM
  f32x2 r;
  __asm__("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(eo), "v"(zz), "v"(coef));
  return r;
#else
  return (f32x2){fmaf(eo[0], zz[1], coef[0]), fmaf(eo[1], zz[1], coef[1])};
#endif
}
// the window weights of a tap pair: (e + z o, e - z o)
__device__ __forceinline__ f32x2 wt_pk_weights(f32x2 eo, f32x2 zz) {
#if DDSP_WT_PK_ASM
  f32x2 r;
  __asm__("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(eo), "v"(zz));
  return r;
#else
  return (f32x2){fmaf(eo[1], zz[0], eo[0]), fmaf(eo[1], -zz[0], eo[0])};
#endif
}
__device__ __forceinline__ f32x2 wt_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
#if DDSP_WT_PK_ASM
  f32x2 r;
  __asm__("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return (f32x2){fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
#endif
}
// the fp16 pair (c2048[0] - 2048 h[0], c2048[1] - 2048 h[1]), c2048 = 2048 c: what hi = h leaves of c, scaled - the
// differences are exact in fp32, the fp16 halves of h are read in place and the results written as fp16 halves
// (v_fma_mixlo_f16 / v_fma_mixhi_f16: no conversion instructions)
__device__ __forceinline__ h16x2 wt_rest_halves(f32x2 c2048, h16x2 h) {
#if defined(__AMDGCN__)
  h16x2 r;
  const float m = -kWtLoScale;
  __asm__("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(r) : "v"(h), "v"(m), "v"(c2048[0]));
  __asm__("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(m), "v"(c2048[1]));
  return r;
#else
  return (h16x2){(__fp16)fmaf((float)h[0], -kWtLoScale, c2048[0]), (__fp16)fmaf((float)h[1], -kWtLoScale, c2048[1])};
#endif
}
// (int)floor(x) in one instruction
__device__ __forceinline__ int wt_floor_int(float x) {
#if defined(__AMDGCN__)
  int r;
  __asm__("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return (int)floorf(x);
#endif
}
// (a0 b1 + c0, a1 b0 + c1): the halves of b swapped
__device__ __forceinline__ f32x2 wt_pk_fma_swap(f32x2 a, f32x2 b, f32x2 c) {
#if DDSP_WT_PK_ASM
  f32x2 r;
  __asm__("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return (f32x2){fmaf(a[0], b[1], c[0]), fmaf(a[1], b[0], c[1])};
#endif
}
// the window weights of two tap pairs at once, E = (e_a, e_b), O = (o_a, o_b): E + z O and E - z O
__device__ __forceinline__ f32x2 wt_pk_plus(f32x2 o, f32x2 zz, f32x2 e) {
#if DDSP_WT_PK_ASM
  f32x2 r;
  __asm__("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(o), "v"(zz), "v"(e));
  return r;
#else
  return (f32x2){fmaf(o[0], zz[0], e[0]), fmaf(o[1], zz[0], e[1])};
#endif
}
__device__ __forceinline__ f32x2 wt_pk_minus(f32x2 o, f32x2 zz, f32x2 e) {
#if DDSP_WT_PK_ASM
  f32x2 r;
  __asm__("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(o), "v"(zz), "v"(e));
  return r;
#else
  return (f32x2){fmaf(-o[0], zz[0], e[0]), fmaf(-o[1], zz[0], e[1])};
#endif
}
// the four 16-byte loads of a k-step have landed when at most N younger loads are still in flight (loads return in order)
template <int N>
__device__ __forceinline__ void wt_frags_landed(ddsp_f32x4& a, ddsp_f32x4& b, ddsp_f32x4& c, ddsp_f32x4& d) {
#if defined(__AMDGCN__)
  __asm__ volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
#else
  (void)a; (void)b; (void)c; (void)d;
#endif
}
template <int N, class Fn>
__device__ __forceinline__ void wt_static_for(Fn&& f) {
  if constexpr (N > 0) {
    wt_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// The window polynomials as packed pairs, Horner order (index 0 = the highest power of z^2).  Tap pair P = taps -P and
// 1 + P, weights e_P(z^2) +- z o_P(z^2).  What sits next to each other in a table row is taps (-P - 1, -P) and
// (1 + P, 2 + P): pairs P + 1 and P are therefore evaluated TOGETHER - (e_{P+1}, e_P) and (o_{P+1}, o_P) - so that one
// packed FMA makes the weights of two neighbouring taps and one 8-byte LDS read brings their table values, with no
// register moves in between (the (e_P, o_P) packing of rounds 2-3 needed two v_mov per tap pair: a tenth of phase B).
// With an odd number of pairs (W = 6) pair 0 is left over and keeps the (e_0, o_0) packing: its two taps are neighbours.
template <int W> struct WtPkCoefs {
  static constexpr int NG = W / 4, MID = (W / 2) & 1, DE = WtPoly<W>::DE, DO = WtPoly<W>::DO;
  f32x2 e[NG][DE + 1], o[NG][DO + 1], m[DE + 1];
  __device__ __forceinline__ void init() {
    wt_static_for<NG>([&](auto gg) {
      constexpr int g = decltype(gg)::value, pb = 2 * g + MID, pa = pb + 1;
      wt_static_for<DE + 1>([&](auto ss) {
        constexpr int s_ = decltype(ss)::value;
        constexpr float va = WtPoly<W>::e(pa, DE - s_), vb = WtPoly<W>::e(pb, DE - s_);
        e[g][s_] = (f32x2){va, vb};
#if DDSP_WT_PK_ASM
        DDSP_KEEP_IN_VGPR(e[g][s_]);          // resident: not re-made from literals at every use (a packed FMA takes no literal;
#endif                                        // the plain FMAs do - v_fmaak_f32 -, and the coefficients need no registers)
      });
      wt_static_for<DO + 1>([&](auto ss) {
        constexpr int s_ = decltype(ss)::value;
        constexpr float va = WtPoly<W>::o(pa, DO - s_), vb = WtPoly<W>::o(pb, DO - s_);
        o[g][s_] = (f32x2){va, vb};
#if DDSP_WT_PK_ASM
        DDSP_KEEP_IN_VGPR(o[g][s_]);
#endif
      });
    });
    wt_static_for<DE + 1>([&](auto ss) {
      constexpr int s_ = decltype(ss)::value;
      constexpr float ve = WtPoly<W>::e(0, DE - s_), vo = (DE - s_) <= DO ? WtPoly<W>::o(0, (DE - s_) <= DO ? DE - s_ : 0) : 0.0f;
      m[s_] = (f32x2){ve, vo};
#if DDSP_WT_PK_ASM
      if (MID) DDSP_KEEP_IN_VGPR(m[s_]);
#endif
    });
  }
};

// NT tiles at once (rows j and j + 1 of each); per tile and group of two tap pairs DE + DO packed FMAs for the
// polynomials, 2 for the weights, 4 for the taps
template <int W, int NT>
__device__ __forceinline__ void wt_taps_pk(const float* const (&t)[kWtNT], const f32x2 (&zz)[kWtNT], const WtPkCoefs<W>& c,
                                           f32x2 (&acc0)[kWtNT], f32x2 (&acc1)[kWtNT]) {
  typedef WtPkCoefs<W> C;
  // row j + 1 is 1072 bytes further: past what ds_read2_b32 reaches from row j's address (1020) - ONE second address
  // per tile, made here (left to itself the compiler makes one per read pair)
  const float* t1[kWtNT];
  int row_stride = WtGeom<W>::TS;
  DDSP_KEEP_IN_VGPR(row_stride);               // (opaque: t1 + 4 stays an immediate offset from ONE address)
#pragma unroll
  for (int u = 0; u < NT; ++u) t1[u] = t[u] + row_stride;
  // (every stage over the NT tiles in turn: neighbouring instructions are independent)
  if constexpr (C::MID) {
    f32x2 eo[NT], w[NT], d0[NT], d1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      d0[u] = (f32x2){t[u][0], t[u][1]};
      d1[u] = (f32x2){t1[u][0], t1[u][1]};
      eo[u] = wt_pk_horner(c.m[0], zz[u], c.m[1]);
    }
#pragma unroll
    for (int s_ = 2; s_ <= C::DE; ++s_)
#pragma unroll
      for (int u = 0; u < NT; ++u) eo[u] = wt_pk_horner(eo[u], zz[u], c.m[s_]);
#pragma unroll
    for (int u = 0; u < NT; ++u) w[u] = wt_pk_weights(eo[u], zz[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) acc0[u] = wt_pk_fma(w[u], d0[u], acc0[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) acc1[u] = wt_pk_fma(w[u], d1[u], acc1[u]);
  }
  wt_static_for<C::NG>([&](auto gg) {
    constexpr int g = decltype(gg)::value, pb = 2 * g + C::MID, pa = pb + 1;
    f32x2 ee[NT], oo[NT], wp[NT], wm[NT], lo0[NT], hi0[NT], lo1[NT], hi1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      lo0[u] = (f32x2){t[u][-pa], t[u][-pb]};                 // taps -pa, -pb
      hi0[u] = (f32x2){t[u][1 + pb], t[u][1 + pa]};           // taps 1 + pb, 1 + pa: the other way round than their weights
      lo1[u] = (f32x2){t1[u][-pa], t1[u][-pb]};
      hi1[u] = (f32x2){t1[u][1 + pb], t1[u][1 + pa]};
      ee[u] = wt_pk_horner(c.e[g][0], zz[u], c.e[g][1]);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) oo[u] = wt_pk_horner(c.o[g][0], zz[u], c.o[g][1]);
#pragma unroll
    for (int s_ = 2; s_ <= C::DE; ++s_) {
#pragma unroll
      for (int u = 0; u < NT; ++u) ee[u] = wt_pk_horner(ee[u], zz[u], c.e[g][s_]);
      if (s_ <= C::DO)
#pragma unroll
        for (int u = 0; u < NT; ++u) oo[u] = wt_pk_horner(oo[u], zz[u], c.o[g][s_]);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) wp[u] = wt_pk_plus(oo[u], zz[u], ee[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) wm[u] = wt_pk_minus(oo[u], zz[u], ee[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) acc0[u] = wt_pk_fma(wp[u], lo0[u], acc0[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) acc1[u] = wt_pk_fma(wp[u], lo1[u], acc1[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) acc0[u] = wt_pk_fma_swap(wm[u], hi0[u], acc0[u]);
#pragma unroll
    for (int u = 0; u < NT; ++u) acc1[u] = wt_pk_fma_swap(wm[u], hi1[u], acc1[u]);
  });
}

// (the output pointer: restrict-qualified unless processors.Add rides in the kernel - include/ddsp_amd.h lets add_signal BE the
// output buffer, every element read and written by the same lane, and a restrict-qualified pointer would promise the compiler
// that no such read exists)
typedef float* __restrict__ WtOutRestrict;
template <bool ADD> struct WtOut { typedef WtOutRestrict type; };
template <> struct WtOut<true> { typedef float* type; };

template <int W, int NK, bool ONE_TILE, bool ADD, bool ROWS16>
__global__ __launch_bounds__(1024) void harm_table_kernel(
    const float* __restrict__ amplitudes, const float* __restrict__ hd, const float* __restrict__ f0_all,
    typename WtOut<ADD>::type audio, float* __restrict__ ctl_amp, float* __restrict__ ctl_hd, const float* add_in, TableArgs p) {
  constexpr int kWtH = WtGeom<W>::H, kWtTS = WtGeom<W>::TS, kWtPS = WtPlane<NK>::PS;
  constexpr bool WIDE = NK > 2;                   // 129 .. 200 harmonics: see the tabulators and the row makers
  __shared__ __attribute__((aligned(16))) float tab_all[2][kWtRows * kWtTS];
  __shared__ __attribute__((aligned(16))) _Float16 planes_all[3][4 * kWtRows * kWtPS];   // [hi, lo][parity][row][k']: a_k / psi_hat(k)
  __shared__ ChunkTables t_all[2];
  __shared__ __attribute__((aligned(16))) WtDesc ring[8];
  __shared__ float amp_tab[3][kWtRows + 4];          // the frames' amplitudes (scaled), per plane buffer: tabulator 1 -> all tabulators
  __shared__ int tsel[4];                            // the table size of chunk c in slot c & 3: tabulator 1 -> tabulators, interpolators

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_t = wave < 4;
  const int F = p.F, K = p.K;
  const int K4 = (K + 3) >> 2;                    // lanes of a row with a harmonic to their name (any K <= 128)
  const float kLog10 = 2.302585092994046f;       // tf.math.log(exponent), ddsp/core.py:403
  const int sub = lane >> 5, kq = lane & 31;     // phase A: 32 lanes per row, lane kq owns harmonics 4 kq + 1 .. 4 kq + 4
  const bool live = kq < K4;
  const float4* __restrict__ hd4 = reinterpret_cast<const float4*>(hd);
  const int mi = lane & 15, mg = lane >> 4;      // MFMA fragment coordinates
#ifdef DDSP_WT_TIMELINE
  const bool dbg_on = p.dbg != nullptr && blockIdx.x == 0 && lane == 0;
#define DDSP_WT_STAMP(i) do { if (dbg_on && tick + 2 < 24) p.dbg[(wave * 24 + tick + 2) * 8 + (i)] = clock64(); } while (0)
#else
#define DDSP_WT_STAMP(i) do { } while (0)
#endif

  // ---- the first two chunk descriptors (the walker), then everybody reads ----------------------------------------
  WtWalk walk;
  walk.pos = (int)blockIdx.x * p.frames_per_block;
  walk.pos_first = walk.pos;
  walk.end = min(walk.pos + p.frames_per_block, p.total_frames);
  walk.seg_left = 0; walk.base = 0; walk.rem = 0; walk.b = 0; walk.j = 0;
  if (wave == kWtWalker) {
    const WtDesc d0 = wt_next_chunk(walk, p), d1 = wt_next_chunk(walk, p);
    if (lane == 0) { ring[0] = d0; ring[1] = d1; }
  }
  __syncthreads();
  // descriptors of the chunks in flight at tick tau: dL = chunk tau + 3 (rows fetched), dA = tau + 2 (phase A),
  // dM = tau + 1 (tabulated), dB = tau (phase B)
  WtDesc dL{0, 0, 0, 0}, dA = wt_read_desc(ring, 0), dM{0, 0, 0, 0}, dB{0, 0, 0, 0};
  int pa = 0, pm = 2, pb = 1;                    // plane buffers of dA, dM, dB: chunk c uses buffer c % 3

  // the descriptor of chunk tick + 3: read issued at the top of a tick, taken into scalar registers at its end
  int4 dnext = make_int4(0, 0, 0, 0);
  auto desc_issue = [&](int slot) { dnext = *reinterpret_cast<const int4*>(ring + slot); };
  auto desc_take = [&]() {
    dL.b = __builtin_amdgcn_readfirstlane(dnext.x);
    dL.j0 = __builtin_amdgcn_readfirstlane(dnext.y);
    dL.nfr = __builtin_amdgcn_readfirstlane(dnext.z);
    dL.fresh = __builtin_amdgcn_readfirstlane(dnext.w);
  };

  if (is_t) {
    const int rw = wave;
    // ---- this wavefront's share of the constant factor, in MFMA A-operand layout (harm_table_frags.h) ---------------
    // (four k-steps, WIDE: 128 registers' worth - the share of one parity and row tile is fetched from L2 where it is
    // used, below: 64 KB per tabulator and tick, which a frame of three or more tiles - what such shapes have - hides)
    // K <= 128: the fragments of ONE table size at a time, fetched when a chunk's size differs from the last one's (16 loads
    // for 512 points, 4 for the smaller tables: a note that stays inside one size class never fetches again)
    f16x8 ahi[2][2][WIDE ? 1 : NK], alo[2][2][WIDE ? 1 : NK];
    int frag_T = 0;
    auto fetch_fragments = [&](int T) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const WtFragSet& set = W == 6 ? kWtFragSet6 : kWtFragSet8;
      // (a wave-uniform base and a 32-bit lane offset made HERE: left to itself the compiler keeps a 64-bit address per
      // fragment alive over the whole tick loop - in scratch)
      unsigned l16 = 16u * (unsigned)(tid & 63);
      DDSP_KEEP_IN_VGPR(l16);
      if (T == kWtT) {
        const char* base = reinterpret_cast<const char*>(set.t512.v[rw]);      // [part][parity][tt][ks][lane][4]
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int ks = 0; ks < (WIDE ? 1 : NK); ++ks) {
              const unsigned off = (unsigned)(((par * 2 + tt) * 2 + ks) * 1024);
              ahi[par][tt][ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + off)));
              alo[par][tt][ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + (off + 8192u))));
            }
      } else if (rw < (T >> 6)) {
        const WtFragsSmall& f = T == 256 ? set.t256 : T == 128 ? set.t128 : set.t64;
        const char* base = reinterpret_cast<const char*>(f.v[rw]);              // [part][parity][lane][4]
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          ahi[par][0][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + 1024u * par)));
          alo[par][0][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + (l16 + 1024u * par + 2048u)));
        }
      }
    };
    // wavefront 3 builds the per-frame phase tables (lanes = frames 0 .. 32): f0 of the chunk's frames, issued before the
    // MFMAs of the tick and used after them; `before` = the sum of f0 over the frames of the row before the chunk, carried
    // from chunk to chunk and summed afresh (fp64: exact, so the same bits in any order) at the start of a row segment
    float pf_cur = 0.0f, pf_next = 0.0f, pf_first = 0.0f;
    double before = 0.0;
    auto fetch_f0 = [&](const WtDesc& d) {
      int lane_ = lane;
      DDSP_KEEP_IN_VGPR(lane_);
      const float* __restrict__ f0row = f0_all + (size_t)d.b * F;
      load_issue(pf_cur, f0row + min(d.j0 + lane_, F - 1));
      load_issue(pf_next, f0row + min(d.j0 + lane_ + 1, F - 1));
      load_issue(pf_first, f0row);
    };
    auto row_prefix = [&](const WtDesc& d) -> double {
      int lane = tid & 63;
      DDSP_KEEP_IN_VGPR(lane);
      const float* __restrict__ f0row = f0_all + (size_t)d.b * F;
      double part = 0.0;
      if ((F & 3) == 0 && (((uintptr_t)f0_all) & 15) == 0) {
        const float4* __restrict__ f4 = reinterpret_cast<const float4*>(f0row);
        const int n4 = d.j0 >> 2;
        for (int i = lane; i < n4; i += 256) {             // four loads in flight per pass
          const float4 a = f4[i];
          const float4 b = f4[min(i + 64, n4 - 1)], c = f4[min(i + 128, n4 - 1)], e = f4[min(i + 192, n4 - 1)];
          part += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
          if (i + 64 < n4) part += ((double)b.x + (double)b.y) + ((double)b.z + (double)b.w);
          if (i + 128 < n4) part += ((double)c.x + (double)c.y) + ((double)c.z + (double)c.w);
          if (i + 192 < n4) part += ((double)e.x + (double)e.y) + ((double)e.z + (double)e.w);
        }
        const int jt = (n4 << 2) + lane;
        if (jt < d.j0) part += (double)f0row[jt];
      } else {
        for (int j = lane; j < d.j0; j += 64) part += (double)f0row[j];
      }
      return wave_sum_dpp(part);
    };

    // wavefront 1 makes the amplitudes of chunk tick + 2 (lanes = frames 0 .. nfr: core.exp_sigmoid of the row's
    // amplitude, ddsp/synths.py:120-121): fetched at the top of a tick, scaled after its MFMAs, multiplied into the table
    // by every tabulator a tick later - the row makers' planes hold the normalised distribution only
    // ... and the chunk's TABLE SIZE: the smallest f0 over the frames of the chunk's segment and the frame behind them (one load
    // per lane: a segment has 62 frames), wt_table_size
    // (in the registers wavefront 3 pins its f0 loads in - pf_cur, pf_next: a wavefront is one OR the other, which the register
    // allocator cannot know; names of their own cost two registers across the MFMAs and pushed a fragment into scratch)
    float& pamp = pf_cur;
    float& pseg = pf_next;
    // The size of chunk tick + 3 is made at tick `tick` (three ticks before its phase B, two before its tabulation): everybody
    // else reads it a tick ahead of use, beside the descriptor, and carries it in a scalar register - read where it is used, the
    // LDS round trip sat at the head of every tick of wavefronts that set the tick's length (+ 1.5 us per launch at batch 128).
    // The very first chunk's is made together with the second's in the first tick (pf_first holds its f0: nothing else of
    // this wavefront uses that register).
    auto fetch_amp = [&](const WtDesc& d) {
      int lane_ = lane;
      DDSP_KEEP_IN_VGPR(lane_);
      load_issue(pamp, amplitudes + (size_t)d.b * F + min(d.j0 + lane_, F - 1));
    };
    auto fetch_segment = [&](const WtDesc& d, float& dst) {
      int lane_ = lane;
      DDSP_KEEP_IN_VGPR(lane_);
      uint32_t in_seg;
      (void)fastdiv((uint32_t)d.j0, p.seg_div, in_seg);
      load_issue(dst, f0_all + (size_t)d.b * F + min(d.j0 - (int)in_seg + lane_, F - 1));
    };
    auto fragments_landed = [&]() {
#if defined(__AMDGCN__)
      __asm__ volatile("s_waitcnt vmcnt(0)");
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int ks = 0; ks < (WIDE ? 1 : NK); ++ks)
            __asm__ volatile("" : "+v"(ahi[par][tt][ks]), "+v"(alo[par][tt][ks]));
#endif
    };
    // 512 points are what most launches start with: fetched now, on the off chance, and landed at the end of the first tick, in
    // which a tabulator has nothing to tabulate (fetched when the first chunk's size is known, the block's first table waits for
    // 64 KB from L2: + 1200 clocks per launch, r04u)
    if constexpr (!WIDE) { fetch_fragments(kWtT); frag_T = kWtT; }
    int Tm = kWtT, Tm_ahead = kWtT;                // the table size of the chunk this tick tabulates; the next tick's, read ahead
    for (int tick = -2;; ++tick) {
      DDSP_WT_STAMP(0);
      if (rw == 3 && dM.nfr > 0) {
        if (dM.fresh) before = row_prefix(dM);
        fetch_f0(dM);
      }
      WtDesc dN{0, 0, 0, 0};                        // chunk tick + 3 (tabulator 1 only: the others take it at the end of the tick)
      if (rw == 1 && dA.nfr > 0) fetch_amp(dA);
      if constexpr (!WIDE) {
        if (rw == kWtSizer) {
          dN = wt_read_desc(ring, (tick + 3) & 7);
          if (dN.nfr > 0) fetch_segment(dN, pseg);
          if (tick == -2 && dA.nfr > 0) fetch_segment(dA, pf_first);
        }
      }
#if defined(DDSP_EXP_T_DUMMY_LOADS)     // experiment: vector memory loads of THIS wavefront landing while its MFMAs run
      ddsp_f32x4 exp_dummy = {0.f, 0.f, 0.f, 0.f};
      auto exp_dummy_issue = [&]() {
        const float4* src = reinterpret_cast<const float4*>(&kWtFragSet6.t512.v[rw][0][0][0][0][0][0]) + lane;
#pragma unroll
        for (int i = 0; i < DDSP_EXP_T_DUMMY_LOADS; ++i) load_issue(exp_dummy, src + 64 * i);
      };
#if !defined(DDSP_EXP_T_DUMMY_AFTER)
      exp_dummy_issue();
#endif
#if defined(DDSP_EXP_T_DUMMY_WAIT_BEFORE)     // .. but landed before the first MFMA
      __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(exp_dummy));
#endif
#endif
      // ---------------- table of chunk tick + 1: O and E on the quarter range ---------------------------------------
      if constexpr (!WIDE) {
#if !defined(DDSP_EXP_FORCE_T512)
        Tm_ahead = tsel[(tick + 2) & 3];           // chunk tick + 2, the next tick's: issued here, taken at the end of the tick
        if (tick == -1) Tm = __builtin_amdgcn_readfirstlane(tsel[0]);      // (the first chunk's was made in the tick before: no tick ahead of that)
#endif
        if (dM.nfr > 0 && Tm != frag_T) {
          fetch_fragments(Tm);
          frag_T = Tm;
          // Landed HERE, inside the branch: left to the compiler the wait sits where the fragments are first used - partial
          // vmcnt waits in the middle of the MFMAs, executed every tick, which also wait for the pinned f0 / amplitude loads of
          // tabulators 3 and 1 that the MFMAs are there to hide
          fragments_landed();
        }
      }
#if defined(DDSP_EXP_NO_SMALL_PATH)
      if (false) {
#else
      if (dM.nfr > 0 && !WIDE && Tm != kWtT) {
#endif
        // ---- 256, 128 or 64 points: T / 64 position tiles - tabulator rw has tile rw or none -, one k-step -------------------
        if (rw < (Tm >> 6)) {
          const int half = Tm >> 1;
#pragma unroll
          for (int rt = 0; rt < kWtRowTiles; ++rt) {
            const _Float16* bsrc = planes_all[pm] + (16 * rt + mi) * kWtPS + 8 * mg;
            const float am = amp_tab[pm][16 * rt + mi], am_lo = am * (1.0f / kWtLoScale);
            f32x4 soe[2];
#pragma unroll
            for (int par = 0; par < 2; ++par) {
              const f16x8 bhi = *reinterpret_cast<const f16x8*>(bsrc + (0 * 2 + par) * kWtRows * kWtPS);
              const f16x8 blo = *reinterpret_cast<const f16x8*>(bsrc + (1 * 2 + par) * kWtRows * kWtPS);
              const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
              const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][0][0], bhi, zero, 0, 0, 0);
              f32x4 accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][0][0], blo, zero, 0, 0, 0);
              accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[par][0][0], bhi, accx, 0, 0, 0);
              soe[par] = acc * am + accx * am_lo;
            }
            float* trow = tab_all[(tick + 1) & 1] + (16 * rt + mi) * kWtTS + kWtH;
            const int n0 = 16 * rw + 4 * mg;
            const f32x4 sp = soe[0] + soe[1], sm = soe[0] - soe[1];     // S(n) = O + E, S(T/2-1-n) = O - E
            *reinterpret_cast<f32x4*>(trow + n0) = sp;
            *reinterpret_cast<f32x4*>(trow + (half - 4 - n0)) = (f32x4){sm.w, sm.z, sm.y, sm.x};
            if (n0 == 0) {                                              // halos (four entries either side: K <= 128)
              *reinterpret_cast<f32x4*>(trow - 4) = (f32x4){-sp.w, -sp.z, -sp.y, -sp.x};
              *reinterpret_cast<f32x4*>(trow + half) = (f32x4){-sm.x, -sm.y, -sm.z, -sm.w};
            }
          }
        }
      } else if (dM.nfr > 0) {
#pragma unroll
       for (int rt = 0; rt < kWtRowTiles; ++rt) {
        // B: element e of lane (j = lane & 15, g = lane >> 4): plane[part][par][row 16 rt + j][32 ks + 8 g + e]
        const _Float16* bsrc = planes_all[pm] + (16 * rt + mi) * kWtPS + 8 * mg;
        // one parity at a time (four accumulators live, not eight: the constant factor already takes 64 registers)
        f32x4 soe[2][2];                                     // [parity][position tile]: a_j (hi.hi + (hi.lo + lo.hi) / 2048)
        const float am = amp_tab[pm][16 * rt + mi], am_lo = am * (1.0f / kWtLoScale);      // this lane's table row's amplitude
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          f32x4 acc[2], accx[2];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
          if constexpr (WIDE) {
            // this parity's sixteen fragments, issued together (pinned: left to itself the compiler hoists these loop-
            // invariant loads out of the tick loop, into registers nobody has) and used k-step by k-step as they land
            ddsp_f32x4 fr[NK][2][2];                          // [k-step][position tile][hi, lo]
            // (one scalar base per parity, a lane offset per position tile and part, the k-step as the immediate offset;
            // spaced: the bases come back from spill lanes - common.h)
            const char* fbase = reinterpret_cast<const char*>(kWtFragsWide.v) + 16384 * (rw * 2 + par);
            unsigned lane16 = 16u * (unsigned)lane;
            DDSP_KEEP_IN_VGPR(lane16);
            wt_static_for<NK>([&](auto kk) {
              constexpr int ks = decltype(kk)::value;
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                  load_issue_spaced<1024 * ks>(fr[ks][tt][hl], fbase, lane16 + 4096u * (unsigned)(2 * tt + hl));
            });
            wt_static_for<NK>([&](auto kk) {
              constexpr int ks = decltype(kk)::value;
              const f16x8 bhi = *reinterpret_cast<const f16x8*>(bsrc + (0 * 2 + par) * kWtRows * kWtPS + wt_wide_kstep_base(ks));
              const f16x8 blo = *reinterpret_cast<const f16x8*>(bsrc + (1 * 2 + par) * kWtRows * kWtPS + wt_wide_kstep_base(ks));
              wt_frags_landed<4 * (NK - 1 - ks)>(fr[ks][0][0], fr[ks][0][1], fr[ks][1][0], fr[ks][1][1]);
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr[ks][tt][0]), bhi, acc[tt], 0, 0, 0);
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
                accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr[ks][tt][0]), blo, accx[tt], 0, 0, 0);
#pragma unroll
              for (int tt = 0; tt < 2; ++tt)
                accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr[ks][tt][1]), bhi, accx[tt], 0, 0, 0);
            });
          } else {
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) {
            const f16x8 bhi = *reinterpret_cast<const f16x8*>(bsrc + (0 * 2 + par) * kWtRows * kWtPS + 32 * ks);
            const f16x8 blo = *reinterpret_cast<const f16x8*>(bsrc + (1 * 2 + par) * kWtRows * kWtPS + 32 * ks);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][tt][ks], bhi, acc[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[par][tt][ks], blo, accx[tt], 0, 0, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
              accx[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[par][tt][ks], bhi, accx[tt], 0, 0, 0);
          }
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) soe[par][tt] = acc[tt] * am + accx[tt] * am_lo;
        }
        // D[row = 4 (lane >> 4) + reg][col = lane & 15]: this lane holds positions n0 .. n0+3 of table row mi
        float* trow = tab_all[(tick + 1) & 1] + (16 * rt + mi) * kWtTS + kWtH;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int n0 = 16 * (2 * rw + tt) + 4 * mg;
          const f32x4 so = soe[0][tt], se = soe[1][tt];     // odd, even harmonics
          const f32x4 sp = so + se;                         // S(n)         = O + E
          const f32x4 sm = so - se;                         // S(T/2-1-n)   = O - E
          *reinterpret_cast<f32x4*>(trow + n0) = sp;
          *reinterpret_cast<f32x4*>(trow + (kWtHalf - 4 - n0)) = (f32x4){sm.w, sm.z, sm.y, sm.x};
          if constexpr (kWtH == 4) {
          if (n0 == 0) {                                     // halos: S(-1-m) = -S(m), S(T/2+m) = -S(T/2-1-m)
            *reinterpret_cast<f32x4*>(trow - kWtH) = (f32x4){-sp.w, -sp.z, -sp.y, -sp.x};
            *reinterpret_cast<f32x4*>(trow + kWtHalf) = (f32x4){-sm.x, -sm.y, -sm.z, -sm.w};
          }
          } else {
          if (n0 < kWtH) {                                   // (eight entries either side: the lanes with n0 = 0 and 4)
            *reinterpret_cast<f32x4*>(trow - 4 - n0) = (f32x4){-sp.w, -sp.z, -sp.y, -sp.x};
            *reinterpret_cast<f32x4*>(trow + kWtHalf + n0) = (f32x4){-sm.x, -sm.y, -sm.z, -sm.w};
          }
          }
        }
       }
      }
#if defined(DDSP_EXP_T_DUMMY_LOADS)
#if defined(DDSP_EXP_T_DUMMY_AFTER)           // .. issued behind the last MFMA instead
      exp_dummy_issue();
#endif
      __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(exp_dummy));
#endif
      DDSP_WT_STAMP(1);
      desc_issue((tick + 3) & 7);
      // ---------------- wavefront 3: the per-frame phase tables of chunk tick + 1 ---------------------------------------
      if (rw == 3) {
        if (dM.nfr > 0) {
          loads_landed(pf_cur, pf_next, pf_first);
          // (an opaque lane number and K: what depends on them is made here, once per tick on one wavefront, instead of
          // being hoisted out of the loop into registers the T-wavefronts do not have)
          int lane = tid & 63, Kc = K;
          DDSP_KEEP_IN_VGPR(lane);
#if defined(__AMDGCN__)
          __asm__ volatile("" : "+s"(Kc));
#endif
          ChunkTables& t = t_all[(tick + 1) & 1];
          const int nfr = dM.nfr;
          const float fj = pf_cur, fj1 = pf_next;
          const double fa = (double)fj, fb = (double)fj1;
          const double mine = (lane < nfr) ? fa : 0.0;
          double incl = mine;                                 // inclusive scan over the chunk's frames (lanes 0..31)
          incl += dpp_mov0<0x111, 0xF>(incl);   // row_shr:1
          incl += dpp_mov0<0x112, 0xF>(incl);   // row_shr:2
          incl += dpp_mov0<0x114, 0xF>(incl);   // row_shr:4
          incl += dpp_mov0<0x118, 0xF>(incl);   // row_shr:8
          const long long bits15 = __builtin_bit_cast(long long, incl);
          const unsigned lo15 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits15 & 0xffffffffll), 15);
          const unsigned hi15 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits15 >> 32), 15);
          const double first16 = __builtin_bit_cast(double, (long long)(((unsigned long long)hi15 << 32) | lo15));
          if (lane >= 16) incl += first16;
          const double s_excl = before + (incl - mine);
          const double run = p.hop_d * s_excl + (fa - (double)pf_first) * p.half_hm1;
          const double cyc = run * p.inv_sr;
          const float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
          int kA = Kc, kN = Kc;
          if (fmx > 0.0f) kA = (int)fminf((float)Kc, floorf(p.nyq_lo * __builtin_amdgcn_rcpf(fmx)));
          if (fmn > 0.0f) kN = (int)fminf((float)Kc, floorf(p.nyq_hi * __builtin_amdgcn_rcpf(fmn)));
          kA = max(min(kA, kN), 0);
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
          if (any != 0ull) {                                   // (wave-uniform; rows lane, lane + 1 <= 31 for a frame)
            float4 cx = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float ck = 0.0f;
            if (crossing) {
              const int k = kA;
              const float kfl = (float)(k + 1);
              const _Float16* pl = planes_all[pm] + ((k & 1) * kWtRows + lane) * kWtPS + (k >> 1);
              const float c0 = fmaf((float)pl[2 * kWtRows * kWtPS], 1.0f / kWtLoScale, (float)pl[0]);
              const float c1 = fmaf((float)pl[2 * kWtRows * kWtPS + kWtPS], 1.0f / kWtLoScale, (float)pl[kWtPS]);
              const float top = rn_mul(fj, kfl), bot = rn_mul(fj1, kfl);
              cx = make_float4(rn_mul(c0, amp_tab[pm][lane]), rn_mul(c1, amp_tab[pm][lane + 1]), top, rn_sub(bot, top));
              ck = kfl;
            }
            if (lane < kWtRows) { t.cx[lane] = cx; t.ck[lane] = ck; }
          }
          // the sum over this chunk's frames: lane 31 holds the inclusive sum of lanes 0 .. 31 (lanes >= nfr added 0)
          const long long bits31 = __builtin_bit_cast(long long, incl);
          const unsigned lo31 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits31 & 0xffffffffll), 31);
          const unsigned hi31 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits31 >> 32), 31);
          before += __builtin_bit_cast(double, (long long)(((unsigned long long)hi31 << 32) | lo31));
        }
      }
      if (rw == 1 && dA.nfr > 0) {
        loads_landed(pamp);
        int lane_ = lane;
        DDSP_KEEP_IN_VGPR(lane_);
        const float a = exp_sigmoid_fast(pamp, kLog10, 2.0f, 1e-7f);
        if (lane_ <= kWtRows) amp_tab[pa][lane_] = a;
        if (ctl_amp != nullptr && lane_ < dA.nfr) ctl_amp[(size_t)dA.b * F + dA.j0 + lane_] = a;
      }
      // the table sizes: the tabulator with no other side job (round 4's first version had tabulator 1 do it behind the
      // amplitudes: its tick was the longest of the four - r04u -, and these dozen instructions cost the 70 Hz case half a microsecond)
      if (rw == kWtSizer && (dN.nfr > 0 || (tick == -2 && dA.nfr > 0))) {
        loads_landed(pseg, pf_first);
        int lane_ = lane;
        DDSP_KEEP_IN_VGPR(lane_);
        if constexpr (!WIDE) {
#if !defined(DDSP_EXP_NO_TSEL_COMPUTE)
          // (lanes 0 .. 62: the segment's frames and the one behind; a NaN is no minimum - fminf - and <= 0 keeps 512 points)
          if (dN.nfr > 0) {
            const int t_sel = wt_table_size(wave_min_dpp(lane_ <= kWtSegment ? pseg : __builtin_inff()), p.size_thr);
            if (lane_ == 0) tsel[(tick + 3) & 3] = t_sel;
          }
          if (tick == -2 && dA.nfr > 0) {
            const int t_sel = wt_table_size(wave_min_dpp(lane_ <= kWtSegment ? pf_first : __builtin_inff()), p.size_thr);
            if (lane_ == 0) tsel[0] = t_sel;
          }
#else
          if (lane_ < 4) tsel[lane_] = kWtT;
#endif
        }
      }
      DDSP_WT_STAMP(2);
      // ---------------- the walker: the descriptor of chunk tick + 4 ---------------------------------------------------
      if (rw == kWtWalker) {
        const WtDesc dn = wt_next_chunk(walk, p);
        if (lane == 0) ring[(tick + 4) & 7] = dn;
      }
      desc_take();
      if constexpr (!WIDE) {
        Tm = __builtin_amdgcn_readfirstlane(Tm_ahead);
        if (tick == -2) fragments_landed();
      }
      DDSP_WT_STAMP(3);
      __syncthreads();
      DDSP_WT_STAMP(4);
      dB = dM; dM = dA; dA = dL;
      { const int t3 = pb; pb = pm; pm = pa; pa = t3; }
      if (tick + 1 >= 0 && dB.nfr == 0) break;
    }
  } else {
    const int sw = wave - 4;                       // 0 .. 11
    if (sw >= 8) {
      // =================== row makers (S-wavefronts 8 .. 11): phase A of four row pairs per tick =====================
      // Row pairs ("units": rows 2 u, 2 u + 1 of a chunk, 32 lanes per row, lane kq owns harmonics 4 kq + 1 .. + 4)
      // 4 (sw - 8) .. + 3 of chunk tick + 2, carried through the stages TOGETHER: a row is one long chain of dependent
      // instructions (exp, log, exp, the sum, 1 / sum, the split), and a wavefront issues one instruction per ~8 clocks
      // only if it has independent ones to issue.  The rows come straight from HBM into registers (below).  Rows past the
      // chunk's halo row are fetched (clamped) and worked on like the others: nobody reads their planes.
      // WIDE (129 .. 200 harmonics): all 64 lanes on ONE row, lane hq owns harmonics 4 hq + 1 .. + 4; a row maker takes rows
      // 8 (sw - 8) .. + 7 of the chunk, four at a time in two passes per tick (such shapes have frames of several tiles:
      // the interpolators set the length of a tick)
      constexpr int NU = 4;
      const int u0 = NU * (sw - 8);
      const int hq = WIDE ? lane : kq;
      const bool live_h = WIDE ? lane < K4 : live;
      auto row_of = [&](int i, int pass) -> int { return 2 * u0 + NU * pass + i; };      // (WIDE)
      // per-lane constants: the harmonic numbers (dead lanes, k > K: 0 and a negative Nyquist limit: always masked)
      float kf[4], nyq_u[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool alive = 4 * hq + u + 1 <= K;              // (K need not be a multiple of 4: the last lane's tail is dead)
        kf[u] = alive ? (float)(4 * hq + u + 1) : 0.0f;
        nyq_u[u] = alive ? p.nyquist : -1.0f;
      }
      // the rows of a chunk in registers: fetched at the top of the tick BEFORE the one that works on them (HBM has a whole
      // tick to answer; fetched at the end of a tick, as in r03h, every tick of the block's slowest wavefronts began
      // with the full latency: 41.0 -> 38.8 us), two sets used in turn (the tick loop is unrolled twice: no copies)
      struct Rows { ddsp_f32x4 x[NU]; float f0[NU]; };
      Rows rows_a, rows_b;
      // (addresses: the clip's first row as wave-uniform bases in scalar registers, the rest as 32-bit byte offsets - a
      // clip's F K floats are < 4 GB and F < 2^24, harm_table_ok: four vector instructions per row pair where 64-bit row
      // arithmetic took twelve)
      const unsigned kq16 = 16u * (unsigned)min(hq, K4 - 1), row_bytes = 4u * (unsigned)K;
      // rows that are not 16-byte aligned (K % 4 != 0, or an odd base): four 4-byte loads per lane, the tail clamped
      unsigned ku4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ku4[u] = 4u * (unsigned)min(4 * hq + u, K - 1);
      // (loads ISSUED where they are written and first touched behind rows_landed() a tick later, common.h; the row format
      // is a template parameter: as a run-time branch - plain loads, two formats - it made the compiler's wait counts at
      // the join conservative and the tick began with part of the HBM latency again: 39.2 - 41.4 instead of 37.5 us, r03s)
      auto prefetch = [&](const WtDesc& d, Rows& r, int pass) {
        const size_t r0 = (size_t)d.b * (size_t)F;                               // (an empty descriptor: row 0 of clip 0)
        const char* hb = reinterpret_cast<const char*>(hd) + r0 * row_bytes;
        const char* fb = reinterpret_cast<const char*>(f0_all) + r0 * 4;
        unsigned ro[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          const unsigned jr = (unsigned)min(WIDE ? d.j0 + row_of(i, pass) : d.j0 + 2 * (u0 + i) + sub, F - 1);
          ro[i] = __umul24(jr, row_bytes);                                                            // F < 2^24
          if constexpr (WIDE) load_issue_spaced<0>(r.f0[i], fb, 4u * jr);      // (spaced: common.h)
          else load_issue(r.f0[i], fb, 4u * jr);
        }
        if constexpr (ROWS16) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            if constexpr (WIDE) load_issue_spaced<0>(r.x[i], hb, ro[i] + kq16);
            else load_issue(r.x[i], hb, ro[i] + kq16);
          }
        } else {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            float e0, e1, e2, e3;
            if constexpr (WIDE) {
              load_issue_spaced<0>(e0, hb, ro[i] + ku4[0]);
              load_issue_spaced<0>(e1, hb, ro[i] + ku4[1]);
              load_issue_spaced<0>(e2, hb, ro[i] + ku4[2]);
              load_issue_spaced<0>(e3, hb, ro[i] + ku4[3]);
            } else {
            load_issue(e0, hb, ro[i] + ku4[0]);
            load_issue(e1, hb, ro[i] + ku4[1]);
            load_issue(e2, hb, ro[i] + ku4[2]);
            load_issue(e3, hb, ro[i] + ku4[3]);
            }
            r.x[i] = (ddsp_f32x4){e0, e1, e2, e3};
          }
        }
      };
      auto rows_landed = [&](Rows& r) {
#if defined(__AMDGCN__)
        __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(r.x[0]), "+v"(r.x[1]), "+v"(r.x[2]), "+v"(r.x[3]), "+v"(r.f0[0]), "+v"(r.f0[1]),
                         "+v"(r.f0[2]), "+v"(r.f0[3]));
#else
        (void)r;
#endif
      };
      // core.exp_sigmoid (core.py:386-404), remove_above_nyquist on f0 * [1..K] (core.py:899-903, 1028-1045),
      // safe_divide by the row sum (core.py:905-907, 207-210), amplitudes * distribution (core.py:1097)
      // (harmonics as pairs: v_pk_mul / v_pk_add / v_pk_fma_f32 do two lanes' worth per issued instruction)
      // (harmonics as pairs of EQUAL parity - (4 kq + 1, 4 kq + 3) and (4 kq + 2, 4 kq + 4): v_pk_mul / v_pk_add /
      // v_pk_fma_f32 do two lanes' worth per issued instruction, and a pair is what one dword of a parity plane holds)
      const f32x2 kf_o = {kf[0], kf[2]}, kf_e = {kf[1], kf[3]};
      auto exp_sigmoid2 = [&](f32x2 v) -> f32x2 {                      // exp_sigmoid_fast on a pair
        const f32x2 t = v * -1.4426950408889634f;
        const f32x2 e = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
        const f32x2 m = (f32x2){__builtin_amdgcn_logf(e[0]), __builtin_amdgcn_logf(e[1])} * -kLog10;
        const f32x2 g = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
        return __builtin_elementwise_fma(g, (f32x2){2.0f, 2.0f}, (f32x2){1e-7f, 1e-7f});
      };
      // e > 0: kept iff fl32(f0 k) < nyquist: median(e, 0, (nyquist - fl32(f0 k)) 2^100) - the scaled difference is one
      // FMA of exact products: 0 or >= 2^76 in magnitude, its sign the comparison's
      const float kHuge = 0x1p100f;
      const f32x2 nyq_o = {nyq_u[0] * kHuge, nyq_u[2] * kHuge}, nyq_e = {nyq_u[1] * kHuge, nyq_u[3] * kHuge};
      auto nyq_mask2 = [&](f32x2 e, float f0r, f32x2 kfp, f32x2 nyq_s) -> f32x2 {
        f32x2 prod;
        { _Pragma("clang fp contract(off)") prod = kfp * f0r; }
        const f32x2 y = __builtin_elementwise_fma(prod, (f32x2){-kHuge, -kHuge}, nyq_s);
        return (f32x2){__builtin_amdgcn_fmed3f(e[0], 0.0f, y[0]), __builtin_amdgcn_fmed3f(e[1], 0.0f, y[1])};
      };
      auto phase_a = [&](const WtDesc& d, _Float16* planes, const Rows& r, int pass) {
        const int nfr = d.nfr;
        f32x2 xo[NU], xe[NU];
        float part[NU], inv[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          xo[i] = nyq_mask2(exp_sigmoid2((f32x2){r.x[i][0], r.x[i][2]}), r.f0[i], kf_o, nyq_o);
          xe[i] = nyq_mask2(exp_sigmoid2((f32x2){r.x[i][1], r.x[i][3]}), r.f0[i], kf_e, nyq_e);
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) { const f32x2 h = xo[i] + xe[i]; part[i] = h[0] + h[1]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0xB1, 0xF>(part[i]);      // quad_perm [1,0,3,2]
#pragma unroll
        for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0x4E, 0xF>(part[i]);      // quad_perm [2,3,0,1]
#pragma unroll
        for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0x141, 0xF>(part[i]);     // row_half_mirror
#pragma unroll
        for (int i = 0; i < NU; ++i) part[i] += dpp_mov0<0x140, 0xF>(part[i]);     // row_mirror: every lane holds its 16-lane row's sum
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          part[i] = row_pair_sum(part[i]);            // + the other 16 lanes of this matrix row
          if constexpr (WIDE) part[i] = wave_half_sum(part[i]);      // + the other 32 lanes: the row is the whole wavefront
          // safe_divide: a sum of values that are 0 or >= 1e-7 is 0 (everything masked: eps instead) or >= 1e-7
          inv[i] = __builtin_amdgcn_rcpf(fmaxf(part[i], 1e-7f));
        }
        // the controls dict (return_outputs_dict=True, how dags.py:171-173 calls every processor); the halo row
        // belongs to the next chunk
        if (ctl_hd != nullptr) {
#pragma unroll
          for (int i = 0; i < NU; ++i) {
            const int arow = WIDE ? row_of(i, pass) : 2 * (u0 + i) + sub;
            const int crow = d.b * F + d.j0 + arow;        // this lane's (batch * frame) row, if arow < nfr
            if (arow < nfr) {
              const f32x2 ho = xo[i] * inv[i], he = xe[i] * inv[i];
              if constexpr (ROWS16) {
                if (live_h) reinterpret_cast<float4*>(ctl_hd)[(size_t)crow * K4 + hq] = make_float4(ho[0], he[0], ho[1], he[1]);
              } else {
                float* __restrict__ crow_p = ctl_hd + (size_t)crow * K + 4 * hq;
                const float h4[4] = {ho[0], he[0], ho[1], he[1]};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                  if (4 * hq + u < K) crow_p[u] = h4[u];
              }
            }
          }
        }
        // d_k (the normalised distribution; 1 / psi_hat(k) is in the constant factor since round 4: harm_table_frags.h) as
        // hi + lo / 2048, two fp16 numbers each (hi rounded toward zero by v_cvt_pkrtz_f16_f32: lo takes up the rest)
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          const f32x2 c[2] = {xo[i] * inv[i], xe[i] * inv[i]};       // k odd (k' = 2 kq, 2 kq + 1), k even
          _Float16* dst = planes + (WIDE ? row_of(i, pass) : 2 * (u0 + i) + sub) * kWtPS + 2 * hq;
#pragma unroll
          for (int par = 0; par < 2; ++par) {
            const h16x2 hi = __builtin_amdgcn_cvt_pkrtz(c[par][0], c[par][1]);
            const h16x2 lo = wt_rest_halves(c[par] * kWtLoScale, hi);
            if (!WIDE || 2 * hq < kWtPS) {              // (WIDE: a plane row ends at k' = 104; dead harmonics write zeros up to there)
              *reinterpret_cast<h16x2*>(dst + (0 * 2 + par) * kWtRows * kWtPS) = hi;
              *reinterpret_cast<h16x2*>(dst + (1 * 2 + par) * kWtRows * kWtPS) = lo;
            }
          }
        }
      };
      prefetch(dA, rows_a, 0);
      if constexpr (WIDE) prefetch(dA, rows_b, 1);
      auto one_tick = [&](int tick, Rows& cur, Rows& next) -> bool {
        DDSP_WT_STAMP(0);
        desc_issue((tick + 3) & 7);
        desc_take();
        rows_landed(cur);                          // issued a tick ago
        if constexpr (WIDE) {
          // cur = the first four rows, next = the other four, both fetched during the tick before; each set is
          // fetched again for the next chunk as soon as it has been used
          rows_landed(next);
          DDSP_WT_STAMP(1);
          DDSP_WT_STAMP(2);
          if (dA.nfr > 0) phase_a(dA, planes_all[pa], cur, 0);
          prefetch(dL, cur, 0);
          if (dA.nfr > 0) phase_a(dA, planes_all[pa], next, 1);
          prefetch(dL, next, 1);
        } else {
        prefetch(dL, next, 0);                     // the next tick's dA
        DDSP_WT_STAMP(1);
        DDSP_WT_STAMP(2);
        if (dA.nfr > 0) phase_a(dA, planes_all[pa], cur, 0);
        }
        DDSP_WT_STAMP(3);
        __syncthreads();
        DDSP_WT_STAMP(4);
        dB = dM; dM = dA; dA = dL;
        { const int t3 = pb; pb = pm; pm = pa; pa = t3; }
        return tick + 1 >= 0 && dB.nfr == 0;
      };
      if constexpr (WIDE) {
        for (int tick = -2;; ++tick)
          if (one_tick(tick, rows_a, rows_b)) break;
      } else {
      for (int tick = -2;; tick += 2) {
        if (one_tick(tick, rows_a, rows_b)) break;
        if (one_tick(tick + 1, rows_b, rows_a)) break;
      }
      }
#if defined(__AMDGCN__)
      __asm__ volatile("s_waitcnt vmcnt(0)");      // (nothing in flight when the wavefront leaves: tests/test_isa_guards.py reads
#endif                                             // the instruction stream as text, past the end of this loop)
    } else {
      // =================== interpolators (S-wavefronts 0 .. 7): phase B, four tiles at a time ========================
      // tile slots s, s + 8, s + 16, s + 24 of every round of 31 (slot 7 has three), s = this wavefront's nibble of kWtSlots
      WtPkCoefs<W> coef;
      coef.init();
      const int slot = (int)((kWtSlots >> (4 * sw)) & 7u);
      // the table size of the chunk phase B works on: read a tick ahead (tabulator 1 wrote it two ticks before the chunk's
      // phase B), taken into a scalar register at the end of the tick - read where it is used, its LDS latency was at the head
      // of every tick of the wavefronts that set the tick's length (+ 0.7 us per launch at batch 128)
      float Tf = (float)kWtT;
      int t_ahead = kWtT;
      for (int tick = -2;; ++tick) {
        DDSP_WT_STAMP(0);
        desc_issue((tick + 3) & 7);
#if !defined(DDSP_EXP_FORCE_T512)
        if constexpr (!WIDE) t_ahead = tsel[(tick + 1) & 3];  // chunk tick + 1: the next tick's
#endif
        DDSP_WT_STAMP(1);
        if (dB.nfr > 0) {
          // ---------------- phase B of chunk tick: tiles of 64 samples, lanes = samples ----------------------------
          const int nfr = dB.nfr;
          const int row0 = dB.b * F + dB.j0;
          const float* tab = tab_all[tick & 1];
          const _Float16* planes = planes_all[pb];
          const ChunkTables& t = t_all[tick & 1];
          const int hop = p.hop;
          const float inv_hop = 1.0f / (float)hop;
          const bool chunk_cross = __builtin_amdgcn_readfirstlane(t.cross) != 0;      // one look per tick, not per tile

          const int n_tiles = ONE_TILE ? nfr : nfr * ((hop + 63) >> 6);      // (hop % 64 != 0: the frame's last tile is cut short)
          const bool ragged = !ONE_TILE && p.ragged != 0;
          const size_t chunk0 = (size_t)row0 * (size_t)hop;
          char* out_chunk = reinterpret_cast<char*>(audio + chunk0);          // (add_in may be this very buffer: no __restrict__)
          const char* add_chunk = ADD ? reinterpret_cast<const char*>(add_in + chunk0) : nullptr;
          // NT tiles (tile, tile + 8, ..) move through the stages together: every stage of a tile is a chain of
          // dependent instructions - fp64 phase, LDS reads, the window polynomials
          auto tiles = [&](int tile, auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            int q[kWtNT], r[kWtNT];
            double cyc[kWtNT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              const int tl = tile + 8 * u;
              if (ONE_TILE) { q[u] = tl; r[u] = lane; }
              else {
                uint32_t rem;
                q[u] = (int)fastdiv((uint32_t)tl, p.tpf_div, rem);
                r[u] = (int)rem * 64 + lane;
              }
              const double rr = (double)r[u];
              // inclusive cumsum of f[t]/sr inside the frame: (r+1) w + r (r+1) dw, in revolutions
              // (every product-sum of phase B is spelled out: which of a b + c d becomes the FMA must not depend on the
              // template instance a tile happens to run in - the cut into chunks depends on the batch size)
              cyc[u] = fma(rr + 1.0, fma(t.dw[q[u]], rr, t.w[q[u]]), t.theta[q[u]]);
            }
            // processors.Add fused in (ddsp/processors.py:162-176): the other signal's samples, fetched now, added at the
            // store (add_in may be the output buffer itself: every element is read and written by the same lane)
            // (a tile's samples are elements 64 tile + lane of the chunk, whatever the hop: a wave-uniform base and a
            // 32-bit offset)
            const unsigned o32 = 4u * (unsigned)(tile * 64 + lane);          // bytes
            // (frames that are not whole tiles - hop % 64 != 0, round 4: the reference's own test shapes, 640 frames of 100 samples -:
            // a tile's samples start at q hop + 64 rem of the chunk, and the lanes past the frame's end neither load nor store)
            float addv[kWtNT];
            if (ADD) {
              if (!ragged) {
#pragma unroll
                for (int u = 0; u < NT; ++u) addv[u] = *reinterpret_cast<const float*>(add_chunk + (o32 + 2048u * (unsigned)u));
              } else {
#pragma unroll
                for (int u = 0; u < NT; ++u)
                  addv[u] = r[u] < hop ? *reinterpret_cast<const float*>(add_chunk + 4u * (unsigned)(q[u] * hop + r[u])) : 0.0f;
              }
            }
            float theta[kWtNT];
            f32x2 zz[kWtNT];
            unsigned sgn[kWtNT];
            const float* t0[kWtNT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              theta[u] = (float)__builtin_amdgcn_fract(cyc[u]);               // v_fract_f64: [0, 1]
              const float hm = 0.5f - theta[u];                               // S(1 - theta) = -S(theta): sign bit <=> theta > 1/2
              sgn[u] = __builtin_bit_cast(unsigned, hm) & 0x80000000u;
              const float th = 0.5f - fabsf(hm);                              // [0, 0.5]
              const float pos = fmaf(th, Tf, -0.5f);                          // table coordinate, [-0.5, T / 2 - 0.5]
              const float z = __builtin_amdgcn_fractf(pos) - 0.5f;            // (v_fract_f32: pos - floor(pos), below 1)
              zz[u] = (f32x2){z, z * z};
              t0[u] = tab + q[u] * kWtTS + kWtH + wt_floor_int(pos);          // floor(pos) in [-1, T / 2 - 1]
            }
            f32x2 acc0[kWtNT], acc1[kWtNT];
#pragma unroll
            for (int u = 0; u < kWtNT; ++u) { acc0[u] = (f32x2){0.0f, 0.0f}; acc1[u] = (f32x2){0.0f, 0.0f}; }
            wt_taps_pk<W, NT>(t0, zz, coef, acc0, acc1);
            float out[kWtNT], w_cur[kWtNT], w_next[kWtNT], lerp[kWtNT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              lerp[u] = (float)r[u] * inv_hop;
              // frame-rate -> audio-rate amplitude envelope: weight of frame j+1 is lerp ('linear', core.resample)
              // or the periodic Hann(2 hop)[r] ('window', core.py:696-698)
              w_next[u] = p.amp_linear ? lerp[u] : fmaf(-0.5f, __builtin_amdgcn_cosf(0.5f * lerp[u]), 0.5f);
              w_cur[u] = 1.0f - w_next[u];
              const float s0 = acc0[u][0] + acc0[u][1], s1 = acc1[u][0] + acc1[u][1];
              const float v = fmaf(w_next[u], s1, rn_mul(w_cur[u], s0));
              out[u] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ sgn[u]);
            }
            if (chunk_cross)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              const int kA = __builtin_amdgcn_readfirstlane(t.kA[q[u]]);
              const int kN = __builtin_amdgcn_readfirstlane(t.kN[q[u]]);
              if (kA < kN) {         // harmonics crossing Nyquist inside this frame: audio-rate mask, TF's fp32 op order
                {                    // the first of them from the chunk tables (two broadcast reads; see ChunkTables)
                  const float4 cx = t.cx[q[u]];
                  const float ck = t.ck[q[u]];
                  const float fk = rn_add(cx.z, rn_mul(cx.w, lerp[u]));
                  const float ak = fmaf(w_next[u], cx.y, rn_mul(w_cur[u], cx.x));
                  const float sv = sin_rev(fmaf(theta[u], ck, -rintf(theta[u] * ck)));     // exact fractional part of k theta
                  if (fk >= p.nyquist) out[u] = fmaf(-ak, sv, out[u]);
                }
                if (kA + 1 < kN) {   // more than one: the general form (an f0 that moves by several per cent within a frame)
                  const float fj = t.f0[q[u]], fj1 = t.f0[q[u] + 1];
                  const float am0 = amp_tab[pb][q[u]], am1 = amp_tab[pb][q[u] + 1];
                  for (int k = kA + 1; k < kN; ++k) {
                    const float kfl = (float)(k + 1);
                    const float top = fj * kfl, bot = fj1 * kfl;
                    const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp[u]));
                    const _Float16* pl = planes + ((k & 1) * kWtRows + q[u]) * kWtPS + (k >> 1);
                    const float c0 = fmaf((float)pl[2 * kWtRows * kWtPS], 1.0f / kWtLoScale, (float)pl[0]);
                    const float c1 = fmaf((float)pl[2 * kWtRows * kWtPS + kWtPS], 1.0f / kWtLoScale, (float)pl[kWtPS]);
                    const float ak = fmaf(w_next[u], rn_mul(c1, am1), rn_mul(w_cur[u], rn_mul(c0, am0)));
                    const float sv = sin_rev(fmaf(theta[u], kfl, -rintf(theta[u] * kfl)));     // exact fractional part of k theta
                    if (fk >= p.nyquist) out[u] = fmaf(-ak, sv, out[u]);
                  }
                }
              }
            }
            if (ADD)
#pragma unroll
              for (int u = 0; u < NT; ++u) out[u] += addv[u];
#if defined(DDSP_EXP_DEBUG_DUMP)       // experiment: the per-sample intermediates, eight floats per sample
            if (p.dbg != nullptr) {
              float* dump = reinterpret_cast<float*>(p.dbg);
#pragma unroll
              for (int u = 0; u < NT; ++u) {
                float* d8 = dump + 8 * (chunk0 + (size_t)(tile * 64 + lane) + 512u * (size_t)u);
                d8[0] = theta[u]; d8[1] = zz[u][0]; d8[2] = acc0[u][0]; d8[3] = acc0[u][1];
                d8[4] = acc1[u][0]; d8[5] = acc1[u][1]; d8[6] = w_next[u]; d8[7] = (float)(t0[u] - tab);
              }
            }
#endif
            if (!ragged) {
#pragma unroll
              for (int u = 0; u < NT; ++u) *reinterpret_cast<float*>(out_chunk + (o32 + 2048u * (unsigned)u)) = out[u];          // N == F * hop
            } else {
#pragma unroll
              for (int u = 0; u < NT; ++u)
                if (r[u] < hop) *reinterpret_cast<float*>(out_chunk + 4u * (unsigned)(q[u] * hop + r[u])) = out[u];
            }
          };
          for (int base = 0; base < n_tiles; base += kWtFrames) {
            const int left = min(n_tiles - base, kWtFrames);
            const int cnt = (left - slot + 7) >> 3;           // slot, slot + 8, .. below `left`: 0 .. 4
            if (cnt >= 4) tiles(base + slot, std::integral_constant<int, 4>{});
            else if (cnt == 3) tiles(base + slot, std::integral_constant<int, 3>{});
            else if (cnt == 2) tiles(base + slot, std::integral_constant<int, 2>{});
            else if (cnt == 1) tiles(base + slot, std::integral_constant<int, 1>{});
          }
        }
        DDSP_WT_STAMP(2);
        desc_take();
        if constexpr (!WIDE) Tf = (float)__builtin_amdgcn_readfirstlane(t_ahead);
        DDSP_WT_STAMP(3);
        __syncthreads();
        DDSP_WT_STAMP(4);
        dB = dM; dM = dA; dA = dL;
        { const int t3 = pb; pb = pm; pm = pa; pa = t3; }
        if (tick + 1 >= 0 && dB.nfr == 0) break;
      }
    }
  }
#undef DDSP_WT_STAMP
}

bool harm_table_ok(int F, int K, int N, const void* hd, const void* ctl_amp, const void* ctl_hd, unsigned flags,
                   int inputs_are_controls) {
  if (flags & DDSP_HARM_DIRECT_SUM) return false;
  if (!(flags & DDSP_HARM_SCALE_EXP_SIGMOID) || !(flags & DDSP_HARM_NORMALIZE_NYQUIST)) return false;
  if (inputs_are_controls || (ctl_amp == nullptr) != (ctl_hd == nullptr) || (flags >> 24) != 0) return false;
  (void)hd;
  // (any frame size since round 4: frames that are not whole tiles of 64 samples run with their last tile cut short)
  return (N % F) == 0 && K >= 1 && K <= 200 && F < (1 << 24) && (long long)(N / F) * 32 < (1ll << 29);
}

// The fragment sets of a window on the current device: made on the host once per process, copied once per device (under a
// lock; the copy is synchronous - the one thing in this library that is, once).  0 on success.
static int wt_upload_fragments(int W) {
  constexpr int kMaxDevices = 16;
  static std::mutex lock;
  static bool done[kMaxDevices][3] = {};
  static WtFragSet* host6 = nullptr;
  static WtFragSet* host8 = nullptr;
  static WtFragsWide* host10 = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 1;
  const int slot = W == 6 ? 0 : W == 8 ? 1 : 2;
  std::lock_guard<std::mutex> guard(lock);
  if (done[dev][slot]) return 0;
  hipError_t rc = hipSuccess;
  if (W == 6) {
    if (!host6) { host6 = new WtFragSet; wt_fill_frag_set(host6, kWtInvPsi6_T512, (int)(sizeof(kWtInvPsi6_T512) / sizeof(float))); }
    rc = hipMemcpyToSymbol(HIP_SYMBOL(kWtFragSet6), host6, sizeof(WtFragSet));
  } else if (W == 8) {
    if (!host8) { host8 = new WtFragSet; wt_fill_frag_set(host8, kWtInvPsi8_T512, (int)(sizeof(kWtInvPsi8_T512) / sizeof(float))); }
    rc = hipMemcpyToSymbol(HIP_SYMBOL(kWtFragSet8), host8, sizeof(WtFragSet));
  } else {
    if (!host10) { host10 = new WtFragsWide; wt_fill_frags_wide(host10, kWtInvPsi10_T512, (int)(sizeof(kWtInvPsi10_T512) / sizeof(float))); }
    rc = hipMemcpyToSymbol(HIP_SYMBOL(kWtFragsWide), host10, sizeof(WtFragsWide));
  }
  if (rc != hipSuccess) return 1;
  done[dev][slot] = true;
  return 0;
}

int harm_table_prepare(int K) { return wt_upload_fragments(K <= 100 ? 6 : K <= 128 ? 8 : 10); }

int launch_harm_table(const float* amplitudes, const float* hd, const float* f0, float* audio, float* ctl_amp,
                     float* ctl_hd, const float* add_in, int B, int F, int K, int N, int sample_rate, unsigned flags,
                     hipStream_t st) {
  if ((long long)B * F >= (1ll << 31)) return DDSP_ERR_UNSUPPORTED;        // (2^31 frames: no HBM holds their controls)
  TableArgs p;
  p.B = B; p.F = F; p.K = K; p.N = N; p.hop = N / F;
  p.total_frames = B * F;
  p.f_div = make_fastdiv((uint32_t)F);
  p.tpf_div = make_fastdiv((uint32_t)((p.hop + 63) >> 6));
  p.ragged = (p.hop & 63) != 0 ? 1 : 0;
  p.seg_div = make_fastdiv((uint32_t)kWtSegment);
  if (wt_upload_fragments(K <= 100 ? 6 : K <= 128 ? 8 : 10) != 0) return DDSP_ERR_LAUNCH;
  p.nyquist = (float)(sample_rate / 2.0);
  p.nyq_lo = p.nyquist * (1.0f - 4e-6f);
  p.nyq_hi = p.nyquist * (1.0f + 4e-6f);
  p.size_thr = wt_size_thresholds(K <= 100 ? 100 : K <= 128 ? 128 : 200, K, p.nyq_hi);
  p.amp_linear = (flags & DDSP_HARM_AMP_LINEAR) ? 1 : 0;
  p.rows16 = ((K & 3) == 0 && (((uintptr_t)hd | (uintptr_t)ctl_hd) & 15) == 0) ? 1 : 0;
  p.inv_sr = 1.0 / (double)sample_rate;
  p.inv_2hop = 0.5 / (double)p.hop;
  p.hop_d = (double)p.hop;
  p.half_hm1 = ((double)p.hop - 1.0) * 0.5;
  // persistent grid: one block of 16 wavefronts per CU, each with a contiguous run of frames
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  int blocks = (p.total_frames + kWtFrames - 1) / kWtFrames;
  if (blocks > n_cu) blocks = n_cu;
  p.frames_per_block = (p.total_frames + blocks - 1) / blocks;
  blocks = (p.total_frames + p.frames_per_block - 1) / p.frames_per_block;       // no empty block
  const dim3 grid((unsigned)blocks), block(1024);
  p.dbg = nullptr;
#if defined(DDSP_EXP_DEBUG_DUMP)
  if (const char* e = getenv("DDSP_EXP_DUMP_PTR")) p.dbg = reinterpret_cast<long long*>(strtoull(e, nullptr, 10));
#endif
#ifdef DDSP_WT_TIMELINE
  // DDSP_EXP_TABLE_TIMELINE=1 (a -DDDSP_WT_TIMELINE build): block 0 records shader-clock stamps per tick
  static const bool timeline = getenv("DDSP_EXP_TABLE_TIMELINE") != nullptr;
  static long long* dbg_buf = nullptr;
  if (timeline) {
    if (!dbg_buf && hipMalloc(&dbg_buf, 16 * 24 * 8 * sizeof(long long)) != hipSuccess) dbg_buf = nullptr;
    if (dbg_buf) (void)hipMemsetAsync(dbg_buf, 0, 16 * 24 * 8 * sizeof(long long), st);
    p.dbg = dbg_buf;
  }
#endif
  hipEvent_t ev0, ev1;
  profile_kernel_events(kHarmTable, &ev0, &ev1);
#define DDSP_LAUNCH_TABLE__(W, NK, ONE, ADD, R16)                                                                 \
  hipExtLaunchKernelGGL((harm_table_kernel<W, NK, ONE, ADD, R16>), grid, block, 0, st, ev0, ev1, 0, amplitudes, hd, f0, \
                        audio, ctl_amp, ctl_hd, add_in, p)
#define DDSP_LAUNCH_TABLE_(W, NK, ONE, ADD)                                                                     \
  do {                                                                                                         \
    if (p.rows16) DDSP_LAUNCH_TABLE__(W, NK, ONE, ADD, true);                                                  \
    else DDSP_LAUNCH_TABLE__(W, NK, ONE, ADD, false);                                                          \
  } while (0)
#define DDSP_LAUNCH_TABLE(W, NK)                                                                                \
  do {                                                                                                         \
    if (p.hop == 64) {                                                                                         \
      if (add_in != nullptr) DDSP_LAUNCH_TABLE_(W, NK, true, true);                                            \
      else DDSP_LAUNCH_TABLE_(W, NK, true, false);                                                             \
    } else {                                                                                                   \
      if (add_in != nullptr) DDSP_LAUNCH_TABLE_(W, NK, false, true);                                           \
      else DDSP_LAUNCH_TABLE_(W, NK, false, false);                                                            \
    }                                                                                                          \
  } while (0)
  // the 6-tap window holds its 6.3e-6 up to K = 100 (T / 2K >= 2.56); denser spectra take 8 taps
  if (K <= 64) DDSP_LAUNCH_TABLE(6, 1);
  else if (K <= 100) DDSP_LAUNCH_TABLE(6, 2);
  else if (K <= 128) DDSP_LAUNCH_TABLE(8, 2);
  else DDSP_LAUNCH_TABLE(10, 4);        // 129 .. 200 harmonics (config 5's 48 kHz shapes): ten taps on the same 512 points, <= 5.3e-6
#undef DDSP_LAUNCH_TABLE
#undef DDSP_LAUNCH_TABLE_
#undef DDSP_LAUNCH_TABLE__
#ifdef DDSP_WT_TIMELINE
  if (p.dbg) {
    static long long host[16 * 24 * 8];
    if (hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(host, p.dbg, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess) {
      const long long t0 = host[0];
      for (int w = 0; w < 16; ++w)
        for (int i = 0; i < 24 && host[(w * 24 + i) * 8] != 0; ++i) {
          const long long* r = host + (w * 24 + i) * 8;
          if (w < 4)
            fprintf(stderr, "[timeline] T%d tick %3d  start %8lld  mfma+table +%6lld  tables/desc +%6lld  barrier +%6lld\n", w, i - 2,
                    r[0] - t0, r[1] - r[0], r[3] - r[1], r[4] - r[3]);
          else
            fprintf(stderr, "[timeline] S%-2d tick %3d  start %8lld  fetch +%6lld  phase B +%6lld  phase A +%6lld  barrier +%6lld\n", w - 4,
                    i - 2, r[0] - t0, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3]);
        }
    }
  }
#endif
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

}  // namespace ddsp
