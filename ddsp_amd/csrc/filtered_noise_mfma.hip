// FilteredNoise.__call__ (ddsp/synths.py:165-196) for the canonical filter (65 bands -> 128 taps, full Hann window)
// with BOTH of its contractions on the gfx950 matrix cores: the cosine transform of the IR design
// (core.frequency_impulse_response, ddsp/core.py:1534-1565) and the time-varying FIR that core.fft_convolve
// (ddsp/core.py:1382-1473) is algebraically equal to.
//
// The FIR as a matrix product.  core.fft_convolve frames the audio (frame = 64 samples here), convolves frame f with
// ITS taps h_f and overlap-adds; in the time domain
//     z[m] = sum_i x[i] h_{frame(i)}[m - i],          out[n] = z[n + start]     (crop_and_compensate_delay, :1338-1379).
// Take a PAIR of frames (f, f+1) whose first sample is Z: they only reach the 256 outputs z[Z .. Z+255] (64 + 64 + 127).
// Write an output index as Z + 16 a + b (a = 0..15 the column, b = 0..15 the row), a sample of frame f as
// j = 16 p + b - d (p = 0..4, d = 0..15; 16 p - d runs over -15..64 once, samples outside 0..63 are zero).  Then
//     C[b][a] = sum_{p,d} X_f[b][(p,d)] H_f[(p,d)][a],    X_f[b][(p,d)] = x_f[16 p + b - d],   H_f[(p,d)][a] = h_f[16 (a - p) + d]
// (h outside 0..127 is zero), and the second frame of the pair adds the same with a -> a - 4.  That is a
// [16 x 160] . [160 x 16] product per pair: five v_mfma_f32_16x16x32_f16 steps.  X is Toeplitz in the noise, H is dense
// in the taps; a lane's 8 consecutive k values are 8 consecutive samples (descending: the noise is stored reversed) and
// 8 consecutive taps.  fp32 results come from fp16 matrix cores the way harmonic_table.hip gets them: both factors are
// split hi + lo / 2048 and hi.hi, hi.lo + lo.hi are accumulated in fp32 (3 MFMAs per step, 15 per pair = 7.5 per
// frame against 128 v_fma per lane and frame on the vector ALUs).  Successive pairs overlap by 128 outputs: the right
// half of a pair's tile is shifted 8 columns (DPP row_shl:8) into the accumulator of the next pair, the left half is
// complete and is stored.
//
// Alignment.  The Toeplitz operand wants 16 bytes of fp16 from an address that moves by ONE element (2 bytes) from row
// to row.  gfx950 reads such a thing correctly (ds_read_b128 at 2-byte alignment) but ten times slower than an aligned
// one; at 4-byte alignment two ds_read2_b32 cost 1.7x (profiles/r02a_microbench_lds_unaligned.txt).  The reversed noise
// is therefore kept twice, the second copy shifted by one element, and a row reads the copy its parity selects.
//
// One tile = 30 output frames (1920 samples) of one batch row = 32 staged frames (the first two are history).  Persistent
// blocks of 8 wavefronts, two per CU, walk their tiles with two LDS buffers and one barrier per tile (see the kernel;
// the numbers in the comments below are for the 16-wavefront build, halve them):
//   wavefronts 0-3 make the NEXT tile: the 32 x 65 magnitudes straight from HBM into B-fragments (no LDS staging),
//      exp_sigmoid, hi / lo split, 6 MFMAs each against the constant cosine fragments (fp16 hi / lo pairs made at compile
//      time), window, hi / lo split of the taps into the LDS tap table; and the Philox noise tile (reversed, hi / lo
//      split, two copies);
//   wavefronts 4-7 run the FIR of the CURRENT tile: four pairs each, fully unrolled (every LDS address is a per-lane
//      base plus an immediate), 128 samples stored per pair; the 128 outputs that straddle two wavefronts are finished
//      a tile later from a right half left in LDS.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>
#include "../../include/ddsp_amd.h"
#include "common.h"
#include "noise_ir65.h"
#include "profile.h"
#include "filtered_noise_mfma.h"

namespace ddsp {

// A tile: kMfRows staged frames = kMfRows / 2 pairs; the first pair is only history (the 128 taps reach 127 samples back),
// the others are output.
// Block size.  8 wavefronts (4 producers + 4 FIR) on tiles of 30 output frames, TWO blocks per CU: each block has its own
// barrier, so one block's wait at it is the other block's time (13.7 / 39.1 us at batch 32 / 128).  The two blocks of a
// CU do not advance together - the arbiter prefers the older block's wavefronts: its ticks take 2.8 us, the younger
// block's 4.5-6 us until the older one is done (DDSP_MF_DBG_BLOCK picks the block the timeline shows).  16 wavefronts on tiles of
// 62 frames, one block per CU (-DDDSP_MF_WAVES=16): 3 % less history to recompute, but every tick ends with sixteen
// wavefronts waiting for the slowest (15.2 / 40.7 us; profiles/r02_final2_noise_mfma_two_blocks_per_cu.txt).
#ifndef DDSP_MF_WAVES
#define DDSP_MF_WAVES 8
#endif
constexpr int kMfWaves = DDSP_MF_WAVES;
constexpr int kMfPW = kMfWaves / 2;            // producer wavefronts = FIR wavefronts
constexpr int kMfRows = 8 * kMfPW;             // staged frames: 2 of history + the tile's (a producer row group = 16 rows
                                               // x 2 tap tiles; an FIR wavefront = 4 pairs = 8 frames): 32
constexpr int kMfFrames = kMfRows - 2;         // output frames per tile: 30
constexpr int kMfTile = kMfFrames * 64;        // 1920 output samples per tile
constexpr float kMfLoScale = 2048.0f;          // x = hi + lo / 2048 in two fp16 numbers
// tap table: a hi plane and a lo plane; per row 16 groups of 8 taps (16 bytes) + one group of zeros that lanes outside the
// filter's support read.  The 16 lanes of a ds_read_b128 pass read 16 DIFFERENT groups of one row (or the zero group):
// with 16-byte groups they cover all 64 banks once (32-byte {hi, lo} groups were a 2-way conflict, and the LDS is what
// bounds the FIR phase: profiles/r02g_noise_mfma_v5_pipelined_fir.txt)
constexpr int kMfTapRowBytes = 17 * 16;        // 272
constexpr int kMfTapPlane = kMfRows * kMfTapRowBytes;          // 17408: hi plane, then lo plane
// noise: reversed frames, sample j of staged frame s at element 16 + 80 s + 63 - j; the 16 elements between frames
// stay zero (the Toeplitz blocks run over both ends of a frame).  Stored as dwords of two elements, hi and lo parts in
// separate planes: copy E holds elements (2k, 2k+1) in dword k, copy O holds elements (2k-1, 2k).  A fragment is then
// four consecutive dwords of one plane at a 4-byte aligned address: two ds_read2_b32 into four consecutive registers.
constexpr int kMfXStride = 80;
constexpr int kMfXElems = 16 + kMfRows * kMfXStride + 16;     // 5152
constexpr int kMfXPairs = kMfXElems / 2 + 1;                   // 2577 (copy O needs one more)
// bytes of one plane: 10336 = 2584 dwords, so that copy O starts 5168 = 16 (mod 32) dwords after copy E: the E lanes
// (odd rows) and the O lanes (even rows) of one ds_read2_b32 pass then sit in different halves of the banks
constexpr int kMfXPlane = kMfWaves == 16 ? 10336 : 5216;
static_assert(kMfXPlane >= kMfXPairs * 4 && kMfXPlane % 16 == 0 && (2 * kMfXPlane / 4) % 32 == 16, "noise plane layout");
// Generated noise has no lo planes (round 4): copy O follows copy E's hi plane at kMfXOGen bytes - the same 16 (mod 32) dwords -,
// and what the lo planes held is where the scaled magnitudes travel from the noise makers to the designers (MfHandoff).
constexpr int kMfXOGen = kMfXPlane + 96;
static_assert(kMfXOGen % 16 == 0 && (kMfXOGen / 4) % 32 == 16 && kMfWaves == 8, "noise plane layout, generated noise");
template <bool GEN> struct MfX {
  static constexpr int O = GEN ? kMfXOGen : 2 * kMfXPlane;                // byte offset of copy O's hi plane
  static constexpr int BYTES = GEN ? kMfXOGen + kMfXPlane : 4 * kMfXPlane;
};
// The scaled magnitudes of a tile as the designers' MFMA B-fragments (rows 16 rg + i, bins 16 g .. + 15 split even / odd, hi / lo),
// bin 64 and the finished tap 32 of every row: written by the noise makers a tile ahead, read by the designers.
struct MfHandoff {
  uint4 frag[kMfPW / 2][4][64];        // [row group][even hi, even lo, odd hi, odd lo][lane]
  float last[kMfRows];                 // bin 64, scaled (0 for rows outside the clip)
  float tap32[kMfRows];                // window[32] * sum over even bins of cos(pi m / 2) m (taps 32 and 96)
};

typedef _Float16 mf_f16x8 __attribute__((ext_vector_type(8)));
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 mf_h16x2 __attribute__((ext_vector_type(2)));

struct MfArgs {
  int N, F, start, scale, fs;       // fs = frame size (64: one tap row per staged frame; 64 c: c staged frames per row)
  float inv_fs;
  float bias;
  uint32_t k0, k1;
  uint64_t batch_offset;
  int tiles_per_row, n_tiles;       // tiles of 62 frames per batch row; B * tiles_per_row
  FastDiv whole_per_row, fs_div;    // max(tiles_per_row - 1, 1); fs
  int n_whole;                      // B * (tiles_per_row - 1): the tiles before the rows' last ones
  int dbg_block;                    // the block whose times are stamped (DDSP_MF_DBG_BLOCK, default 0)
  int dbg_wave;                     // the FIR wavefront (8 .. 15) whose times are stamped; the producer is dbg_wave - 8
  long long* dbg;                   // block 0's per-tick stamps [16 ticks][3 roles][begin, end] (tools/exp_noise_fir.py), or null
  // The hi / lo instances (GEN_NOISE = false) serve two callers.  Noise the caller SUPPLIES comes at any scale: x_absmax[row] =
  // max |noise[row]| (row_absmax_kernel, a launch ahead of this one) gives the power of two that brings the row to [1/2, 1)
  // before the fp16 split and the outputs back (the FIR is linear in the noise).  And DDSP_NOISE_BITS_23: x == null,
  // the noise is generated on chip with 23-bit samples (common.h), which need the hi / lo pair as supplied noise does.
  const float* x_absmax;
  int bits23;
};

struct __attribute__((packed, aligned(4))) MfU4f { float x, y, z, w; };        // 16 bytes from a 4-byte aligned address
struct __attribute__((packed, aligned(4))) MfU4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t mf_pack(_Float16 a, _Float16 b) {
  return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}
__device__ __forceinline__ void mf_split(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * kMfLoScale);
}
__device__ __forceinline__ void mf_split8(const float (&v)[8], mf_f16x8& hi, mf_f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 h, l;
    mf_split(v[e], h, l);
    hi[e] = h;
    lo[e] = l;
  }
}
__device__ __forceinline__ mf_f16x8 mf_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(mf_f16x8, v);
}

static __constant__ Ir65Frags kIr65Frags = make_ir65_frags();

// Four taps that sit next to each other in a row of the tap table, v[0] at tap t0 (t0 a multiple of 4): split and
// stored as two 8-byte pieces (hi part, lo part).
__device__ __forceinline__ void mf_put4(unsigned char* hrow, int t0, float v0, float v1, float v2, float v3) {
  _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
  mf_split(v0, h0, l0); mf_split(v1, h1, l1); mf_split(v2, h2, l2); mf_split(v3, h3, l3);
  unsigned char* q = hrow + t0 * 2;
  *reinterpret_cast<uint2*>(q) = make_uint2(mf_pack(h0, h1), mf_pack(h2, h3));
  *reinterpret_cast<uint2*>(q + kMfTapPlane) = make_uint2(mf_pack(l0, l1), mf_pack(l2, l3));
}
// The mirror image: v[0] at tap t1, v[1] at t1 - 1, ... v[3] at t1 - 3 (t1 a multiple of 4; t1 itself is skipped when
// `first` is false).  t1 - 1, t1 - 2 share a dword; t1 - 3 and t1 are single halves.
__device__ __forceinline__ void mf_put4_down(unsigned char* hrow, int t1, bool first, float v0, float v1, float v2, float v3) {
  _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
  mf_split(v0, h0, l0); mf_split(v1, h1, l1); mf_split(v2, h2, l2); mf_split(v3, h3, l3);
  unsigned char* q = hrow + (t1 - 3) * 2;                            // taps t1-3 (odd element), then the dword (t1-2, t1-1)
  *reinterpret_cast<uint16_t*>(q) = __builtin_bit_cast(uint16_t, h3);
  *reinterpret_cast<uint16_t*>(q + kMfTapPlane) = __builtin_bit_cast(uint16_t, l3);
  *reinterpret_cast<uint32_t*>(q + 2) = mf_pack(h2, h1);
  *reinterpret_cast<uint32_t*>(q + kMfTapPlane + 2) = mf_pack(l2, l1);
  if (first) {
    *reinterpret_cast<uint16_t*>(q + 6) = __builtin_bit_cast(uint16_t, h0);
    *reinterpret_cast<uint16_t*>(q + kMfTapPlane + 6) = __builtin_bit_cast(uint16_t, l0);
  }
}

// One quad of the noise tile into the LDS planes: samples v.x .. v.w of staged frame s at j .. j + 3 (reversed: element u0 =
// sample j + 3).  LO: also the lo parts (supplied noise; generated noise IS fp16: its lo planes are never written nor read).
template <bool LO>
__device__ __forceinline__ void mf_put_quad(float4 v, int qd, unsigned char* s_xe, unsigned char* s_xo) {
  _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
  mf_split(v.w, h0, l0);               // element u0     = sample j + 3
  mf_split(v.z, h1, l1);               // element u0 + 1 = sample j + 2
  mf_split(v.y, h2, l2);
  mf_split(v.x, h3, l3);
  const int s = qd >> 4, j = 4 * (qd & 15);
  const int u0 = 16 + kMfXStride * s + 60 - j;                     // a multiple of 4
  // copy E: dwords u0/2 and u0/2 + 1 of each plane (8 bytes, 8-byte aligned)
  *reinterpret_cast<uint2*>(s_xe + u0 * 2) = make_uint2(mf_pack(h0, h1), mf_pack(h2, h3));
  if (LO) *reinterpret_cast<uint2*>(s_xe + kMfXPlane + u0 * 2) = make_uint2(mf_pack(l0, l1), mf_pack(l2, l3));
  // copy O: element e is half (e + 1) & 1 of dword (e + 1) >> 1, i.e. at byte 2 (e + 1): u0 -> high half of
  // dword u0/2, (u0+1, u0+2) -> dword u0/2 + 1, u0+3 -> low half of dword u0/2 + 2
  unsigned char* po = s_xo + (u0 + 1) * 2;
  *reinterpret_cast<uint16_t*>(po) = __builtin_bit_cast(uint16_t, h0);
  *reinterpret_cast<uint32_t*>(po + 2) = mf_pack(h1, h2);
  *reinterpret_cast<uint16_t*>(po + 6) = __builtin_bit_cast(uint16_t, h3);
  if (LO) {
    *reinterpret_cast<uint16_t*>(po + kMfXPlane) = __builtin_bit_cast(uint16_t, l0);
    *reinterpret_cast<uint32_t*>(po + kMfXPlane + 2) = mf_pack(l1, l2);
    *reinterpret_cast<uint16_t*>(po + kMfXPlane + 6) = __builtin_bit_cast(uint16_t, l3);
  }
}

// The noise tile x[z0-128 .. z0+128+kMfTile) into the LDS planes (reversed, two copies): NLANES lanes (ltid).  Supplied noise:
// 16 kMfRows / NLANES quads of four samples per lane, hi / lo split.  Generated noise (common.h: 2048 levels, every value an fp16
// number): 8 kMfRows / NLANES octets per lane - a Philox block is eight samples -, hi planes only.
template <bool GEN_NOISE, int NLANES>
__device__ __forceinline__ void mf_noise_tile(int ltid, int b, int z0, const float* __restrict__ x, unsigned char* s_xe,
                                              unsigned char* s_xo, const MfArgs& p) {
  // supplied noise: the exponent of the row's largest magnitude (a scalar load: the row is wave-uniform)
  int x_exp = 0;
  if constexpr (!GEN_NOISE) {
    if (p.x_absmax) x_exp = pow2_exponent(p.x_absmax[b]);
  }
  if constexpr (GEN_NOISE) {
    static_assert((8 * kMfRows) % NLANES == 0, "octets per lane");
#pragma unroll
    for (int h = 0; h < 8 * kMfRows / NLANES; ++h) {
      const int oc = ltid + NLANES * h;
      const int i = z0 - 128 + 8 * oc;                              // a multiple of 8 (z0 is a multiple of 64)
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (i >= 0 && i < p.N) {                                       // (i < 0: i <= -8, the whole octet lies before the clip)
        const U4 r = noise_philox(U4{(uint32_t)(i >> 3), (uint32_t)(p.batch_offset + b), 0u, 0u}, p.k0, p.k1);
        v0 = noise_quad(r, 0);
        v1 = noise_quad(r, 1);
        if (i + 7 >= p.N) {                                          // the clip's last octet: zeros behind its end
          if (i + 1 >= p.N) v0.y = 0.f;
          if (i + 2 >= p.N) v0.z = 0.f;
          if (i + 3 >= p.N) v0.w = 0.f;
          if (i + 4 >= p.N) v1.x = 0.f;
          if (i + 5 >= p.N) v1.y = 0.f;
          if (i + 6 >= p.N) v1.z = 0.f;
          v1.w = 0.f;
        }
      }
      mf_put_quad<false>(v0, 2 * oc, s_xe, s_xo);
      mf_put_quad<false>(v1, 2 * oc + 1, s_xe, s_xo);
    }
  } else {
    static_assert((16 * kMfRows) % NLANES == 0, "quads per lane");
#pragma unroll
    for (int h = 0; h < 16 * kMfRows / NLANES; ++h) {
      const int qd = ltid + NLANES * h;
      const int i = z0 - 128 + 4 * qd;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i >= 0 && i < p.N) {
        if (x == nullptr) {                                          // (launch-uniform) 23-bit samples made here
          v = noise_quad_at((uint32_t)i, p.batch_offset + b, p.k0, p.k1, true);
          if (i + 1 >= p.N) v.y = 0.f;
          if (i + 2 >= p.N) v.z = 0.f;
          if (i + 3 >= p.N) v.w = 0.f;
        } else {
          const float* src = x + (size_t)b * p.N + i;
          if (i + 3 < p.N && ((p.N & 3) == 0)) {
            v = *reinterpret_cast<const float4*>(src);
          } else {
            v.x = src[0];
            if (i + 1 < p.N) v.y = src[1];
            if (i + 2 < p.N) v.z = src[2];
            if (i + 3 < p.N) v.w = src[3];
          }
          if (x_exp) { v.x = ldexpf(v.x, -x_exp); v.y = ldexpf(v.y, -x_exp); v.z = ldexpf(v.z, -x_exp); v.w = ldexpf(v.w, -x_exp); }
        }
      }
      mf_put_quad<true>(v, qd, s_xe, s_xo);
    }
  }
}

// FS64: frames of exactly 64 samples (the canonical hop): staged frame s uses tap row s.
//
// Persistent blocks of kMfWaves wavefronts (two blocks of 8 per CU), tiles dealt round-robin; two LDS buffers; one barrier
// per tick (with P = kMfPW producers):
//     tick k:   producer wavefronts 0 .. P-1 fill buffer (k+1) & 1 with tile k+1 (vector ALUs): wavefront w designs the
//               taps n = 16 (w / (P/2)) .. + 15 of rows 16 (w % (P/2)) .. + 15 and generates its share of the noise tile;
//               FIR wavefronts P .. 2P-1 turn buffer k & 1 (tile k) into audio (matrix cores + LDS reads), two per SIMD
//               (of the CU's two blocks together) so that one's LDS latency hides behind the other's MFMAs (with one per
//               SIMD a pair took 1400 clocks, profiles/r02e_noise_mfma_v3_persistent_12waves.txt).
// The producers fetch the magnitudes of tile k+2 at the top of the tick and use them a tick later, so no wavefront
// ever waits for HBM.
template <bool GEN_NOISE, bool FS64>
__global__ __launch_bounds__(64 * kMfWaves, 4) void noise_mfma65_kernel(
    const float* __restrict__ mag /*[B,F,65]*/, const float* __restrict__ x /*[B,N] or null*/,
    float* __restrict__ ctl_out /*[B,F,65] or null*/, float* __restrict__ out /*[B,N]*/, MfArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char s_taps_all[2][2 * kMfTapPlane];       // hi plane, lo plane
  constexpr int kXO = MfX<GEN_NOISE>::O;             // copy O's hi plane (its lo plane, supplied noise only: + kMfXPlane)
  __shared__ __attribute__((aligned(16))) unsigned char s_x_all[2][MfX<GEN_NOISE>::BYTES];      // planes E hi, [E lo,] O hi [, O lo]
  __shared__ __attribute__((aligned(16))) std::conditional_t<GEN_NOISE, MfHandoff, uint4> s_hand[2];      // (generated noise: see the producers)
  __shared__ __attribute__((aligned(16))) float s_carry[2][kMfPW - 1][128];       // right halves handed from FIR wavefront w to w + 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = lane & 15, mg = lane >> 4;            // MFMA fragment coordinates
  const int n_my = (p.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // tiles of this block
  long long* dbg = (p.dbg && (int)blockIdx.x == p.dbg_block) ? p.dbg : nullptr;                              // block 0: [tick][role][2]
#define DDSP_MF_STAMP(tick, role, i) do { if (dbg && lane == 0 && (tick) + 1 < 16) dbg[(((tick) + 1) * 3 + (role)) * 2 + (i)] = wall_clock64(); } while (0)
  // tile T -> (batch row, tile tx of the row): first output z0 = 3968 tx (a multiple of 64); tap rows: frames f_first ..
  // f_first + 63 (frames of the inputs x[z0-128 ..]); negative for the first tile.  The rows' LAST tiles come last in the
  // order (T >= n_whole): they are mostly beyond the end of the row (a 4 s clip is 16.1 tiles) and cost a fraction of a
  // tick, so that a block's longest schedule is 8 whole tiles + 1 short one at batch 128 (2 + 1 at batch 32), not 9 (3).
  // (divisions by launch constants: fastdiv, common.h)
#define DDSP_MF_TILE(T, b_, z0_, ffirst_, rel0_)                                                        \
  const bool last_ = (T) >= p.n_whole;                                                                  \
  uint32_t tx_;                                                                                         \
  const int bq_ = (int)fastdiv((uint32_t)(T), p.whole_per_row, tx_);                                    \
  const int b_ = last_ ? (T) - p.n_whole : bq_;                                                         \
  const int z0_ = (last_ ? p.tiles_per_row - 1 : (int)tx_) * kMfTile;                                   \
  uint32_t fr_;                                                                                         \
  const int fq_ = (int)fastdiv((uint32_t)(z0_ >= 128 ? z0_ - 128 : 128 - z0_ + p.fs - 1), p.fs_div, fr_); \
  const int ffirst_ = (z0_ >= 128) ? fq_ : -fq_;                                                        \
  const int rel0_ = (z0_ - 128) - ffirst_ * p.fs

  if (wave < kMfPW) {
    // =========================== producer wavefronts ==================================================================
    // Two kinds (round 3).  DESIGNERS (wavefronts 0 .. P/2 - 1): wavefront w turns the 16 x 65 magnitudes of rows
    // 16 w .. + 15 into BOTH tap tiles - 2 x 6 MFMAs against the constant cosine fragments, window, split, tap writes.
    // NOISE MAKERS (wavefronts P/2 .. P - 1): the noise tile, a half each.
    // Round 4, generated noise (HANDOFF): its noise costs a third of what it did (eight fp16 samples per Philox block, hi planes
    // only) and the designers had become what a tick waits for (2.45 us against the FIR wavefronts' 1.7 - 2.2 and the noise makers'
    // 0.9: profiles/r04z_*): the noise makers now also FETCH, SCALE (exp_sigmoid) and SPLIT the magnitudes - of the tile after the
    // one the designers are working on - and hand them over as ready-made B-fragments in the LDS the noise's lo planes used to
    // take; they write the controls and finish tap 32 (a dot product over the even bins) while they are at it.  A designer's tick is
    // four fragment reads, the MFMAs and the tap writes.  The block's first tile has nobody a tile ahead of it: its designers scale
    // for themselves, as every tile's do when the noise is supplied.
    constexpr bool HANDOFF = GEN_NOISE;
    const float kLog10 = 2.302585092994046f;
    const bool designer = wave < kMfPW / 2;
    const int rg = wave & (kMfPW / 2 - 1);           // a designer's (and a scaling noise maker's) row group
    // zeros that stay: the zero group of the tap rows (what lanes outside the filter's support read) and the 16 elements
    // between the reversed noise frames (both copies), in both buffers
    if (designer)
      *reinterpret_cast<uint4*>(s_taps_all[lane >> 5] + (lane & 1) * kMfTapPlane + (16 * rg + ((lane >> 1) & 15)) * kMfTapRowBytes + 256) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 2 * (kMfRows + 1) * 16; i += 64 * kMfPW) {
      unsigned char* const s_xe = s_x_all[i >= (kMfRows + 1) * 16 ? 1 : 0];
      unsigned char* const s_xo = s_xe + kXO;
      const int ii = i >= (kMfRows + 1) * 16 ? i - (kMfRows + 1) * 16 : i;
      const int e = 80 * (ii >> 4) + (ii & 15);                       // element index of a padding element
      // copy E: element e is half e & 1 of dword e >> 1; copy O: half (e + 1) & 1 of dword (e + 1) >> 1
      *reinterpret_cast<uint16_t*>(s_xe + e * 2) = 0;
      *reinterpret_cast<uint16_t*>(s_xo + (e + 1) * 2) = 0;
      if (!GEN_NOISE) {
        *reinterpret_cast<uint16_t*>(s_xe + kMfXPlane + e * 2) = 0;
        *reinterpret_cast<uint16_t*>(s_xo + kMfXPlane + (e + 1) * 2) = 0;
      }
    }
    const int rrow = 16 * rg + mi;
    // Frames longer than 64 samples: the tile's 32 staged frames of 64 share ceil-ish 2048 / fs tap rows - at 192 samples (BASELINE
    // configs[4]) twelve, all in row group 0.  A row group none of whose rows any staged frame reads is neither fetched nor scaled
    // nor designed (until round 4 it was: 130 MB of HBM traffic per launch against 82 algorithmic at that shape, pmc_traffic_config5.json).
    // (the frames a tile owns for the controls output - scale_rows - lie among the rows its taps need or one past them)
    auto group_unused = [&](int rel0, int z0, int f_first) -> bool {
      if (FS64) return false;
      uint32_t r_;
      const int last_tap_row = (int)fastdiv((uint32_t)(rel0 + 64 * (kMfRows - 1)), p.fs_div, r_);
      const int own_hi = (int)fastdiv((uint32_t)(z0 + kMfTile - 128), p.fs_div, r_) + 2;          // as in scale_rows
      return 16 * rg > last_tap_row && 16 * rg >= own_hi - f_first;
    };
    // the 16 bins of this lane's B-fragments (row = 16 rg + i, bins 16 g .. + 15) and bin 64 of that row, straight from
    // HBM; rows outside [0, F) are fetched from frame 0 and masked afterwards (unconditional loads)
    MfU4f rq[4];
    float r_last;
    auto fetch_rows = [&](int tick_of_tile) {        // the rows of the block's tile number `tick_of_tile` (clamped to its last)
      const int T = (int)blockIdx.x + min(max(tick_of_tile, 0), n_my - 1) * (int)gridDim.x;
      DDSP_MF_TILE(T, b, z0, f_first, rel0);
      if (group_unused(rel0, z0, f_first)) return;     // (wave-uniform)
      const int rfr = f_first + rrow;
      const float* __restrict__ src = mag + ((size_t)b * p.F + ((rfr >= 0 && rfr < p.F) ? rfr : 0)) * 65;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) rq[c4] = *reinterpret_cast<const MfU4f*>(src + 16 * mg + 4 * c4);
      r_last = src[64];
    };
    // the fetched rows of tile T scaled, masked and split: the four B-fragments, bin 64, tap 32's value; the controls written.
    // (false: the tile's rows lie past the end of the clip - nobody reads their taps)
    auto scale_rows = [&](int T, mf_f16x8& be_hi, mf_f16x8& be_lo, mf_f16x8& bo_hi, mf_f16x8& bo_lo, float& m_last, float& t32) -> bool {
      DDSP_MF_TILE(T, b, z0, f_first, rel0);
      // controls ownership: tile t writes frames [own_lo, own_hi) so that every frame is written once
      const int own_lo = (z0 == 0) ? 0 : f_first + 2;
      uint32_t own_r;
      const int own_hi = (int)fastdiv((uint32_t)(z0 + kMfTile - 128), p.fs_div, own_r) + 2;
      const int rfr = f_first + rrow;
      const bool rvalid = rfr >= 0 && rfr < p.F;
      // rows past frame F + 1 are read by no FIR wavefront that stores anything (see `active` there): skipped whole
      if (f_first + 16 * rg > p.F + 1 || group_unused(rel0, z0, f_first)) return false;
      float y[16];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) { y[4 * c4] = rq[c4].x; y[4 * c4 + 1] = rq[c4].y; y[4 * c4 + 2] = rq[c4].z; y[4 * c4 + 3] = rq[c4].w; }
      m_last = r_last;
      if (p.scale & 1) {
#pragma unroll
        for (int c = 0; c < 16; ++c) y[c] = exp_sigmoid_fast(y[c] + p.bias, kLog10, 2.0f, 1e-7f);
        m_last = exp_sigmoid_fast(m_last + p.bias, kLog10, 2.0f, 1e-7f);
      }
      if (!rvalid) {
#pragma unroll
        for (int c = 0; c < 16; ++c) y[c] = 0.0f;
        m_last = 0.0f;
      }
      if (ctl_out && rvalid && rfr >= own_lo && rfr < own_hi) {       // written by the owning tile only
        float* __restrict__ dst = ctl_out + ((size_t)b * p.F + rfr) * 65;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
          *reinterpret_cast<MfU4f*>(dst + 16 * mg + 4 * c4) = MfU4f{y[4 * c4], y[4 * c4 + 1], y[4 * c4 + 2], y[4 * c4 + 3]};
        if (mg == 0) dst[64] = m_last;
      }
      float ve[8], vo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { ve[e] = y[2 * e]; vo[e] = y[2 * e + 1]; }
      mf_split8(ve, be_hi, be_lo);
      mf_split8(vo, bo_hi, bo_lo);
      // tap 32: cos(pi m / 2) vanishes for odd bins; this lane's 8 even bins, then the row's four lanes together
      const float* __restrict__ c32 = kIr65.c + 32 * kIrRowStride + 8 * mg;
      float part = (mg == 0) ? m_last * kIr65.c[32 * kIrRowStride + 32] : 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(c32[e], ve[e], part);
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      t32 = kIr65.win[32] * part;
      return true;
    };
    // the noise of tile tick + 1.  Supplied noise: the noise makers, a half each.  Generated noise (HANDOFF): all four producers, a
    // quarter each - the noise makers scale the magnitudes as well now, and with the whole noise tile on top they were what a tick
    // waited for (r05a timeline: designers 1.1 - 1.6 us, FIR 1.7 - 1.9, ticks of 2.6)
    auto make_noise = [&](int tick) {
      if (tick >= 0 && tick + 1 < n_my) {            // (tile 0's noise: the FIR wavefronts, idle in tick -1)
        const int T = (int)blockIdx.x + (tick + 1) * (int)gridDim.x;
        DDSP_MF_TILE(T, b, z0, f_first, rel0);
        (void)rel0; (void)f_first;
        unsigned char* const s_xe = s_x_all[(tick + 1) & 1];
        if constexpr (HANDOFF) mf_noise_tile<GEN_NOISE, 64 * kMfPW>(tid, b, z0, x, s_xe, s_xe + kXO, p);
        else mf_noise_tile<GEN_NOISE, 32 * kMfPW>(tid - 32 * kMfPW, b, z0, x, s_xe, s_xe + kXO, p);
      }
    };
    if (!designer) {
      // ------------------------- noise makers -------------------------------------------------------------------------
      if constexpr (HANDOFF) fetch_rows(1);          // (the block's second tile: its first is the designers' own)
#pragma unroll 1
      for (int tick = -1; tick < n_my; ++tick) {
        if (wave == p.dbg_wave - 8) DDSP_MF_STAMP(tick, 0, 0);
        if constexpr (HANDOFF) {
          // the magnitudes of tile tick + 2, for the designers' next tick
          if (tick + 2 < n_my) {
            const int T = (int)blockIdx.x + (tick + 2) * (int)gridDim.x;
            MfHandoff& h = s_hand[(tick + 2) & 1];
            mf_f16x8 be_hi, be_lo, bo_hi, bo_lo;
            float m_last, t32;
            if (scale_rows(T, be_hi, be_lo, bo_hi, bo_lo, m_last, t32)) {
              h.frag[rg][0][lane] = __builtin_bit_cast(uint4, be_hi);
              h.frag[rg][1][lane] = __builtin_bit_cast(uint4, be_lo);
              h.frag[rg][2][lane] = __builtin_bit_cast(uint4, bo_hi);
              h.frag[rg][3][lane] = __builtin_bit_cast(uint4, bo_lo);
              if (mg == 0) { h.last[rrow] = m_last; h.tap32[rrow] = t32; }
            }
          }
          fetch_rows(tick + 3);                      // used a tick from now
        }
        make_noise(tick);
        if (wave == p.dbg_wave - 8) DDSP_MF_STAMP(tick, 0, 1);
        __syncthreads();
      }
    } else {
      // ------------------------- designers ----------------------------------------------------------------------------
      // the constant cosine factor of both tap tiles as fp16 hi / lo A-fragments, made at compile time:
      // [tap tile][even / odd bins][hi / lo]
      mf_f16x8 afr[2][2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int hl = 0; hl < 2; ++hl) {
            const uint4 v = *reinterpret_cast<const uint4*>(kIr65Frags.v[mt][par][hl][lane]);
            afr[mt][par][hl] = mf_frag(v.x, v.y, v.z, v.w);
          }
      // the two tap tiles of this wavefront's rows from their scaled, split magnitudes
      auto design_rows = [&](unsigned char* s_taps, const mf_f16x8& be_hi, const mf_f16x8& be_lo, const mf_f16x8& bo_hi,
                             const mf_f16x8& bo_lo, float m_last, float t32) {
          unsigned char* __restrict__ hrow = s_taps + rrow * kMfTapRowBytes;
          const mf_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            // e(n) = sum_i ce[n][i] m[2i], o(n) = sum_i co[n][i] m[2i+1], n = 16 mt + 4 g + r: D[n][row i]
            mf_f32x4 ea = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[mt][0][0], be_hi, zero, 0, 0, 0);
            mf_f32x4 oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[mt][1][0], bo_hi, zero, 0, 0, 0);
            mf_f32x4 ex = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[mt][0][0], be_lo, zero, 0, 0, 0);
            mf_f32x4 ox = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[mt][1][0], bo_lo, zero, 0, 0, 0);
            ex = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[mt][0][1], be_hi, ex, 0, 0, 0);
            ox = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[mt][1][1], bo_hi, ox, 0, 0, 0);
            const mf_f32x4 ev = ea + ex * (1.0f / kMfLoScale), ov = oa + ox * (1.0f / kMfLoScale);
            const int n0 = 16 * mt + 4 * mg;                                // this lane's taps n0 .. n0 + 3
            float g0[4], g1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = n0 + r;
              const float e = fmaf(m_last, kIr65.c[n * kIrRowStride + 32], ev[r]);      // + bin 64 (a rank-1 update)
              const float o = ov[r];
              g0[r] = kIr65.win[n] * (e + o);                                // g[n]:    taps 64 + n and 64 - n
              g1[r] = (n >= 1) ? kIr65.win[64 - n] * (e - o) : 0.0f;        // g[64-n]: taps 128 - n and n; tap 0 is 0
            }
            mf_put4(hrow, 64 + n0, g0[0], g0[1], g0[2], g0[3]);
            mf_put4(hrow, n0, g1[0], g1[1], g1[2], g1[3]);
            mf_put4_down(hrow, 64 - n0, true, g0[0], g0[1], g0[2], g0[3]);  // n0 = 0: tap 64 once more, the same value
            mf_put4_down(hrow, 128 - n0, n0 != 0, g1[0], g1[1], g1[2], g1[3]);
          }
          if (mg == 0) {                                                     // tap 32 (and its mirror image, tap 96)
            _Float16 h, l;
            mf_split(t32, h, l);
            const uint16_t hb = __builtin_bit_cast(uint16_t, h), lb = __builtin_bit_cast(uint16_t, l);
            *reinterpret_cast<uint16_t*>(hrow + 96 * 2) = hb;              // tap 96
            *reinterpret_cast<uint16_t*>(hrow + kMfTapPlane + 96 * 2) = lb;
            *reinterpret_cast<uint16_t*>(hrow + 32 * 2) = hb;              // tap 32
            *reinterpret_cast<uint16_t*>(hrow + kMfTapPlane + 32 * 2) = lb;
          }
      };
      fetch_rows(0);                                 // the block's first tile
      // one tick of a designer: tile tick + 1, its magnitudes scaled here (SCALE_HERE: the block's first tile, and every tile when the
      // noise is supplied) or taken from the noise makers' handoff
      auto designer_tick = [&](int tick, auto scale_here_tag) {
        constexpr bool SCALE_HERE = decltype(scale_here_tag)::value;
        if (wave == p.dbg_wave - 8) DDSP_MF_STAMP(tick, 0, 0);
        if (tick + 1 < n_my) {
          const int T = (int)blockIdx.x + (tick + 1) * (int)gridDim.x;
          unsigned char* const s_taps = s_taps_all[(tick + 1) & 1];
          if (wave == p.dbg_wave - 8) DDSP_MF_STAMP(tick, 1, 0);
          if constexpr (SCALE_HERE) {
            mf_f16x8 be_hi, be_lo, bo_hi, bo_lo;
            float m_last = 0.0f, t32 = 0.0f;
            if (scale_rows(T, be_hi, be_lo, bo_hi, bo_lo, m_last, t32)) design_rows(s_taps, be_hi, be_lo, bo_hi, bo_lo, m_last, t32);
          } else if constexpr (HANDOFF) {
            DDSP_MF_TILE(T, b, z0, f_first, rel0);
            (void)b;
            if (f_first + 16 * rg <= p.F + 1 && !group_unused(rel0, z0, f_first)) {        // (the noise maker's own test: scale_rows)
              const MfHandoff& h = s_hand[(tick + 1) & 1];
              design_rows(s_taps, __builtin_bit_cast(mf_f16x8, h.frag[rg][0][lane]), __builtin_bit_cast(mf_f16x8, h.frag[rg][1][lane]),
                          __builtin_bit_cast(mf_f16x8, h.frag[rg][2][lane]), __builtin_bit_cast(mf_f16x8, h.frag[rg][3][lane]),
                          h.last[rrow], h.tap32[rrow]);
            }
          }
          if (wave == p.dbg_wave - 8) DDSP_MF_STAMP(tick, 1, 1);
        }
        // supplied noise: the magnitudes of tile tick + 2, used a tick from now - fetched at the END of the tick, into the
        // registers this tick's magnitudes have just left (past the block's last tile: the last one again)
        if (!HANDOFF) fetch_rows(tick + 2);
        if constexpr (HANDOFF) make_noise(tick);
        if (wave == p.dbg_wave - 8) DDSP_MF_STAMP(tick, 0, 1);
        __syncthreads();
      };
      designer_tick(-1, std::true_type{});
#pragma unroll 1
      for (int tick = 0; tick < n_my; ++tick) designer_tick(tick, std::integral_constant<bool, !HANDOFF>{});
    }
  } else {
    // =========================== FIR wavefronts: pairs of frames on the matrix cores =================================
    // per-lane constants of the five k-steps.  Step c, lane (i = lane & 15, g = lane >> 4): block p = c of the pair's
    // first (g < 2) or second (g >= 2) frame, d = 8 (g & 1) + e.
    //   A (row b = i): x_frame[16 p + b - d], e = 0..7 = reversed elements u .. u + 7, u = 79 + 80 s - 16 p - b + 8 (g & 1)
    //   B (col a = i): h_row[16 (a' - p) + 8 (g & 1) + e], a' = a - 4 (second frame); zero group outside 0 <= a' - p <= 7
    // With p = c in every lane, B of step c + 1 is B of step c moved ONE column up (a' - p - 1 is what column a - 1 had),
    // zeros entering at column 0: only step 0 reads the tap table, steps 1 .. 4 are 8 DPP moves each (row_shr:1).  The LDS
    // is what bounds the FIR wavefronts; this takes 8 of a pass's 30 reads and 40 % of its bytes away.
    // Pair P = staged frames 2 P, 2 P + 1; pairs 1 .. 31 are output.  FIR wavefront cw walks the four pairs 4 cw ..
    // 4 cw + 3 (unrolled: every LDS address is a per-lane register plus an immediate).  Pair 0 only builds the carry
    // into pair 1 (its own outputs belong to the previous tile).  The left half of wavefront cw >= 1's first pair lacks
    // what pair 4 cw - 1 spills into it: that right half is left in LDS by wavefront cw - 1 (s_carry), the incomplete
    // left half stays in registers, and the two are added and stored at the start of the NEXT tick, behind the block's
    // barrier (a warm-up pair per wavefront instead would be a fifth pass: 25 % more MFMAs and LDS reads, and the LDS is
    // what bounds this phase).
    const int cw = wave - kMfPW;
    const int p_first = 4 * cw;
    mf_f32x4 kept = {0.f, 0.f, 0.f, 0.f};      // the incomplete left half of this wavefront's first pair (cw >= 1), last tile
    float* kept_ot = nullptr;                  // where it goes: the tile's base pointer, offset and bounds of the previous tick
    long kept_n = 0;
    int kept_exp = 0;                          // (supplied noise: the exponent its row was normalised by)
    const int xsel = ((mi & 1) ? 0 : kXO) + 2 * kMfXStride * 2 * p_first;       // i odd -> u even -> copy E
    const int second = mg >> 1;
    // u even -> copy E, dword u / 2; u odd -> copy O, dword (u + 1) / 2 (u's parity is the lane's: 79 - i).  Step c
    // reads 32 c bytes below step 0, pair `it` 320 `it` bytes above pair 0: ONE per-lane address per plane and pair of
    // pairs, the rest immediates (ds_read2_b32 reaches 1020 bytes).  (Ten addresses, one per step and plane, cost ten
    // VGPRs that the 128-VGPR budget spilled to scratch: 5.8 MB of HBM writes per launch, profiles/pmc_traffic.json.)
    const int u4 = 79 + kMfXStride * second - 16 * 4 - mi + 8 * (mg & 1);              // step 4, relative to the pair's first frame
    const int a_base0 = xsel + ((u4 + 1) >> 1) * 4;
    const int q0 = mi - 4 * second;
    const int b_off = (q0 >= 0 && q0 <= 7) ? (2 * q0 + (mg & 1)) * 16 : 256;
    const int b_ptr0 = (2 * p_first + second) * kMfTapRowBytes + b_off;
    const int lane_off = 16 * mi + 4 * mg;
    // the deferred boundary outputs of the previous tile: kept left half + the neighbour's right half
    auto flush_kept = [&](int parity) {
      if (cw >= 1 && kept_ot != nullptr && mi < 8) {
        const float4 cr = *reinterpret_cast<const float4*>(&s_carry[parity][cw - 1][lane_off]);
        float v[4] = {kept[0] + cr.x, kept[1] + cr.y, kept[2] + cr.z, kept[3] + cr.w};
        if constexpr (!GEN_NOISE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ldexpf(v[r], kept_exp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kept_n + lane_off + r >= 0 && kept_n + lane_off + r < p.N) kept_ot[lane_off + r] = v[r];
      }
    };
#pragma unroll 1
    for (int tick = -1; tick < n_my; ++tick) {
      if (wave == p.dbg_wave - 8 + kMfPW) DDSP_MF_STAMP(tick, 2, 0);
      if (tick >= 1) flush_kept((tick - 1) & 1);
      if (tick < 0) {
        // nothing to filter yet: the noise of the block's first tile, while the producers design its taps
        const int T = (int)blockIdx.x;
        DDSP_MF_TILE(T, b, z0, f_first, rel0);
        (void)f_first; (void)rel0;
        mf_noise_tile<GEN_NOISE, 64 * kMfPW>(tid - 64 * kMfPW, b, z0, x, s_x_all[0], s_x_all[0] + kXO, p);
      }
      bool active = false;
      if (tick >= 0) {
        const int T = (int)blockIdx.x + tick * (int)gridDim.x;
        DDSP_MF_TILE(T, b, z0, f_first, rel0);
        (void)f_first;
        // (a wavefront whose four pairs lie past the end of the row has nothing to store: most of a row's last tile)
        active = (long)z0 - 128 + 128L * p_first - p.start < (long)p.N;
      }
      if (active) {
        const int T = (int)blockIdx.x + tick * (int)gridDim.x;
        DDSP_MF_TILE(T, b, z0, f_first, rel0);
        (void)f_first;
        const unsigned char* const s_taps = s_taps_all[tick & 1];
        const unsigned char* const s_x = s_x_all[tick & 1];
        float* __restrict__ o = out + (size_t)b * p.N;
        // out index of this lane's first value of pair 8 cw: z = z0 - 128 + 128 P + 16 a + 4 g (+ r)
        const long n_tile = (long)z0 - 128 + 128L * p_first - p.start;
        // every store of this wavefront inside [0, N) and 8-byte aligned: no per-element checks
        const bool interior = n_tile + 128 >= 0 && n_tile + 128 * 4 <= (long)p.N && (((p.start | p.N) & 1) == 0);
        float* __restrict__ ot = o + n_tile;                               // wave-uniform base; lanes add a 32-bit offset
        int o_exp = 0;                                                     // supplied noise: its row's exponent back at the stores
        if constexpr (!GEN_NOISE) {
          if (p.x_absmax) o_exp = pow2_exponent(p.x_absmax[b]);
        }
        mf_f32x4 carry = {0.f, 0.f, 0.f, 0.f};
        int a_hi[2], a_lo[2], b_ptr = b_ptr0;
        DDSP_KEEP_IN_VGPR(b_ptr);
        __builtin_assume((b_ptr & 15) == 0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {                  // pairs 0, 1 and pairs 2, 3
          a_hi[k] = a_base0 + 640 * k; a_lo[k] = a_base0 + 640 * k + kMfXPlane;
          DDSP_KEEP_IN_VGPR(a_hi[k]);
          DDSP_KEEP_IN_VGPR(a_lo[k]);
          // what the compiler no longer sees through the barrier: dword-aligned fragment addresses (two ds_read2_b32
          // each; a ds_read_b128 at 4-byte alignment would take the slow unaligned path), 16-byte aligned tap groups
          __builtin_assume((a_hi[k] & 3) == 0);
          __builtin_assume((a_lo[k] & 3) == 0);
        }
        // The 4 x 5 (pair, k-step) sequence runs as ONE software pipeline (no control flow inside, so the compiler's
        // s_waitcnt counts stay exact): the fragment reads of step j + 3 are issued right
        // after the MFMAs of step j (three steps = 18 LDS operations in flight per wavefront: the LDS only reaches its rate
        // with many operations outstanding, and two FIR wavefronts share a SIMD), the epilogue of a pair (combine, store,
        // shift the right half over) runs under the reads of the next pair.  The carry enters in the epilogue, not as
        // the accumulator's start value, so no MFMA waits for a previous pair.
        mf_f16x8 fah[3], fal[3], fbh_in[2], fbl_in[2], fbh, fbl;
        auto load_step = [&](int j, int slot) {
          const int it = j / 5, c = j - 5 * it;
          const MfU4 qh = *reinterpret_cast<const MfU4*>(s_x + a_hi[it >> 1] + 32 * (4 - c) + 2 * kMfXStride * 2 * (it & 1));
          fah[slot] = mf_frag(qh.x, qh.y, qh.z, qh.w);
          if constexpr (!GEN_NOISE) {           // (generated noise is fp16 as it stands: no lo plane, no lo . hi product)
            const MfU4 ql = *reinterpret_cast<const MfU4*>(s_x + a_lo[it >> 1] + 32 * (4 - c) + 2 * kMfXStride * 2 * (it & 1));
            fal[slot] = mf_frag(ql.x, ql.y, ql.z, ql.w);
          }
          if (c == 0) {
            // the pair's taps: first frame's row in lanes g < 2, second frame's in g >= 2
            const unsigned char* tr;
            if (FS64) {
              tr = s_taps + b_ptr + 2 * kMfTapRowBytes * it;
            } else {
              // tap rows of the pair's two frames (frame size fs = 64 c: c staged frames share a row)
              const int P = p_first + it;
              const int row = (int)(((float)(rel0 + 128 * P + 64 * second) + 0.5f) * p.inv_fs);
              tr = s_taps + row * kMfTapRowBytes + b_off;
            }
            fbh_in[it & 1] = *reinterpret_cast<const mf_f16x8*>(tr);
            fbl_in[it & 1] = *reinterpret_cast<const mf_f16x8*>(tr + kMfTapPlane);
          }
        };
        // one column up (row_shr:1 within the 16 lanes of a g; column 0 receives zeros)
        auto column_up = [](mf_f16x8 v) {
          typedef int i32x4 __attribute__((ext_vector_type(4)));
          const i32x4 w = __builtin_bit_cast(i32x4, v);
          i32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int t = w[k];                // (a scalar temporary: see the carry below)
            o[k] = __builtin_amdgcn_update_dpp(0, t, 0x111, 0xF, 0xF, true);
          }
          return __builtin_bit_cast(mf_f16x8, o);
        };
        load_step(0, 0);
        load_step(1, 1);
        load_step(2, 2);
        __builtin_amdgcn_sched_barrier(0);
        mf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_hl = {0.f, 0.f, 0.f, 0.f}, acc_lh = {0.f, 0.f, 0.f, 0.f};   // three independent chains
        auto pipeline = [&](auto interior_tag) {
          constexpr bool kInterior = decltype(interior_tag)::value;
#pragma unroll
          for (int j = 0; j < 20; ++j) {
            const int it = j / 5, c = j - 5 * it, slot = j % 3;
            if (c == 0) { fbh = fbh_in[it & 1]; fbl = fbl_in[it & 1]; }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[slot], fbh, acc, 0, 0, 0);
            acc_hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[slot], fbl, acc_hl, 0, 0, 0);
            if constexpr (!GEN_NOISE) acc_lh = __builtin_amdgcn_mfma_f32_16x16x32_f16(fal[slot], fbh, acc_lh, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 3 < 20) load_step(j + 3, slot);
            if (c < 4) { fbh = column_up(fbh); fbl = column_up(fbl); }      // under the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            if (c == 4) {
              const mf_f32x4 comb = (acc + carry) + (acc_hl + acc_lh) * (1.0f / kMfLoScale);
              acc = (mf_f32x4){0.f, 0.f, 0.f, 0.f};
              acc_hl = (mf_f32x4){0.f, 0.f, 0.f, 0.f};
              acc_lh = (mf_f32x4){0.f, 0.f, 0.f, 0.f};
              // D[row b = 4 g + r][col a = i]: columns 0..7 are complete (the first pair's: see above)
              if (it == 0) {
                kept = comb;
              } else if (mi < 8) {
                const int idx = lane_off + 128 * it;                         // out index n = n_tile + idx
                mf_f32x4 st4 = comb;
                if constexpr (!GEN_NOISE) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) { const float v = comb[r]; st4[r] = ldexpf(v, o_exp); }
                }
                if (kInterior) {
                  *reinterpret_cast<float2*>(ot + idx) = make_float2(st4[0], st4[1]);
                  *reinterpret_cast<float2*>(ot + idx + 2) = make_float2(st4[2], st4[3]);
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const float v = st4[r];
                    if (n_tile + idx + r >= 0 && n_tile + idx + r < p.N) ot[idx + r] = v;
                  }
                }
              }
              // columns 8..15 -> columns 0..7 of the next pair's tile (row_shl:8, out-of-row lanes read 0)
              // (through a scalar temporary: __builtin_bit_cast applied to a vector ELEMENT reads element 0 - clang 22)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = comb[r];
                carry[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x108, 0xF, 0xF, true));
              }
              // the last pair's right half: the next wavefront's missing part (the last wavefront's lies past the tile)
              if (it == 3 && cw < kMfPW - 1 && mi >= 8)
                *reinterpret_cast<float4*>(&s_carry[tick & 1][cw][lane_off - 128]) = make_float4(comb[0], comb[1], comb[2], comb[3]);
            }
          }
        };
        if (interior) pipeline(std::true_type{});
        else pipeline(std::false_type{});
        kept_ot = ot;
        kept_n = n_tile;
        kept_exp = o_exp;
      } else {
        kept_ot = nullptr;
      }
      if (wave == p.dbg_wave - 8 + kMfPW) DDSP_MF_STAMP(tick, 2, 1);
      __syncthreads();
    }
    flush_kept((n_my - 1) & 1);                // the last tile's boundaries (nothing writes s_carry any more)
  }
#undef DDSP_MF_STAMP
#undef DDSP_MF_TILE
}

// (scale = 0 - magnitudes that are not squashed by exp_sigmoid, at any scale - goes to the general kernels, which normalise what
// they split: this kernel's designers split the magnitudes as they come)
bool noise_mfma65_ok(int F, int M, int N, int padding, const void* noise, int scale) {
  const int frame_size = (N + F - 1) / F;
  return scale != 0 && M == 65 && padding == 0 && frame_size >= 64 && (frame_size % 64) == 0 && frame_size <= 4096 &&
         (N + frame_size - 1) / frame_size == F && (noise == nullptr || (((uintptr_t)noise) & 15) == 0);
}

// max |x[row]| of every row: what supplied noise is normalised by (one block per row; a parity / FIRFilter entry, not a hot path)
__global__ __launch_bounds__(1024) void row_absmax_kernel(const float* __restrict__ x, int N, float* __restrict__ out) {
  __shared__ float s_m[16];
  const float* row = x + (size_t)blockIdx.x * N;
  float m = 0.0f;
  for (int i = threadIdx.x; i < N; i += 1024) m = fmaxf(m, fabsf(row[i]));
  m = wave_max_nonneg_dpp(m);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = 0.0f;
    for (int w = 0; w < 16; ++w) r = fmaxf(r, s_m[w]);
    out[blockIdx.x] = r;
  }
}

int launch_noise_mfma65(const float* magnitudes, const float* noise, float* audio, float* ctl_magnitudes, int B, int F,
                        int N, int start, float initial_bias, int scale, uint64_t seed, uint64_t batch_offset,
                        long long* dbg, int bits23, float* row_scratch /* B floats */, hipStream_t st) {
  MfArgs q;
  q.x_absmax = nullptr; q.bits23 = bits23;
  if (noise) {
    if (!row_scratch) return DDSP_ERR_WORKSPACE;
    hipLaunchKernelGGL(row_absmax_kernel, dim3((unsigned)B), dim3(1024), 0, st, noise, N, row_scratch);
    q.x_absmax = row_scratch;
  }
  const bool lo_planes = noise != nullptr || bits23 != 0;          // the hi / lo instances: supplied noise, or 23-bit samples made on chip
  q.N = N; q.F = F; q.start = start; q.bias = initial_bias; q.scale = scale;
  q.fs = (N + F - 1) / F; q.inv_fs = 1.0f / (float)q.fs;
  q.k0 = (uint32_t)seed; q.k1 = (uint32_t)(seed >> 32); q.batch_offset = batch_offset;
  q.dbg = dbg;
  static const int dbg_wave = [] { const char* e = getenv("DDSP_MF_DBG_WAVE"); const int v = e ? atoi(e) : 8; return v >= 8 && v < 8 + kMfPW ? v : 8; }();
  q.dbg_wave = dbg_wave;
  static const int dbg_block = [] { const char* e = getenv("DDSP_MF_DBG_BLOCK"); return e ? atoi(e) : 0; }();
  q.dbg_block = dbg_block;
  q.tiles_per_row = (N + start + kMfTile - 1) / kMfTile;
  q.n_tiles = B * q.tiles_per_row;
  q.whole_per_row = make_fastdiv((uint32_t)(q.tiles_per_row > 1 ? q.tiles_per_row - 1 : 1));
  q.fs_div = make_fastdiv((uint32_t)q.fs);
  q.n_whole = B * (q.tiles_per_row - 1);
  // persistent grid: one block of 12 wavefronts per CU (two LDS buffers of 76 KB)
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  const int slots = n_cu * (kMfWaves == 16 ? 1 : 2);
  const dim3 grid((unsigned)(q.n_tiles < slots ? q.n_tiles : slots));
  hipEvent_t ev0, ev1;
  profile_kernel_events(kNoiseMfma, &ev0, &ev1);
#define DDSP_LAUNCH_MF(GEN, FS64)                                                                                      \
  hipExtLaunchKernelGGL((noise_mfma65_kernel<GEN, FS64>), grid, dim3(64 * kMfWaves), 0, st, ev0, ev1, 0, magnitudes, noise, \
                        ctl_magnitudes, audio, q)
  if (q.fs == 64) { if (lo_planes) DDSP_LAUNCH_MF(false, true); else DDSP_LAUNCH_MF(true, true); }
  else { if (lo_planes) DDSP_LAUNCH_MF(false, false); else DDSP_LAUNCH_MF(true, false); }
#undef DDSP_LAUNCH_MF
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

}  // namespace ddsp
