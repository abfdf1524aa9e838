// Backward of Harmonic.__call__ w.r.t. the per-frame harmonic amplitudes as the ADJOINT of the wavetable synthesis
// (harmonic_table.hip): what tf.GradientTape computes through ddsp/core.py:912-962 (oscillator_bank) and :1048-1111
// (harmonic_synthesis) for dL/d(amplitude envelopes), reduced to frame rate.
//
//   P[j,k] = sum_r w_cur(r)  g(n) s_k(n),   Q[j,k] = sum_r w_next(r) g(n) s_k(n),   n = j hop + r,  s_k(n) = sin(2 pi k theta(n))
// (harmonic.hip, "Backward pass": harm_bwd_chain_kernel turns P and Q into dL/d amplitudes and dL/d harmonic_distribution).
// harm_bwd_pq_kernel evaluates one quarter-rate v_sin_f32 per sample AND harmonic: 55-60 us at batch 32 against 15 us for
// the forward, which evaluates none.  The forward's trick run backwards: with the interpolation window psi and its transform
//     sin(2 pi k theta) ~= 1 / psi_hat(k) sum_i sin(2 pi k (i + 1/2) / T) psi(T theta - i - 1/2)            (T = 512 points)
// so   P[k] ~= 1 / psi_hat(k) sum_i sin(2 pi k (i + 1/2) / T) G[i],     G[i] = sum_n c_n psi(T theta_n - i - 1/2),  c_n = w(r) g(n)
// - a type-1 nonuniform FFT: SPREAD the weighted gradient samples onto the table grid through the window's W taps, then ONE
// product with the transposed constant sine matrix, which belongs on the matrix cores (the forward's fragments, transposed).
// Same window, same 1 / psi_hat, same accuracy as the forward (<= 6.5e-6 sum |c| per harmonic).
//
// One block = 8 wavefronts; a GROUP = 8 consecutive frames (of the flattened [B F] rows); a block walks groups blockIdx.x,
// blockIdx.x + gridDim.x, ..
//   1. Spreading, wavefront w = frame w, lanes = samples of a tile of 64.  A scatter in which neighbouring lanes hit overlapping
//      entries - round 2 tried it with LDS float atomics and lost (165 us; round 4 measured why: a ds_add_f32 holds the LDS
//      pipeline ~100 clocks).  Here no atomics and no conflicts: tap number t of lane n goes to entry floor(pos_n) + t - W/2 + 1;
//      positions grow by >= 1 entry per sample (checked per frame: f0 >= sr / 512), so for a FIXED tap number the lanes of one
//      revolution write DISTINCT entries - a plain read-modify-write per tap number is race free, W of them in sequence (LDS
//      operations of a wavefront execute in order).  Lanes are grouped by revolution (unwrapped entry index >> 9) and the
//      groups take turns.  G is {P, Q} pairs (one 8-byte RMW for both), XOR-swizzled by 32-entry block so that strides of 8, 16
//      entries (f0 = 250, 500 Hz) do not pile onto two banks.
//   2. Folding: the sine matrix's symmetries (s_k(T-1-i) = -s_k(i), s_k(T/2-1-i) = +-s_k(i) for odd / even k) bring G onto the
//      quarter range: Go / Ge for odd / even harmonics, split into fp16 hi / lo B-fragment planes, 16 columns = 8 frames x {P, Q}.
//   3. Wavefront w = (parity w & 1, harmonic tile w >> 1): 4 k-steps x 3 v_mfma_f32_16x16x32_f16 against its constant
//      A-fragments (32 registers, loaded once per block: blocks are persistent), the results through LDS to whole rows of
//      the P / Q workspace (or, DDSP_EXP_HARM_BWD=fused, into the chain rule of harmonic_bwd_chain.h without leaving LDS).
// Harmonics that cross Nyquist inside a frame ([kA, kN): the audio-rate mask of oscillator_bank, core.py:942-944) get the masked
// samples' contribution subtracted again, evaluated directly (one sine per sample and crossing harmonic, a wave reduction);
// harmonics >= kN are zero.  Frames the scheme does not cover (f0 < sr / 512: positions closer than one entry; f0 <= 0 or NaN;
// more than 8 crossing harmonics) take harm_bwd_pq_kernel's sum, lanes = harmonics, inside this kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "../../include/ddsp_amd.h"
#include "common.h"
#include "wavetable_coeffs.h"
#include "harm_table_frags.h"
#include "profile.h"
#include "harmonic_bwd_table.h"
#include "harmonic_bwd_chain.h"

namespace ddsp {

#ifndef DDSP_BT_MIN_WAVES
#define DDSP_BT_MIN_WAVES 4
#endif
constexpr int kBtT = 512;
constexpr int kBtFrames = 8;                 // frames = wavefronts per block
constexpr int kBtCol = 136;                  // halves per column of a folded plane: 128 + 8 (columns 68 dwords apart)
constexpr int kBtMaxCross = 8;
constexpr float kBtLoScale = 2048.0f;

typedef _Float16 bt_f16x8 __attribute__((ext_vector_type(8)));
typedef float bt_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bt_u32x4 __attribute__((ext_vector_type(4)));

struct BtArgs {
  const float* f0;             // [B, F]
  const double* theta0;        // [B, F]: phase at the start of frame j, revolutions, wrapped
  const float* g;              // [B, N]: dL / d audio
  float* pq;                   // P at 0, Q at q_offset: [B F, K] each
  size_t q_offset;
  const bt_u32x4* frags;       // [parity][harmonic tile][k-step][hi / lo][lane]
  int F, K, N, hop;
  long rows;                   // B F
  float sample_rate, nyquist;
  int amp_linear;
  // CHAIN: the frame-rate chain rule in the same launch (harmonic_bwd_chain.h); P and Q never leave LDS
  const float* amplitudes; const float* hd; float* grad_amp; float* grad_hd;
  BwdArgs chain;
  long long* dbg;              // per-phase clock stamps of block 0 (tools/exp_bwd_timeline.py: DDSP_EXP_BT_TIMELINE = a device pointer), or null
};

template <int W> struct BtPoly;
template <> struct BtPoly<6> {
  static constexpr int DE = kWtDegE6, DO = kWtDegO6;
  static constexpr float e(int p, int d) { return kWtE6[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO6[p * (DO + 1) + d]; }
};
template <> struct BtPoly<8> {
  static constexpr int DE = kWtDegE8, DO = kWtDegO8;
  static constexpr float e(int p, int d) { return kWtE8[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO8[p * (DO + 1) + d]; }
};

template <> struct BtPoly<10> {
  static constexpr int DE = kWtDegE10, DO = kWtDegO10;
  static constexpr float e(int p, int d) { return kWtE10[p * (DE + 1) + d]; }
  static constexpr float o(int p, int d) { return kWtO10[p * (DO + 1) + d]; }
};

// entry i of a frame's G at slot i ^ (block index of 32): a permutation inside every block of 32 entries
__device__ __forceinline__ int bt_swz(int i) { return i ^ ((i >> 5) & 15); }

__device__ __forceinline__ void bt_split(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * kBtLoScale);
}

// CHAIN: groups advance by SEVEN frames - wavefront 0's frame is the one before the group's (its Q is what the group's first
// frame adds to its P), computed here once more rather than fetched from another block
template <int W, bool CHAIN>
__global__ __launch_bounds__(W == 10 ? 1024 : 64 * kBtFrames, DDSP_BT_MIN_WAVES) void harm_bwd_table_kernel(BtArgs p) {
  // 129 .. 200 harmonics (W = 10: BASELINE configs[4]'s shapes): SIXTEEN wavefronts - eight of them spread a frame each as below,
  // fourteen take a (parity, harmonic tile) of the product: seven tiles of 16 per parity, each with its 32 registers of fragments
  constexpr int kMt = W == 10 ? 7 : 4, kMtAlloc = W == 10 ? 8 : 4;
  constexpr int kNch = W == 10 ? 4 : 2;                         // harmonics per lane where lanes = harmonics
  constexpr int kOutStride = W == 10 ? 212 : 132;
  constexpr int kStep = CHAIN ? kBtFrames - 1 : kBtFrames, kBack = CHAIN ? 1 : 0;
  int dbg_iter = 0;
#define DDSP_BT_STAMP(pt) do { if (p.dbg != nullptr && blockIdx.x == 0 && lane == 0 && dbg_iter < 16) p.dbg[(dbg_iter * 8 + wave) * 8 + (pt)] = (long long)__builtin_readcyclecounter(); } while (0)
  __shared__ __attribute__((aligned(16))) float2 s_g[kBtFrames][kBtT];            // 32 KB
  __shared__ __attribute__((aligned(16))) _Float16 s_b[2][2][16][kBtCol];         // [hi / lo][parity][column][n]: 17 KB
  __shared__ float s_corr[kBtFrames][2][kBtMaxCross];
  __shared__ int s_k[kBtFrames][4];                                               // kA, kN, direct, the exponent G was normalised by
  __shared__ float s_out[16][kOutStride];                                                // step 3's results: [column][harmonic]
  using C = BtPoly<W>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this wavefront's share of the constant factor for step 3: fetched ONCE - blocks are persistent (a block per group of
  // eight frames fetched its 64 KB from L2 every time: 256 MB per launch at batch 32, and with the 4-byte scattered stores of
  // the first version 38 of the kernel's 45 us - r05t)
  const int par = wave & 1, mt = wave >> 1;
  const bool has_task = mt < kMt;                               // (wave-uniform)
  const bool has_frame = wave < kBtFrames;
  bt_u32x4 afr[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int part = 0; part < 2; ++part)
      afr[ks][part] = p.frags[((((size_t)par * kMtAlloc + (has_task ? mt : 0)) * 4 + ks) * 2 + part) * 64 + lane];

  // the first tile's gradient samples of a group are requested a group ahead (a group is a chain of dependent steps with two
  // block barriers in it; the HBM latency at its head was a fifth of it)
  auto first_tile = [&](long r0) -> float {
    const long rw = r0 + wave - kBack;
    if (!has_frame || rw < 0 || rw >= p.rows || lane >= p.hop) return 0.0f;
    return p.g[(size_t)rw * p.hop + lane];           // row (b, j): sample b N + j hop + lane = row hop + lane
  };
  // ... and so are the frame's two f0 values and its phase: everything a group's first instruction depends on
  struct Head { float fj, fj1; double th0; };
  auto frame_head = [&](long r0) -> Head {
    const long rw = r0 + wave - kBack;
    Head h{0.0f, 0.0f, 0.0};
    if (has_frame && rw >= 0 && rw < p.rows) {
      const long jj = rw % p.F;
      h.fj = p.f0[rw];
      h.fj1 = p.f0[jj + 1 < p.F ? rw + 1 : rw];
      h.th0 = p.theta0[rw];
    }
    return h;
  };
  float g_next = first_tile((long)blockIdx.x * kStep);
  Head h_next = frame_head((long)blockIdx.x * kStep);
#pragma unroll 1
  for (long row0 = (long)blockIdx.x * kStep; row0 < p.rows; row0 += (long)gridDim.x * kStep) {
  const long row = row0 + wave - kBack;
  const bool row_ok = has_frame && row >= 0 && row < p.rows;
  const float g_first = g_next;
  const Head head = h_next;
  g_next = first_tile(row0 + (long)gridDim.x * kStep);
  h_next = frame_head(row0 + (long)gridDim.x * kStep);
  float dP[kNch], dQ[kNch];                                      // the plain sum's results (CHAIN: into LDS behind the barrier)
#pragma unroll
  for (int c = 0; c < kNch; ++c) { dP[c] = 0.0f; dQ[c] = 0.0f; }
  DDSP_BT_STAMP(0);

  // ---- 1. spreading -----------------------------------------------------------------------------------------------------------
  if (has_frame) {
    float4* z4 = reinterpret_cast<float4*>(&s_g[wave][0]);
    for (int k = lane; k < kBtT / 2; k += 64) z4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // (a wavefront's LDS operations execute in order, its lanes in lockstep: the wave barriers here and below generate no
  // instruction - they say so to the compiler, and to the CPU emulation of tests/hip_emu, whose lanes run one after another)
  __builtin_amdgcn_wave_barrier();
  int kA = 0, kN = 0, direct = 1;
  if (row_ok) {
    const int b = (int)(row / p.F), j = (int)(row - (long)b * p.F);
    const float* __restrict__ f0 = p.f0 + (size_t)b * p.F;
    const float fj = head.fj, fj1 = head.fj1;
    const float fmx = fmaxf(fj, fj1), fmn = fminf(fj, fj1);
    kA = p.K; kN = p.K;
    if (fmx > 0.0f) kA = (int)fminf((float)p.K, floorf(p.nyquist * (1.0f - 2e-6f) / fmx));
    if (fmn > 0.0f) kN = (int)fminf((float)p.K, floorf(p.nyquist * (1.0f + 2e-6f) / fmn));
    kA = max(min(kA, kN), 0);
    // positions more than one entry apart - by a margin: the increment of the fp64 position is f(n) / sr * 512 with f between the
    // two frames' f0, and two samples on one entry in one read-modify-write would lose one of them - (and finite, and a bounded
    // number of revolutions per tile): else the plain sum
    const float stride_min = fmn * ((float)kBtT / p.sample_rate);
    direct = (stride_min >= 1.001f && fmx <= 0.5f * p.sample_rate && kN - kA <= kBtMaxCross) ? 0 : 1;
    const double inv_sr = 1.0 / (double)p.sample_rate, inv_2hop = 0.5 / (double)p.hop;
    const float inv_hop = 1.0f / (float)p.hop;
    const double wj = (double)fj * inv_sr, dw = ((double)fj1 - (double)fj) * inv_sr * inv_2hop;
    const double th0 = head.th0;
    const float* __restrict__ g = p.g + (size_t)b * p.N + (size_t)j * p.hop;
    float2* const G = &s_g[has_frame ? wave : 0][0];
    // per-sample values of a tile: lanes = samples
    auto sample = [&](int t0, double& cyc, float& lerp, float& c_cur, float& c_next, bool& live) {
      const int r = t0 + lane;
      live = r < p.hop;
      // (lanes past the frame's last sample - frame sizes that are not multiples of 64 - carry that sample's phase: the ramp
      // extrapolated past the frame can run backwards when f0 falls steeply)
      const double rr = (double)min(r, p.hop - 1);
      cyc = th0 + (rr + 1.0) * (wj + dw * rr);
      lerp = (float)r * inv_hop;
      const float w_next = p.amp_linear ? lerp : 0.5f - 0.5f * __builtin_amdgcn_cosf(0.5f * lerp);
      const float gv = t0 == 0 ? g_first : (live ? g[r] : 0.0f);
      c_cur = (1.0f - w_next) * gv;
      c_next = w_next * gv;
    };
#ifdef DDSP_BT_NO_SPREAD
    if (p.K == 12345)
#endif
    if (!direct) {
      if (lane < 2 * kBtMaxCross) s_corr[wave][lane >> 3][lane & 7] = 0.0f;
#pragma unroll 1
      for (int t0 = 0; t0 < p.hop; t0 += 64) {
        double cyc;
        float lerp, c_cur, c_next;
        bool live;
        sample(t0, cyc, lerp, c_cur, c_next, live);
        // unwrapped table position: entry index i_abs, offset z in [-1/2, 1/2) from the middle between entries i and i + 1
        const double pu = fma(cyc, (double)kBtT, -0.5);
        const double flu = floor(pu);
        const float z = (float)(pu - flu) - 0.5f;
        const int i_abs = (int)flu;
        const int i0 = i_abs & (kBtT - 1);
        const int rev_abs = i_abs >> 9;
        // the revolutions the tile's LIVE samples span: positions grow along the frame, so the last live lane holds the last
        // (lane 63 of a frame's last tile is past the frame when hop % 64 != 0: taken from there, live lanes' turns were
        // skipped and their share of G dropped - ADVICE r4)
        const int last_live = min(63, p.hop - 1 - t0);
        int rev = rev_abs - __builtin_amdgcn_readfirstlane(rev_abs);
        int rev_last = __builtin_amdgcn_readlane(rev, last_live);
        // A tile whose live samples span fewer than T entries takes ONE turn even when it crosses the end of the table: the
        // entries (i_abs + tap offset) mod T of a fixed tap number are distinct as long as i_abs grows by >= 1 per lane and by
        // less than T over the tile, whatever revolution they fall in.  At 70 Hz a tile spans 143 entries, 28 % of the tiles wrap,
        // and with eight frames per group nearly every group waited for a wavefront that ran the W read-modify-writes twice
        // (round 6).
#ifndef DDSP_EXP_BT_TWO_TURNS         // (the A/B switch of tools/exp_bwd.py)
        if (__builtin_amdgcn_readlane(i_abs, last_live) - __builtin_amdgcn_readfirstlane(i_abs) < kBtT) { rev = 0; rev_last = 0; }
#endif
        const float zz = z * z;
        float w_lo[W / 2], w_hi[W / 2];
#pragma unroll
        for (int pr = 0; pr < W / 2; ++pr) {
          float e = C::e(pr, C::DE), o = C::o(pr, C::DO);
#pragma unroll
          for (int d = C::DE - 1; d >= 0; --d) e = fmaf(e, zz, C::e(pr, d));
#pragma unroll
          for (int d = C::DO - 1; d >= 0; --d) o = fmaf(o, zz, C::o(pr, d));
          w_lo[pr] = fmaf(z, o, e);          // entry i0 - pr
          w_hi[pr] = fmaf(-z, o, e);         // entry i0 + 1 + pr
        }
#pragma unroll 1
        for (int rv = 0; rv <= rev_last; ++rv) {
#ifndef DDSP_BT_NO_RMW
          if (rev == rv && live) {
            // W read-modify-writes, one tap number at a time: within a tap number the lanes of a revolution hit distinct entries
#pragma unroll
            for (int tn = 0; tn < W; ++tn) {
              const int pr = tn < W / 2 ? W / 2 - 1 - tn : tn - W / 2;
              const float w = tn < W / 2 ? w_lo[pr] : w_hi[pr];
              const int idx = tn < W / 2 ? i0 - pr : i0 + 1 + pr;
              float2* e = G + bt_swz(idx & (kBtT - 1));
              float2 v = *e;
              v.x = fmaf(c_cur, w, v.x);
              v.y = fmaf(c_next, w, v.y);
              *e = v;
            }
          }
#endif
        }
        // harmonics that cross Nyquist in this frame: what their masked samples put into G comes out again
        if (kN > kA) {
          const float th = (float)(cyc - floor(cyc));
          for (int c = 0; c < kN - kA; ++c) {
            const float kf = (float)(kA + c + 1);
            const float top = fj * kf, bot = fj1 * kf;
            const float fk = rn_add(top, rn_mul(rn_sub(bot, top), lerp));      // TF's fp32 op order (core.py:942-944)
            const float s = fk >= p.nyquist ? sin_rev(th * kf) : 0.0f;
            const float sp = wave_sum_dpp(c_cur * s), sq = wave_sum_dpp(c_next * s);
            if (lane == 0) { s_corr[wave][0][c] += sp; s_corr[wave][1][c] += sq; }
          }
        }
      }
    } else {
      // the plain sum (harm_bwd_pq_kernel's), lanes = harmonics lane and lane + 64; a tile's per-sample values come from the
      // lanes that hold them.  (A loop of its own: its accumulators then share registers with the spreading's.)
      float Pd[kNch], Qd[kNch], kf[kNch], top[kNch], bot[kNch];
#pragma unroll
      for (int c = 0; c < kNch; ++c) {
        Pd[c] = 0.0f; Qd[c] = 0.0f;
        kf[c] = (float)(lane + 64 * c + 1);
        top[c] = fj * kf[c]; bot[c] = fj1 * kf[c];
      }
#pragma unroll 1
      for (int t0 = 0; t0 < p.hop; t0 += 64) {
        double cyc;
        float lerp, c_cur, c_next;
        bool live;
        sample(t0, cyc, lerp, c_cur, c_next, live);
        const float th = (float)(cyc - floor(cyc));
        const int n_live = min(64, p.hop - t0);
#pragma unroll 1
        for (int s_ = 0; s_ < n_live; ++s_) {
          const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_cur), s_));
          const float cn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_next), s_));
          const float ths = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, th), s_));
          const float lp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lerp), s_));
#pragma unroll
          for (int c = 0; c < kNch; ++c) {
            const float fk = rn_add(top[c], rn_mul(rn_sub(bot[c], top[c]), lp));
            const float sv = (fk >= p.nyquist || lane + 64 * c >= kN) ? 0.0f : sin_rev(ths * kf[c]);
            Pd[c] = fmaf(cc, sv, Pd[c]);
            Qd[c] = fmaf(cn, sv, Qd[c]);
          }
        }
      }
      if constexpr (CHAIN) {
#pragma unroll
        for (int c = 0; c < kNch; ++c) { dP[c] = Pd[c]; dQ[c] = Qd[c]; }
      } else {
        const size_t at = (size_t)row * p.K;
#pragma unroll
        for (int c = 0; c < kNch; ++c)
          if (lane + 64 * c < p.K) { p.pq[at + lane + 64 * c] = Pd[c]; p.pq[p.q_offset + at + lane + 64 * c] = Qd[c]; }
      }
    }
  }
  if (has_frame && lane == 0) { s_k[wave][0] = kA; s_k[wave][1] = kN; s_k[wave][2] = direct; }
  __builtin_amdgcn_wave_barrier();
  DDSP_BT_STAMP(1);

  // ---- 2. folding onto the quarter range, split, B-fragment planes: columns 2 w (P) and 2 w + 1 (Q) -----------------------------
#ifndef DDSP_BT_NO_FOLD
  if (has_frame) {
    const float2* G = &s_g[wave][0];
    float v[2][2][2];                                                           // [half][parity: odd k, even k][P, Q]
    float mx = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = lane + 64 * h;
      const float2 g0 = G[bt_swz(n)], g1 = G[bt_swz(kBtT - 1 - n)], g2 = G[bt_swz(kBtT / 2 - 1 - n)], g3 = G[bt_swz(kBtT / 2 + n)];
      const float ax = g0.x - g1.x, ay = g0.y - g1.y, bx = g2.x - g3.x, by = g2.y - g3.y;
      v[h][0][0] = ax + bx; v[h][0][1] = ay + by; v[h][1][0] = ax - bx; v[h][1][1] = ay - by;
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[h][0][0]), fabsf(v[h][0][1]))), fmaxf(fabsf(v[h][1][0]), fabsf(v[h][1][1])));
    }
    // G carries the scale of dL/d audio, whatever it is (loss scaling, sum-reduced losses, vanishing gradients): the frame's
    // four columns are brought to [1/2, 1) by a power of two before the fp16 hi / lo split and the products taken back by it
    // in step 3 - the result is as scale invariant as the plain sum's (3e4 overflowed, 1e-10 lost 7 %)
    const int ge = pow2_exponent(wave_max_nonneg_dpp(mx));
    if (lane == 0) s_k[wave][3] = ge;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = lane + 64 * h;
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          _Float16 hi, lo;
          bt_split(ldexpf(v[h][pa][q], -ge), hi, lo);
          s_b[0][pa][2 * wave + q][n] = hi;
          s_b[1][pa][2 * wave + q][n] = lo;
        }
    }
  }
#endif
  DDSP_BT_STAMP(2);
  __syncthreads();
  DDSP_BT_STAMP(3);

  // ---- 3. D[harmonic][column] = sum_n A[harmonic][n] G_folded[n][column] ------------------------------------------------------------
#ifdef DDSP_BT_NO_P3
  if (p.K == 12345)
#endif
  if (has_task) {
    const int i16 = lane & 15, g4 = lane >> 4;
    bt_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_hl = {0.f, 0.f, 0.f, 0.f}, acc_lh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bt_f16x8 bh = *reinterpret_cast<const bt_f16x8*>(&s_b[0][par][i16][32 * ks + 8 * g4]);
      const bt_f16x8 bl = *reinterpret_cast<const bt_f16x8*>(&s_b[1][par][i16][32 * ks + 8 * g4]);
      const bt_f16x8 ah = __builtin_bit_cast(bt_f16x8, afr[ks][0]), al = __builtin_bit_cast(bt_f16x8, afr[ks][1]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
      acc_hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_hl, 0, 0, 0);
      acc_lh = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc_lh, 0, 0, 0);
    }
    // D[row 4 g + r = harmonic k' of the tile][column i16 = 2 frame + quantity] -> rows of the P / Q workspace through LDS
    // (straight from the accumulators every lane stored 4 bytes into a cache line of its own)
    const int fr = i16 >> 1, q = i16 & 1;
    const int fkA = s_k[fr][0], fkN = s_k[fr][1], fdirect = s_k[fr][2], fge = s_k[fr][3];
    if (!fdirect) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k0 = 2 * (16 * mt + 4 * g4 + r) + par;                 // 0-based harmonic index (harmonic k0 + 1)
        float v = ldexpf(acc[r] + (acc_hl[r] + acc_lh[r]) * (1.0f / kBtLoScale), fge);
        if (k0 >= fkN) v = 0.0f;
        else if (k0 >= fkA) v -= s_corr[fr][q][k0 - fkA];
        if (k0 < kOutStride) s_out[i16][k0] = v;
      }
    }
  }
  DDSP_BT_STAMP(4);
  if constexpr (CHAIN) {
    if (row_ok && s_k[wave][2]) {          // (s_out's readers of the previous group are behind the barrier above)
#pragma unroll
      for (int c = 0; c < kNch; ++c) { s_out[2 * wave][lane + 64 * c] = dP[c]; s_out[2 * wave + 1][lane + 64 * c] = dQ[c]; }
    }
  }
  __syncthreads();
  DDSP_BT_STAMP(5);
  if constexpr (CHAIN) {
    // the chain rule of frame `row` (wavefronts 1 .. 7): dL/da = P[row] + Q[row - 1] (wavefront w - 1's, same batch row) (+ Q[row]
    // for the held last frame)
    if (wave >= 1 && row_ok) {
      const int j = (int)(row % p.F);
      const float* P = s_out[2 * wave];
      const float* Qp = s_out[2 * wave - 1];
      const float* Qo = s_out[2 * wave + 1];
      harm_chain_row<kNch>(lane, row, j, p.amplitudes, p.hd, p.f0, p.grad_amp, p.grad_hd, p.chain, [&](int k) {
        float gsum = P[k];
        if (j > 0) gsum += Qp[k];
        if (j == p.F - 1) gsum += Qo[k];
        return gsum;
      }, [](float v) { return v; });
    }
  } else
#ifdef DDSP_BT_NO_STORE
  if (p.K == 12345)
#endif
  if (row_ok && !s_k[wave][2]) {
    // wavefront w writes its own frame's two rows
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float* dst = p.pq + (q ? p.q_offset : 0) + (size_t)row * p.K;
#pragma unroll
      for (int c = 0; c < kNch; ++c)
        if (lane + 64 * c < p.K) dst[lane + 64 * c] = s_out[2 * wave + q][lane + 64 * c];
    }
  }
  DDSP_BT_STAMP(6);
  ++dbg_iter;
  }    // the block's next group of frames
#undef DDSP_BT_STAMP
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
// the transposed constant factor: A[k'][n] = sin(2 pi k (n + 1/2) / T) / psi_hat(k / T), k = 2 k' + 1 + parity, as A-fragments:
// [parity][harmonic tile of 16][k-step of 32 points][hi / lo][lane (i = k' & 15, g)][8 halves: n = 32 ks + 8 g + e]
static const bt_u32x4* bt_fragments(int W) {
  static std::mutex mu;
  static const bt_u32x4* cache[3][16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  const int wi = W == 6 ? 0 : (W == 8 ? 1 : 2);
  const int MT = W == 10 ? 8 : 4;                                // harmonic tiles of 16 per parity
  std::lock_guard<std::mutex> lock(mu);
  if (cache[wi][dev]) return cache[wi][dev];
  const float* invpsi = W == 6 ? kWtInvPsi6_T512 : (W == 8 ? kWtInvPsi8_T512 : kWtInvPsi10_T512);
  const int n_invpsi = W == 6 ? (int)(sizeof(kWtInvPsi6_T512) / sizeof(float))
                     : (W == 8 ? (int)(sizeof(kWtInvPsi8_T512) / sizeof(float)) : (int)(sizeof(kWtInvPsi10_T512) / sizeof(float)));
  std::vector<uint32_t> f((size_t)2 * MT * 4 * 2 * 64 * 4, 0u);
  for (int pa = 0; pa < 2; ++pa)
    for (int mt = 0; mt < MT; ++mt)
      for (int ks = 0; ks < 4; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int d = 0; d < 4; ++d) {
            uint32_t hi2 = 0, lo2 = 0;
            for (int h = 0; h < 2; ++h) {
              const int kp = 16 * mt + (lane & 15), n = 32 * ks + 8 * (lane >> 4) + 2 * d + h;
              unsigned short hb, lb;
              wt_frag_element(kBtT, n, 2 * kp + 1 + pa, invpsi, n_invpsi, &hb, &lb);
              hi2 |= (uint32_t)hb << (16 * h);
              lo2 |= (uint32_t)lb << (16 * h);
            }
            const size_t at = ((((size_t)pa * MT + mt) * 4 + ks) * 2) * 64 + lane;
            f[(at + 0) * 4 + d] = hi2;
            f[(at + 64) * 4 + d] = lo2;
          }
  void* dptr = nullptr;
  if (hipMalloc(&dptr, f.size() * 4) != hipSuccess) return nullptr;
  if (hipMemcpy(dptr, f.data(), f.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(dptr);
    return nullptr;
  }
  cache[wi][dev] = (const bt_u32x4*)dptr;
  return cache[wi][dev];
}

int harm_bwd_table_prepare(int K) { return bt_fragments(K <= 100 ? 6 : (K <= 128 ? 8 : 10)) ? 0 : 1; }

bool harm_bwd_table_ok(int F, int K, int N) {
  static const bool off = [] { const char* e = getenv("DDSP_EXP_HARM_BWD"); return e && e[0] == 'p'; }();   // "plain": the sums
  const int hop = F > 0 ? N / F : 0;
  return !off && K >= 1 && K <= 200 && hop >= 1 && (long)F * hop == N && hop <= 4096;
}

int launch_harm_bwd_table(const float* f0_hz, const double* theta0, const float* grad_audio, float* pq, size_t q_offset, int B,
                          int F, int K, int N, int sample_rate, int amp_linear, hipStream_t st, const float* amplitudes,
                          const float* hd, float* grad_amp, float* grad_hd, unsigned flags, int inputs_are_controls) {
  // The chain rule in the same launch (CHAIN) is correct and is NOT the default: it puts seven more dependent steps into every
  // group's chain and recomputes a frame in eight - 60.1 us against 41.6 + 21.2 in two launches at batch 32, 215 against 135 + 66
  // at batch 128 (profiles/r05_harm_bwd_table.txt).  DDSP_EXP_HARM_BWD=fused runs it.
  static const bool fused = [] { const char* e = getenv("DDSP_EXP_HARM_BWD"); return e && e[0] == 'f'; }();
  const bool chain = amplitudes != nullptr && fused;
  const int W = K <= 100 ? 6 : (K <= 128 ? 8 : 10);
  const bt_u32x4* frags = bt_fragments(W);
  if (!frags) return DDSP_ERR_LAUNCH;
  BtArgs a;
  a.f0 = f0_hz; a.theta0 = theta0; a.g = grad_audio; a.pq = pq; a.q_offset = q_offset; a.frags = frags;
  a.F = F; a.K = K; a.N = N; a.hop = N / F;
  a.rows = (long)B * F;
  a.sample_rate = (float)sample_rate; a.nyquist = (float)(sample_rate / 2.0);
  a.amp_linear = amp_linear;
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  a.dbg = nullptr;
#ifdef DDSP_BT_TIMELINE      // tools/exp_bwd_timeline.py's build only (the product does not take raw pointers from the environment)
  { const char* e = getenv("DDSP_EXP_BT_TIMELINE"); a.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  a.amplitudes = amplitudes; a.hd = hd; a.grad_amp = grad_amp; a.grad_hd = grad_hd;
  a.chain.F = F; a.chain.K = K; a.chain.N = N; a.chain.hop = a.hop;
  a.chain.sample_rate = a.sample_rate; a.chain.nyquist = a.nyquist; a.chain.amp_linear = amp_linear;
  a.chain.flags = flags; a.chain.inputs_are_controls = inputs_are_controls;
  const int step = chain ? kBtFrames - 1 : kBtFrames;
  const long groups = (a.rows + step - 1) / step;
  const dim3 grid((unsigned)std::min<long>(groups, 2L * n_cu));       // persistent: two blocks per CU (59 KB of LDS, 128 registers)
  if (chain) {
    if (W == 6) hipLaunchKernelGGL((harm_bwd_table_kernel<6, true>), grid, dim3(64 * kBtFrames), 0, st, a);
    else if (W == 8) hipLaunchKernelGGL((harm_bwd_table_kernel<8, true>), grid, dim3(64 * kBtFrames), 0, st, a);
    else hipLaunchKernelGGL((harm_bwd_table_kernel<10, true>), dim3(std::min<unsigned>(grid.x, (unsigned)n_cu)), dim3(1024), 0, st, a);
  } else {
    if (W == 6) hipLaunchKernelGGL((harm_bwd_table_kernel<6, false>), grid, dim3(64 * kBtFrames), 0, st, a);
    else if (W == 8) hipLaunchKernelGGL((harm_bwd_table_kernel<8, false>), grid, dim3(64 * kBtFrames), 0, st, a);
    else hipLaunchKernelGGL((harm_bwd_table_kernel<10, false>), dim3(std::min<unsigned>(grid.x, (unsigned)n_cu)), dim3(1024), 0, st, a);   // (one block of sixteen wavefronts per CU)
  }
  return hipGetLastError() == hipSuccess ? (chain ? 1 : DDSP_OK) : DDSP_ERR_LAUNCH;       // 1: the chain rule is done too
}

}  // namespace ddsp
