// Long-impulse-response convolution for gfx950 (MI355X): effects.Reverb (ddsp/effects.py:27-117) and
// the single-frame case of core.fft_convolve (ddsp/core.py:1382-1473).
//
// The reference zero-pads audio [B,N] and one IR [B,L] (L = 48 000 by default) to one FFT of
// 131 072 points.  Here the same linear convolution is evaluated as a uniformly partitioned
// overlap-save convolution whose FFTs (8192 points, 64 KB of float2) live entirely in LDS:
//
//   x blocks   X_j = FFT(x[(j-1)P .. (j+1)P)),  P = 4096, j = 0 .. nb-1
//   IR parts   H_p = FFT([h[pP .. (p+1)P), 0 ... 0]),   p = 0 .. np-1        (rv_fft_kernel)
//   per bin    Y_j = sum_{p <= j} X_{j-p} H_p
//   y[jP + i] = IFFT(Y_j)[P + i],  out[n] = y[n + delay] (+ audio[n])
//
// x and h are real, so two x blocks ride in one complex FFT: Z_m = FFT(x_{m-1} + i x_m) = X_{m-1} + i X_m
// (rv_fft_kernel).  The sum is linear in Z, so W_m = sum_p Z_{m-p} H_p = Y_{m-1} + i Y_m needs no
// untangling: it is formed for odd m only - a sliding window over m in registers, every spectrum
// read once, W_m overwriting Z_m in place (rv_mac_kernel) - and one inverse FFT returns output block
// m-1 in its real part and block m in its imaginary part (rv_ifft_kernel): half the inverse FFTs and
// half the multiply-adds of the plain scheme.
//
// Round 5, one impulse response for the whole batch (the trainable Reverb of the shipped configurations: effects.py:62-80,
// gin/models/solo_instrument.gin:26-40): two ROWS ride in one complex transform instead of two blocks of one row,
// Z_j = X_j(row a) + i X_j(row b); Z_j H = Y_j(a) + i Y_j(b) for every j because H is the spectrum of a real response both rows
// share.  Every block is then transformed ONCE (above, each is the imaginary part of one spectrum and the real part of the
// next), the multiply-add pass reads half the spectra, and the audio's blocks and the impulse response's partitions are one
// launch (the partitions' launch ahead of the audio's was 15 - 19 us of latency at any batch size): 126 -> 100 us at batch 128,
// 50 -> 44 at batch 32 (profiles/r05_reverb_row_pairs_and_the_fused_experiment.txt, which also has the form that keeps every
// spectrum on chip - bin classes with the ring of spectra in registers - built, measured at 99 us and taken out again).  Later in
// the round: the transform blocks persistent (they fetch the next input ahead, first / last pass fused with load / store: 91 us),
// and no loop waits for its own stores any more (the multiply-add pass's full passes are straight-line code, the dry signal
// is fetched at clamped indices: 80 - 85 us of kernels; profiles/r05q_*, r05u_*, r05v_*).
//
// The forward transform is an in-place radix-8 (+ one radix-2 stage) decimation-in-frequency FFT that leaves its bins in
// digit-reversed order; the inverse undoes it stage by stage: the per-bin products do not care
// about the order, so no reordering pass exists anywhere.  Twiddles come from
// v_sin_f32 / v_cos_f32 on exact binary fractions of a revolution (abs. error 1.2e-7,
// profiles/r01_microbench_alu.txt).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <cstdlib>
#include "common.h"
#include "profile.h"
#include "fft_radix8.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {
constexpr unsigned DDSP_CONV_EXP_PLAIN_ORDER = 1u << 30;      // internal: rv_fft_kernel deals its items in block order


constexpr int kRvP = 4096;             // output samples per block = taps per IR partition
constexpr int kRvN = 2 * kRvP;         // FFT size
constexpr int kRvMaxParts = 16;        // IR partitions held in registers by the MAC kernel
constexpr int kRvThreads = 1024;      // FFT blocks: 16 wavefronts on one 64 KB LDS array (one radix-8 butterfly per lane per
                                       // stage): with 4 wavefronts a stage took 1.3 us - one wavefront per SIMD cannot overlap
                                       // its own LDS and VALU phases (tools/microbench5)
constexpr int kRvMacThreads = 256;
// LDS layout of the FFT array: 2 float2 of padding after every 16, so that the small-stride stages
// (4 lanes per 16-element chunk, chunks 128 B apart) do not land 8-16 lanes on the same banks
// (this took the SpectralLoss FFTs from 234 to 165 us).  Pairs (2i, 2i+1) stay adjacent and 16-byte aligned.
constexpr int kRvStore = kRvN + kRvN / 8;
__device__ __forceinline__ int RP(int i) { return i + ((i >> 4) << 1); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// In place, 1024 threads, one barrier per stage.  Forward (decimation in frequency, kernel exp(-2 pi i nk/N)) leaves the bins in a
// digit-reversed order; the inverse is the exact algebraic inverse of the forward pipeline - the
// stages undone one by one in reverse order (conjugate twiddles, conjugate 4-point DFT, factor 1/N
// left to the caller) - so it accepts that order and returns natural order, whatever the order is.
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {          // a * conj(b)
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// 8192 = 8^4 * 2: four radix-8 stages (one butterfly per thread: its eight loads are issued before any arithmetic and its
// stores after it - the butterflies of a stage touch disjoint elements, which the compiler cannot see through the LDS
// indices) and the radix-2 stage on neighbours: five passes and five barriers where radix 4 took seven (rounds 1-3; the
// SpectralLoss transforms went the same way: csrc/fft_radix8.h).  Forward (decimation in frequency, kernel exp(-2 pi i nk/N))
// leaves the bins in a digit-reversed order; the inverse is the exact algebraic inverse of the forward pipeline - the stages
// undone one by one in reverse order (conjugate twiddles, conjugate 8-point DFT, factor 1/N applied at the output) - so it
// accepts that order and returns natural order.  Both live inside the two loop kernels below.
constexpr int kRvPairs = kRvN / 2 / kRvThreads;                // float4 pairs per thread in the radix-2 stage (4)
static_assert(kRvN / 8 == kRvThreads, "one radix-8 butterfly per thread");

struct RvArgs {
  int N, L, n_out, nb, np, delay;      // n_out: samples written per row (out[n] = y[n + delay], n < n_out)
  unsigned flags;
  int ir_batch;                         // 1: one IR for every batch row
  // ROW PAIRS (round 5; one impulse response for the whole batch, the trainable Reverb of the shipped configurations): two ROWS
  // ride in one complex transform - Z_j = X_j(row a) + i X_j(row b), block j of both - instead of two blocks of one row.  Z H is
  // then Y_j(a) + i Y_j(b) for EVERY j because H is the spectrum of a real response both rows share: half the forward
  // transforms (each block is transformed once, not as the real part of one spectrum and the imaginary part of the next) and
  // half the spectra written and read by the multiply-add pass.  B: rows (an odd batch's last pair holds one).
  int pairs, B;
  // the first output spectrum any kept sample lies in (delay > 0: the correlations of the backward pass keep samples from
  // `delay` on - dL/d ir starts N - 1 samples into the convolution, half of its blocks are never looked at)
  int m_first;
};

// One thread per pair of bins (16-byte accesses).  The IR spectra of all partitions sit in
// registers, the Z spectra slide through a register window, W_m replaces Z_m (odd m; every m for row pairs) in memory.
// (Round 5 measured the spectra requested three blocks ahead of their use - the loop is a load, its wait and the products, sixteen
// times -: 35.6 -> 42.1 us at batch 128, twelve more registers at three wavefronts per SIMD cost more than the waits;
// profiles/r05f_reverb_three_ways.txt.)
__global__ __launch_bounds__(kRvMacThreads) void rv_mac_kernel(float4* __restrict__ xspec,
                                                            const float4* __restrict__ hspec, RvArgs p) {
  const int idx = blockIdx.x * kRvMacThreads + threadIdx.x;      // < kRvN / 2
  const int b = blockIdx.y;
  const float4* __restrict__ hb = hspec + (size_t)(p.ir_batch == 1 ? 0 : b) * p.np * (kRvN / 2) + idx;
  float4* __restrict__ xb = xspec + (size_t)b * p.nb * (kRvN / 2) + idx;
  float4 h[kRvMaxParts], w[kRvMaxParts];
#pragma unroll
  for (int q = 0; q < kRvMaxParts; ++q) {
    h[q] = (q < p.np) ? hb[(size_t)q * (kRvN / 2)] : make_float4(0.f, 0.f, 0.f, 0.f);
    w[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int j = 0; j < p.nb; ++j) {
#pragma unroll
    for (int q = kRvMaxParts - 1; q > 0; --q) w[q] = w[q - 1];
    w[0] = xb[(size_t)j * (kRvN / 2)];
    if ((p.pairs == 0 && (j & 1) == 0) || j < p.m_first) continue;      // W_m only for odd m (blocks m-1 and m); row pairs: every m; none below the first kept sample
    float2 y0 = make_float2(0.f, 0.f), y1 = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < kRvMaxParts; ++q) {
      if (q < p.np) {                                          // wave-uniform
        const float2 a0 = cmul(make_float2(w[q].x, w[q].y), make_float2(h[q].x, h[q].y));
        const float2 a1 = cmul(make_float2(w[q].z, w[q].w), make_float2(h[q].z, h[q].w));
        y0.x += a0.x; y0.y += a0.y; y1.x += a1.x; y1.y += a1.y;
      }
    }
    xb[(size_t)j * (kRvN / 2)] = make_float4(y0.x, y0.y, y1.x, y1.y);
  }
}

// ---- the transform kernels: persistent blocks ---------------------------------------------------------------------------------
// A block LOOPS over transforms (two blocks of 16 wavefronts per CU, 64 registers) and requests the next one's input before it
// starts on the current one.  The first pass of the forward transform (q = 1024: elements tid + 1024 m) works on those
// registers and the last (radix 2 on neighbours) goes straight from LDS to the spectrum in memory: 8 instead of 12 passes over
// the LDS array per transform; the inverse likewise (radix 2 on the way in, the last radix-8 pass straight to the output rows:
// only its outputs 4 .. 7 - the second half of the block, what overlap-save keeps - become samples).  The audio's blocks and
// the impulse responses' partitions are items of one launch (as two launches the second waited for the first's twelve blocks:
// 15 - 19 us at any batch size, profiles/r05e).  Until late in round 5 a block was ONE transform - load, five passes, store,
// both blocks of a CU in the same phase at the same time -: 34.9 / 33.5 us per launch at batch 128 against 30.4 / 32.0 now
// (profiles/r05q_reverb_persistent_ab.txt); what is left is the sum of LDS time and ALU time of the passes, which overlap
// poorly behind their barriers.  The index t the addresses and twiddles are made from is laundered once per transform
// (DDSP_KEEP_IN_VGPR): they are functions of the thread index alone, and the compiler otherwise makes them ahead of the loop,
// keeps sixty registers of them alive, spills, and waits for the prefetch at every reload.
struct RvItem { int j, b, ir; };
__device__ __forceinline__ RvItem rv_item(int w, int n_ir, const RvArgs& p) {
  RvItem it;
  if (w < n_ir) { it.ir = 1; it.b = w / p.np; it.j = w - it.b * p.np; }
  else { const int a = w - n_ir; it.ir = 0; it.b = a / p.nb; it.j = a - it.b * p.nb; }
  return it;
}

// elements tid + 1024 m (m = 0 .. 7) of the transform's input: one path for the three kinds of
// input - an IR partition (P taps, then zeros; no imaginary part), two blocks of one row (the imaginary part is the row one
// block later), block j of a pair of rows
__device__ __forceinline__ void rv_fetch(float2 (&v)[8], const float* __restrict__ audio, const float* __restrict__ ir,
                                         const RvArgs& p, RvItem it, int tid) {
  const bool is_ir = it.ir != 0;
  const int len = is_ir ? p.L : p.N;
  const bool pair_mode = !is_ir && p.pairs > 0;
  const float* __restrict__ row = (is_ir ? ir : audio) + (size_t)(pair_mode ? 2 * it.b : it.b) * len;
  const float* __restrict__ row_im = is_ir ? nullptr : (pair_mode ? (2 * it.b + 1 < p.B ? row + len : nullptr) : row);
  const int base = is_ir ? it.j * kRvP : (pair_mode ? (it.j - 1) * kRvP : (it.j - 2) * kRvP);
  const int shift_im = pair_mode ? 0 : kRvP;
  const int hi = is_ir ? min(len, base + kRvP) : len;            // (a partition ends after P taps)
  const bool rev = (p.flags & (is_ir ? DDSP_CONV_REVERSE_IR : DDSP_CONV_REVERSE_AUDIO)) != 0;
  // effects.Reverb._mask_dry_ir (effects.py:50-60): tap 0 carries the dry signal -> 0
  const int lo = (is_ir && (p.flags & DDSP_CONV_MASK_TAP0)) ? 1 : 0;
  const int g0 = base + tid;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int g = g0 + kRvThreads * m, gi = g + shift_im;
    float re = 0.0f, im = 0.0f;
    if (g >= lo && g < hi) re = row[rev ? len - 1 - g : g];
    if (row_im != nullptr && gi >= 0 && gi < hi) im = row_im[rev ? len - 1 - gi : gi];
    v[m] = make_float2(re, im);
  }
}

__global__ __launch_bounds__(kRvThreads, 8) void rv_fft_kernel(const float* __restrict__ audio, const float* __restrict__ ir,
                                                                 float2* __restrict__ xspec, float2* __restrict__ hspec, RvArgs p,
                                                                 int n_ir, int n_items) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int tid = threadIdx.x;
  // Which items a block takes: consecutive blocks of one row share half their input (overlap-save: block j is the row's samples
  // (j - 1) P .. (j + 1) P), and block i of the grid runs on XCD i % 8, each with an L2 of its own - dealt out in the order of
  // the block index the two readers of a sample sat on different XCDs and the L2s fetched the audio twice (FETCH_SIZE 63 MB
  // a launch at batch 128 against 33 MB of input, profiles/r05v_*).  Of every ROUND of gridDim.x items (the last one: of what
  // is left, so that every XCD keeps its share of it) an XCD therefore takes a contiguous eighth, its blocks in order.
  // (A permutation of the items within a round: which block computes a spectrum is all that changes.)
  // Measured (profiles/r05z_reverb_xcd_item_order.txt): with one impulse response for the batch (row pairs: the audio's blocks
  // are all but twelve of the items) 30.9 -> 29.9 us a launch at batch 128; with an impulse response per row - a third of the
  // items are partitions, which read half as much and overlap with nothing - an XCD's share of a round is all partitions or all
  // audio and the launch is 0.8 us slower at batch 32, no faster at 128: those keep the block order.
  const bool xcd_order = (gridDim.x & 7) == 0 && p.pairs > 0 && !(p.flags & DDSP_CONV_EXP_PLAIN_ORDER);
  int round_base = 0;
  auto item_of_round = [&]() -> int {              // this block's item of the round that starts at round_base, or n_items
    if (!xcd_order) return round_base + (int)blockIdx.x < n_items ? round_base + (int)blockIdx.x : n_items;
    const int size = min((int)gridDim.x, n_items - round_base);
    if (size <= 0) return n_items;
    const int x = (int)(blockIdx.x & 7), i = (int)(blockIdx.x >> 3);
    const int lo = (x * size) >> 3, hi = ((x + 1) * size) >> 3;
    return lo + i < hi ? round_base + lo + i : n_items;
  };
  int w = item_of_round();
  float2 nv[8];
  if (w < n_items) rv_fetch(nv, audio, ir, p, rv_item(w, n_ir, p), tid);
  // (the first input waited for HERE, once: left to the loop's head, where the first entry and the loop's own back edge meet,
  //  the wait becomes "everything outstanding" - and behind the back edge that is the four stores just issued, a store's
  //  round trip per transform; with it the loop waits at its foot for the loads alone: vmcnt(4))
  DDSP_WAIT_VMCNT0();
  bool first = true;
  while (w < n_items) {
    const RvItem it = rv_item(w, n_ir, p);
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = nv[m];
    round_base += (int)gridDim.x;
    w = item_of_round();
    if (w < n_items) rv_fetch(nv, audio, ir, p, rv_item(w, n_ir, p), tid);
    if (!first) __syncthreads();                 // the previous transform's last pass has read the array
    first = false;
    int t = tid;
    DDSP_KEEP_IN_VGPR(t);                        // (twiddles are functions of the thread index: left alone the compiler makes them once,
                                                 //  ahead of the loop, and keeps 14 + 42 registers of them alive through it)
    {                                            // q = 1024: pos = tid, i0 = tid
      float2 tw[8];
      const float rev = (float)t * (0.125f / (float)(kRvN / 8));
      fft_powers8(make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)), tw);
      fft_dft8(v);
      s[RP(t)] = v[0];
#pragma unroll
      for (int m = 1; m < 8; ++m) s[RP(t + m * (kRvN / 8))] = cmulc(v[m], tw[m]);
    }
    __syncthreads();
#pragma unroll 1
    for (int q = kRvN / 64; q >= 2; q >>= 3) {   // q = 128, 16, 2
      const float inv_len = 0.125f / (float)q;
      const int pos = t & (q - 1);
      const int i0 = ((t - pos) << 3) + pos;
      float2 x[8], tw[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) x[m] = s[RP(i0 + m * q)];
      const float rev = (float)pos * inv_len;
      fft_powers8(make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)), tw);
      fft_dft8(x);
      s[RP(i0)] = x[0];
#pragma unroll
      for (int m = 1; m < 8; ++m) s[RP(i0 + m * q)] = cmulc(x[m], tw[m]);
      // behind the q = 16 pass the next one (q = 2) reads what the SAME sixteen lanes wrote - a 128-point group belongs to
      // lanes 16 k .. 16 k + 15 in both - : the wavefront's own LDS order is enough, no block barrier
      if (q == 16) DDSP_WAVE_LDS_SYNC(); else __syncthreads();
    }
    float2* __restrict__ spec = it.ir ? hspec : xspec;
    float4* __restrict__ dst = reinterpret_cast<float4*>(spec + ((size_t)it.b * (it.ir ? p.np : p.nb) + it.j) * kRvN);
    float4 y[kRvPairs];
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u) y[u] = *reinterpret_cast<const float4*>(&s[RP(2 * (t + kRvThreads * u))]);
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u)            // radix 2 on neighbours, twiddle 1
      dst[t + kRvThreads * u] = make_float4(y[u].x + y[u].z, y[u].y + y[u].w, y[u].x - y[u].z, y[u].y - y[u].w);
  }
}

__global__ __launch_bounds__(kRvThreads, 8) void rv_ifft_kernel(const float2* __restrict__ yspec, const float* __restrict__ audio,
                                                                  float* __restrict__ out, RvArgs p, int nbo, int j0, int n_items) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int tid = threadIdx.x;
  const bool pair_mode = p.pairs > 0;
  const float scale = 1.0f / (float)kRvN;
  const bool dry = (p.flags & DDSP_CONV_ADD_DRY) != 0;           // only with n_out == N (checked by the host)
  const bool rev_a = (p.flags & DDSP_CONV_REVERSE_AUDIO) != 0, rev_o = (p.flags & DDSP_CONV_REVERSE_OUT) != 0;
  const bool zero0 = (p.flags & DDSP_CONV_ZERO_OUT0) != 0;
  // items: (row, j) for j0 <= j < j0 + nbo - the spectra that hold kept samples
  auto spectrum = [&](int w) {
    const int b = w / nbo, j = j0 + (w - b * nbo);
    return reinterpret_cast<const float4*>(yspec + ((size_t)b * p.nb + (pair_mode ? j : 2 * j + 1)) * kRvN);
  };
  int w = blockIdx.x;
  float4 nv[kRvPairs];
  if (w < n_items) {
    const float4* __restrict__ src = spectrum(w);
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u) nv[u] = src[tid + kRvThreads * u];
  }
  DDSP_WAIT_VMCNT0();                             // (as in rv_fft_kernel)
  bool first = true;
  while (w < n_items) {
    const int b = w / nbo, j = j0 + (w - b * nbo);
    float4 v[kRvPairs];
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u) v[u] = nv[u];
    w += gridDim.x;
    if (w < n_items) {
      const float4* __restrict__ src = spectrum(w);
#pragma unroll
      for (int u = 0; u < kRvPairs; ++u) nv[u] = src[tid + kRvThreads * u];
    }
    if (!first) __syncthreads();
    first = false;
    int t = tid;
    DDSP_KEEP_IN_VGPR(t);                         // (as in rv_fft_kernel)
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u)            // radix 2 on neighbours on the way in
      *reinterpret_cast<float4*>(&s[RP(2 * (t + kRvThreads * u))]) =
          make_float4(v[u].x + v[u].z, v[u].y + v[u].w, v[u].x - v[u].z, v[u].y - v[u].w);
    __syncthreads();
#pragma unroll 1
    for (int q = 2; q <= kRvN / 64; q <<= 3) {   // q = 2, 16, 128
      const float inv_len = 0.125f / (float)q;
      const int pos = t & (q - 1);
      const int i0 = ((t - pos) << 3) + pos;
      const float rev = (float)pos * inv_len;
      float2 x[8], tw[8];
      fft_powers8(make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)), tw);
      x[0] = fft_conj(s[RP(i0)]);
#pragma unroll
      for (int m = 1; m < 8; ++m) x[m] = fft_conj(cmul(s[RP(i0 + m * q)], tw[m]));
      fft_dft8(x);
#pragma unroll
      for (int m = 0; m < 8; ++m) s[RP(i0 + m * q)] = fft_conj(x[m]);
      if (q == 2) DDSP_WAVE_LDS_SYNC(); else __syncthreads();      // (q = 2 -> q = 16: the same sixteen lanes, as in rv_fft_kernel)
    }
    // q = 1024 (pos = i0 = tid): outputs 4 .. 7 are elements P + tid + 1024 (jj - 4) - the samples overlap-save keeps
    {
      float2 x[8];
      {
        float2 tw[8];
        const float rev = (float)t * (0.125f / (float)(kRvN / 8));
        fft_powers8(make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)), tw);
        x[0] = fft_conj(s[RP(t)]);
#pragma unroll
        for (int m = 1; m < 8; ++m) x[m] = fft_conj(cmul(s[RP(t + m * (kRvN / 8))], tw[m]));
      }
      // the dry signal, requested behind the twiddle products (their registers are free) and ahead of the butterfly - at
      // CLAMPED indices, no test around a load (a test per load is a wait per load: eight round trips one after the other);
      // samples that are not written are not used either
      float d0[4], d1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) d0[r] = d1[r] = 0.0f;
      if (dry) {                                  // (uniform)
        const int hi = p.N - 1;
        if (pair_mode) {
          const float* __restrict__ a0 = audio + (size_t)(2 * b) * p.N;
          const float* __restrict__ a1 = audio + (size_t)min(2 * b + 1, p.B - 1) * p.N;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = j * kRvP + (t + kRvThreads * r) - p.delay;
            const int na = min(max(rev_a ? p.N - 1 - n : n, 0), hi);
            d0[r] = a0[na];
            d1[r] = a1[na];
          }
        } else {
          const float* __restrict__ arow = audio + (size_t)b * p.N;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n0 = 2 * j * kRvP + (t + kRvThreads * r) - p.delay, n1 = n0 + kRvP;
            d0[r] = arow[min(max(rev_a ? p.N - 1 - n0 : n0, 0), hi)];
            d1[r] = arow[min(max(rev_a ? p.N - 1 - n1 : n1, 0), hi)];
          }
        }
      }
      fft_dft8(x);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = t + kRvThreads * r;
        const float2 y = fft_conj(x[4 + r]);
        if (pair_mode) {
          const int n = j * kRvP + i - p.delay;
          if (n < 0 || n >= p.n_out) continue;
          const int no = rev_o ? p.n_out - 1 - n : n;
          const bool z = zero0 && n == 0;
          out[(size_t)(2 * b) * p.n_out + no] = z ? 0.0f : fmaf(y.x, scale, d0[r]);
          if (2 * b + 1 < p.B) out[(size_t)(2 * b + 1) * p.n_out + no] = z ? 0.0f : fmaf(y.y, scale, d1[r]);
        } else {
          float* __restrict__ orow = out + (size_t)b * p.n_out;
          const int n0 = 2 * j * kRvP + i - p.delay, n1 = n0 + kRvP;
          if (n0 >= 0 && n0 < p.n_out) orow[rev_o ? p.n_out - 1 - n0 : n0] = (zero0 && n0 == 0) ? 0.0f : fmaf(y.x, scale, d0[r]);
          if (n1 >= 0 && n1 < p.n_out) orow[rev_o ? p.n_out - 1 - n1 : n1] = (zero0 && n1 == 0) ? 0.0f : fmaf(y.y, scale, d1[r]);
        }
      }
    }
  }
}

// The multiply-add pass with the partition count a template parameter: the window of spectra rotates through its registers
// by unrolling instead of by moves, and twelve partitions (48 000 taps) take under 128 registers - four wavefronts per SIMD, so
// that batch 128's 4096 wavefronts are resident at once (at three per SIMD a quarter of them ran as a second round).
// FULL passes of NP spectra are STRAIGHT-LINE code: loads with clamped indices, no test around a load, a product or a store.
// With a test per spectrum (the first form of this kernel) every iteration began with `s_waitcnt vmcnt(0)` - at a join of
// control flow the compiler waits for everything outstanding, and on gfx9 that includes the STORE of the iteration before:
// sixteen store round trips one after the other were most of the kernel's 30 us.  In straight-line code the waits count
// exactly (the spectrum requested two iterations ago, not the store behind it).  What does not fill a pass runs in the
// tested form.  ODD_ONLY: two blocks of one row per spectrum - W_m for odd m only.
template <int NP>
__device__ __forceinline__ float4 rv_window_product(const float4 (&w)[NP], const float4 (&h)[NP], int r) {
  float2 y0 = make_float2(0.f, 0.f), y1 = make_float2(0.f, 0.f);
#pragma unroll
  for (int q = 0; q < NP; ++q) {                                   // slot (r - q) mod NP holds spectrum j - q (zeros before the first)
    const float4 x = w[(r - q + NP) % NP];
    const float2 a0 = cmul(make_float2(x.x, x.y), make_float2(h[q].x, h[q].y));
    const float2 a1 = cmul(make_float2(x.z, x.w), make_float2(h[q].z, h[q].w));
    y0.x += a0.x; y0.y += a0.y; y1.x += a1.x; y1.y += a1.y;
  }
  return make_float4(y0.x, y0.y, y1.x, y1.y);
}

// a 16-byte access at a wave-uniform base plus a 32-bit byte offset of the thread's own: the base stays in scalar registers
__device__ __forceinline__ float4 rv_ld16(const float4* base, unsigned off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void rv_st16(float4* base, unsigned off, float4 v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + off) = v;
}

template <int NP, bool ODD_ONLY>
__global__ __launch_bounds__(kRvMacThreads) void rv_mac_np_kernel(float4* __restrict__ xspec, const float4* __restrict__ hspec,
                                                                  RvArgs p) {
  // every address is a wave-uniform base (scalar registers: row, spectrum) plus this thread's bin pair: no address registers
  // per spectrum (twelve of them made ahead of the unrolled loop were 24 registers - three wavefronts per SIMD instead of four)
  const unsigned off = (blockIdx.x * kRvMacThreads + threadIdx.x) * 16u;      // bin pair < kRvN / 2, in bytes
  const int b = blockIdx.y;
  const float4* __restrict__ hrow = hspec + (size_t)(p.ir_batch == 1 ? 0 : b) * p.np * (kRvN / 2);
  float4* __restrict__ xrow = xspec + (size_t)b * p.nb * (kRvN / 2);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 h[NP], w[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    h[q] = (q < p.np) ? rv_ld16(hrow + (size_t)q * (kRvN / 2), off) : zero;
    w[q] = zero;
  }
  const int last = p.nb - 1;
  float4 n1 = rv_ld16(xrow, off), n2 = rv_ld16(xrow + (size_t)min(1, last) * (kRvN / 2), off);
  int j0 = 0;
  // full passes with every spectrum kept (the forward pass: m_first = 0): straight-line code.  The spectrum two ahead is read
  // without a test - past a row's last spectrum lies the next row's first, past the last row's the impulse response's spectra
  // (at least two of them: np >= 2).
  if (p.m_first == 0 && p.np >= 2) {
    for (; j0 + NP <= p.nb; j0 += NP) {
      float4* __restrict__ pass = xrow + (size_t)j0 * (kRvN / 2);
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        w[r] = n1; n1 = n2; n2 = rv_ld16(pass + (size_t)(r + 2) * (kRvN / 2), off);
        if (!ODD_ONLY || (r & 1) != 0)                            // (NP is even: the parity of j0 + r is r's)
          rv_st16(pass + (size_t)r * (kRvN / 2), off, rv_window_product<NP>(w, h, r));
      }
    }
  }
  // what does not fill a pass, and the correlations of the backward pass (spectra below the first kept sample only fill the window)
  for (; j0 < p.nb; j0 += NP) {
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      const int j = j0 + r;
      if (j < p.nb) {                                            // wave-uniform
        w[r] = n1; n1 = n2; n2 = rv_ld16(xrow + (size_t)min(j + 2, last) * (kRvN / 2), off);
        if ((!ODD_ONLY || (j & 1) != 0) && j >= p.m_first) rv_st16(xrow + (size_t)j * (kRvN / 2), off, rv_window_product<NP>(w, h, r));
      }
    }
  }
}

// More than sixteen partitions (impulse responses beyond 65 536 taps: gin/models/vst/vst_48k.gin's 72 000-tap FilteredNoiseReverb,
// the dL/d ir correlation of any clip longer than that): no register window holds them.  W_j = sum_q Z_{j-q} H_q reads spectra
// with indices <= j only, so IN PLACE still works if j runs DOWNWARDS - everything at or below j is untouched when W_j is formed.
// Every term is fetched from memory (np reads per output where the windowed kernels make one: a bin slice of a row's spectra is
// np x 16 bytes x nb per thread, re-read from L2); the plain form of the same sum, for the shapes the fast kernels do not take.
template <bool ODD_ONLY>
__global__ __launch_bounds__(kRvMacThreads) void rv_mac_desc_kernel(float4* __restrict__ xspec, const float4* __restrict__ hspec,
                                                                    RvArgs p) {
  const unsigned off = (blockIdx.x * kRvMacThreads + threadIdx.x) * 16u;
  const int b = blockIdx.y;
  const float4* __restrict__ hrow = hspec + (size_t)(p.ir_batch == 1 ? 0 : b) * p.np * (kRvN / 2);
  float4* __restrict__ xrow = xspec + (size_t)b * p.nb * (kRvN / 2);
  for (int j = p.nb - 1; j >= max(p.m_first, 0); --j) {
    if (ODD_ONLY && (j & 1) == 0) continue;
    float2 y0 = make_float2(0.f, 0.f), y1 = make_float2(0.f, 0.f);
    const int nq = min(p.np, j + 1);
    for (int q = 0; q < nq; ++q) {                                 // the same order of additions as rv_window_product: q upwards
      const float4 x = rv_ld16(xrow + (size_t)(j - q) * (kRvN / 2), off);
      const float4 h = rv_ld16(hrow + (size_t)q * (kRvN / 2), off);
      const float2 a0 = cmul(make_float2(x.x, x.y), make_float2(h.x, h.y));
      const float2 a1 = cmul(make_float2(x.z, x.w), make_float2(h.z, h.w));
      y0.x += a0.x; y0.y += a0.y; y1.x += a1.x; y1.y += a1.y;
    }
    rv_st16(xrow + (size_t)j * (kRvN / 2), off, make_float4(y0.x, y0.y, y1.x, y1.y));
  }
}

// number of Z spectra: two output blocks each, so the block count rounded up to even
static inline int rv_blocks(int N, int delay) { return (((N + delay + kRvP - 1) / kRvP) + 1) & ~1; }
static inline int rv_parts(int L) { return (L + kRvP - 1) / kRvP; }

}  // namespace ddsp

using namespace ddsp;

extern "C" size_t ddsp_fft_convolve_long_ex_workspace_bytes(int B, int Bir, int N, int L, int n_out, int delay) {
  if (B <= 0 || Bir <= 0 || N <= 0 || L <= 0 || n_out <= 0 || delay < 0) return 0;
  const size_t spectra = (size_t)B * rv_blocks(n_out, delay) + (size_t)Bir * rv_parts(L);
  return spectra * kRvN * sizeof(float2);
}

extern "C" size_t ddsp_fft_convolve_long_workspace_bytes(int B, int Bir, int N, int L, int delay) {
  return ddsp_fft_convolve_long_ex_workspace_bytes(B, Bir, N, L, N, delay);
}

extern "C" int ddsp_fft_convolve_long_ex_f32(const float* audio, const float* impulse_response,
                                             float* out, void* workspace, size_t workspace_bytes,
                                             int B, int Bir, int N, int L, int n_out, int delay,
                                             unsigned flags, void* stream) {
  if (!audio || !impulse_response || !out || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || L <= 0 || n_out <= 0 || delay < 0 || (Bir != B && Bir != 1)) return DDSP_ERR_BAD_SHAPE;
  if ((flags & DDSP_CONV_ADD_DRY) && n_out != N) return DDSP_ERR_BAD_SHAPE;
  if (B > 65535 || (long)L >= (1L << 28)) return DDSP_ERR_UNSUPPORTED;             // (any number of partitions: rv_mac_desc_kernel beyond sixteen)
  if (workspace_bytes < ddsp_fft_convolve_long_ex_workspace_bytes(B, Bir, N, L, n_out, delay) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  RvArgs p;
  p.N = N; p.L = L; p.n_out = n_out; p.nb = rv_blocks(n_out, delay); p.np = rv_parts(L); p.delay = delay;
  p.flags = flags; p.ir_batch = Bir;
  static const bool plain_order = getenv("DDSP_EXP_RV_PLAIN_ORDER") != nullptr;     // (A/B of the item order, tools/exp_xcd_order.sh)
  if (plain_order) p.flags |= DDSP_CONV_EXP_PLAIN_ORDER;
  // row pairs: one impulse response for at least two rows - the trainable Reverb of the shipped configurations, effects.py:62-80
  // (DDSP_EXP_REVERB=single keeps the one-row form for the A/B)
  static const bool single_env = [] { const char* e = getenv("DDSP_EXP_REVERB"); return e && e[0] == 's'; }();
  const bool pair_mode = Bir == 1 && B >= 2 && !single_env;
  p.pairs = pair_mode ? (B + 1) / 2 : 0; p.B = B;
  if (pair_mode) p.nb = (n_out + delay + kRvP - 1) / kRvP;          // (no rounding up to even: a spectrum is one block)
  const int rows_z = pair_mode ? p.pairs : B;
  float2* xspec = (float2*)workspace;
  float2* hspec = xspec + (size_t)rows_z * p.nb * kRvN;
  const size_t lds = (size_t)kRvStore * sizeof(float2);
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  static const bool attr2_set = [] {
    (void)hipFuncSetAttribute((const void*)rv_fft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvStore * sizeof(float2)));
    (void)hipFuncSetAttribute((const void*)rv_ifft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvStore * sizeof(float2)));
    return true;
  }();
  (void)attr2_set;
  const int q0 = delay / kRvP;                                          // the first block with a kept sample
  const int j0 = pair_mode ? q0 : q0 / 2;                               // ... and the inverse transform it comes out of
  p.m_first = pair_mode ? q0 : 2 * j0 + 1;
  const int nbo = (pair_mode ? p.nb : p.nb / 2) - j0;
  {
    ProfileScope prof(kReverbFft, st);
    const int n_ir = Bir * p.np, n_items = n_ir + rows_z * p.nb;
    hipLaunchKernelGGL(rv_fft_kernel, dim3((unsigned)(n_items < 2 * n_cu ? n_items : 2 * n_cu)), dim3(kRvThreads), lds, st,
                       audio, impulse_response, xspec, hspec, p, n_ir, n_items);
  }
  {
    ProfileScope prof(kReverbMac, st);
    const dim3 grid(kRvN / 2 / kRvMacThreads, (unsigned)rows_z), block(kRvMacThreads);
#define DDSP_RV_MAC(NP_)                                                                                                     \
  do {                                                                                                                       \
    if (pair_mode) hipLaunchKernelGGL((rv_mac_np_kernel<NP_, false>), grid, block, 0, st, (float4*)xspec, (const float4*)hspec, p); \
    else hipLaunchKernelGGL((rv_mac_np_kernel<NP_, true>), grid, block, 0, st, (float4*)xspec, (const float4*)hspec, p);      \
  } while (0)
    if (p.np <= 4) DDSP_RV_MAC(4);
    else if (p.np <= 8) DDSP_RV_MAC(8);
    else if (p.np <= 12) DDSP_RV_MAC(12);
    else if (p.np <= kRvMaxParts) hipLaunchKernelGGL(rv_mac_kernel, grid, block, 0, st, (float4*)xspec, (const float4*)hspec, p);      // (13 .. 16 partitions: the window of sixteen by moves, 138 registers)
    else if (pair_mode) hipLaunchKernelGGL((rv_mac_desc_kernel<false>), grid, block, 0, st, (float4*)xspec, (const float4*)hspec, p);
    else hipLaunchKernelGGL((rv_mac_desc_kernel<true>), grid, block, 0, st, (float4*)xspec, (const float4*)hspec, p);
#undef DDSP_RV_MAC
  }
  {
    ProfileScope prof(kReverbIfft, st);
    const int n_items = rows_z * nbo;
    hipLaunchKernelGGL(rv_ifft_kernel, dim3((unsigned)(n_items < 2 * n_cu ? n_items : 2 * n_cu)), dim3(kRvThreads), lds, st,
                       (const float2*)xspec, audio, out, p, nbo, j0, n_items);
  }
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

// dL/d ir of a Reverb whose ONE impulse response serves the whole batch (the trainable Reverb: effects.py:62-80): the rows'
// correlations [B, L] added up in a fixed order (row 0 first), tap 0 - the masked dry tap, effects.py:50-60 - set to zero.
namespace ddsp {
__global__ __launch_bounds__(256) void rv_sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int L,
                                                          int zero_tap0) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  float acc = 0.0f;
  for (int b0 = 0; b0 < B; b0 += 8) {            // eight rows requested together, added in row order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = b0 + u < B ? x[(size_t)(b0 + u) * L + l] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  out[l] = (zero_tap0 && l == 0) ? 0.0f : acc;
}
}  // namespace ddsp

extern "C" int ddsp_sum_rows_f32(const float* x, float* out, int B, int L, int zero_first, void* stream) {
  if (!x || !out) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || L <= 0) return DDSP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(ddsp::rv_sum_rows_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, B, L,
                     zero_first);
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

extern "C" int ddsp_fft_convolve_long_f32(const float* audio, const float* impulse_response,
                                          float* out, void* workspace, size_t workspace_bytes,
                                          int B, int Bir, int N, int L, int delay, unsigned flags,
                                          void* stream) {
  return ddsp_fft_convolve_long_ex_f32(audio, impulse_response, out, workspace, workspace_bytes, B,
                                       Bir, N, L, N, delay, flags, stream);
}
