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
// spectrum on chip - bin classes with the ring of spectra in registers - built, measured at 99 us and taken out again).
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
// indices) and the radix-2 stage on neighbours: five passes over the LDS and five barriers where radix 4 took seven
// (rounds 1-3; the SpectralLoss transforms went the same way: csrc/fft_radix8.h).
constexpr int kRvPairs = kRvN / 2 / kRvThreads;                // float4 pairs per thread in the radix-2 stage (4)
static_assert(kRvN / 8 == kRvThreads, "one radix-8 butterfly per thread");

__device__ __forceinline__ void fft_forward(float2* s, int tid) {
#pragma unroll 1
  for (int q = kRvN / 8; q >= 2; q >>= 3) {                    // q = 1024, 128, 16, 2
    const float inv_len = 0.125f / (float)q;                   // exact: powers of two
    const int pos = tid & (q - 1);
    const int i0 = ((tid - pos) << 3) + pos;
    float2 v[8], w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = s[RP(i0 + m * q)];
    const float rev = (float)pos * inv_len;                    // revolutions, exact
    fft_powers8(make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)), w);      // conj of the twiddles
    fft_dft8(v);
    s[RP(i0)] = v[0];
#pragma unroll
    for (int m = 1; m < 8; ++m) s[RP(i0 + m * q)] = cmulc(v[m], w[m]);
    __syncthreads();
  }
  {                                                            // radix-2, neighbours, twiddle 1
    float4 v[kRvPairs];
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u) v[u] = *reinterpret_cast<const float4*>(&s[RP(2 * (tid + kRvThreads * u))]);
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u)
      *reinterpret_cast<float4*>(&s[RP(2 * (tid + kRvThreads * u))]) =
          make_float4(v[u].x + v[u].z, v[u].y + v[u].w, v[u].x - v[u].z, v[u].y - v[u].w);
  }
  __syncthreads();
}

__device__ __forceinline__ void fft_inverse(float2* s, int tid) {
  {
    float4 v[kRvPairs];
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u) v[u] = *reinterpret_cast<const float4*>(&s[RP(2 * (tid + kRvThreads * u))]);
#pragma unroll
    for (int u = 0; u < kRvPairs; ++u)
      *reinterpret_cast<float4*>(&s[RP(2 * (tid + kRvThreads * u))]) =
          make_float4(v[u].x + v[u].z, v[u].y + v[u].w, v[u].x - v[u].z, v[u].y - v[u].w);
  }
  __syncthreads();
#pragma unroll 1
  for (int q = 2; q <= kRvN / 8; q <<= 3) {
    const float inv_len = 0.125f / (float)q;
    const int pos = tid & (q - 1);
    const int i0 = ((tid - pos) << 3) + pos;
    const float rev = (float)pos * inv_len;
    float2 v[8], w[8];
    fft_powers8(make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)), w);
    // undo y_m conj(w^m), then the conjugate transform (factor 8, part of the 1 / N left to the caller)
    v[0] = fft_conj(s[RP(i0)]);
#pragma unroll
    for (int m = 1; m < 8; ++m) v[m] = fft_conj(cmul(s[RP(i0 + m * q)], w[m]));
    fft_dft8(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[RP(i0 + j * q)] = fft_conj(v[j]);
    __syncthreads();
  }
}

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
};

// One block = one 8192-point forward transform: block / partition j of row (pair / impulse response) b, `stride` spectra per row.
// IS_IR: the rows are IR partitions.
template <bool IS_IR>
__device__ __forceinline__ void rv_fft_block(float2* s, const float* __restrict__ src, float2* __restrict__ spec, const RvArgs& p,
                                             int j, int b, int stride) {
  const int tid = threadIdx.x;
  const int len = IS_IR ? p.L : p.N;
  const bool pair_mode = !IS_IR && p.pairs > 0;
  const float* __restrict__ row = src + (size_t)(pair_mode ? 2 * b : b) * len;
  const float* __restrict__ row_im = (pair_mode && 2 * b + 1 < p.B) ? row + len : nullptr;    // (pair mode: the second row, or none)
  // IR partition: taps jP .. (j+1)P-1 then P zeros (imaginary part 0);
  // x spectrum m = j: real part samples (m-2)P .. mP-1 (block m-1), imaginary part (m-1)P .. (m+1)P-1 (block m)
  // pair mode: real and imaginary part are block j (samples (j-1)P .. (j+1)P-1) of the pair's two rows
  const int base = IS_IR ? j * kRvP : (pair_mode ? (j - 1) * kRvP : (j - 2) * kRvP);
  const int live = IS_IR ? kRvP : kRvN;
  // DDSP_CONV_REVERSE_AUDIO / _IR: logical sample g is stored at len-1-g (correlations for the backward pass)
  const bool rev = (p.flags & (IS_IR ? DDSP_CONV_REVERSE_IR : DDSP_CONV_REVERSE_AUDIO)) != 0;
  const bool vec = !rev && ((len & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  auto load4 = [&](const float* __restrict__ row, int g) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row != nullptr && g + 3 >= 0 && g < len) {
      if (rev) {
        if (g >= 0 && g < len) v.x = row[len - 1 - g];
        if (g + 1 >= 0 && g + 1 < len) v.y = row[len - 2 - g];
        if (g + 2 >= 0 && g + 2 < len) v.z = row[len - 3 - g];
        if (g + 3 >= 0 && g + 3 < len) v.w = row[len - 4 - g];
      } else if (vec && g >= 0 && g + 3 < len) {
        v = *reinterpret_cast<const float4*>(row + g);
      } else {
        if (g >= 0 && g < len) v.x = row[g];
        if (g + 1 >= 0 && g + 1 < len) v.y = row[g + 1];
        if (g + 2 >= 0 && g + 2 < len) v.z = row[g + 2];
        if (g + 3 >= 0 && g + 3 < len) v.w = row[g + 3];
      }
    }
    return v;
  };
  for (int i4 = tid; i4 < kRvN / 4; i4 += kRvThreads) {
    const int i = 4 * i4, g = base + i;
    float4 re = make_float4(0.f, 0.f, 0.f, 0.f), im = re;
    if (i < live) {
      re = load4(row, g);
      // effects.Reverb._mask_dry_ir (effects.py:50-60): tap 0 carries the dry signal -> 0
      if (IS_IR && g == 0 && (p.flags & DDSP_CONV_MASK_TAP0)) re.x = 0.0f;
      if (!IS_IR) im = pair_mode ? load4(row_im, g) : load4(row, g + kRvP);
    }
    *reinterpret_cast<float4*>(&s[RP(4 * i4)]) = make_float4(re.x, im.x, re.y, im.y);
    *reinterpret_cast<float4*>(&s[RP(4 * i4 + 2)]) = make_float4(re.z, im.z, re.w, im.w);
  }
  __syncthreads();
  fft_forward(s, tid);
  float4* __restrict__ dst = reinterpret_cast<float4*>(spec + ((size_t)b * stride + j) * kRvN);
  for (int i2 = tid; i2 < kRvN / 2; i2 += kRvThreads) dst[i2] = *reinterpret_cast<const float4*>(&s[RP(2 * i2)]);
}

// The audio's blocks and the impulse responses' partitions in ONE launch (round 5): grid (max(nb, np), rows + Bir) - rows
// blockIdx.y < rows are audio rows (row pairs), the rest impulse responses.  As two launches the second waited for the first's
// twelve blocks: one 8192-point transform of latency (15 - 19 us at any batch size, profiles/r05e) ahead of every call.
__global__ __launch_bounds__(kRvThreads) void rv_fft_kernel(const float* __restrict__ audio, const float* __restrict__ ir,
                                                            float2* __restrict__ xspec, float2* __restrict__ hspec, RvArgs p,
                                                            int rows) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int j = blockIdx.x, b = blockIdx.y;
  if (b < rows) {
    if (j < p.nb) rv_fft_block<false>(s, audio, xspec, p, j, b, p.nb);
  } else {
    if (j < p.np) rv_fft_block<true>(s, ir, hspec, p, j, b - rows, p.np);
  }
}

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
    if (p.pairs == 0 && (j & 1) == 0) continue;                // W_m only for odd m (blocks m-1 and m); row pairs: every m
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

__global__ __launch_bounds__(kRvThreads) void rv_ifft_kernel(const float2* __restrict__ yspec,
                                                             const float* __restrict__ audio,
                                                             float* __restrict__ out, RvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float2 s[];
  const int tid = threadIdx.x, j = blockIdx.x, b = blockIdx.y;
  const bool pair_mode = p.pairs > 0;
  // spectrum m = 2j+1 holds output blocks 2j (real part) and 2j+1 (imaginary part); row pairs: spectrum j holds output block j
  // of the pair's first row (real part) and of its second (imaginary part)
  const float4* __restrict__ srcv = reinterpret_cast<const float4*>(yspec + ((size_t)b * p.nb + (pair_mode ? j : 2 * j + 1)) * kRvN);
  for (int i2 = tid; i2 < kRvN / 2; i2 += kRvThreads) *reinterpret_cast<float4*>(&s[RP(2 * i2)]) = srcv[i2];
  __syncthreads();
  fft_inverse(s, tid);
  // overlap-save: the last P samples of the block are y[jP .. (j+1)P); out[n] = y[n + delay]
  const float scale = 1.0f / (float)kRvN;
  const bool dry = (p.flags & DDSP_CONV_ADD_DRY) != 0;           // only with n_out == N (checked by the host)
  const bool rev_a = (p.flags & DDSP_CONV_REVERSE_AUDIO) != 0, rev_o = (p.flags & DDSP_CONV_REVERSE_OUT) != 0;
  if (pair_mode) {
    const int b0 = 2 * b;
    const bool two = b0 + 1 < p.B;
    for (int i = tid; i < kRvP; i += kRvThreads) {
      const float2 y = s[RP(kRvP + i)];
      const int n = j * kRvP + i - p.delay;
      if (n < 0 || n >= p.n_out) continue;
      const int no = rev_o ? p.n_out - 1 - n : n, na = rev_a ? p.N - 1 - n : n;
      out[(size_t)b0 * p.n_out + no] = fmaf(y.x, scale, dry ? audio[(size_t)b0 * p.N + na] : 0.0f);
      if (two) out[(size_t)(b0 + 1) * p.n_out + no] = fmaf(y.y, scale, dry ? audio[(size_t)(b0 + 1) * p.N + na] : 0.0f);
    }
    return;
  }
  const float* __restrict__ arow = audio + (size_t)b * p.N;
  float* __restrict__ orow = out + (size_t)b * p.n_out;
  for (int i = tid; i < kRvP; i += kRvThreads) {
    const float2 y = s[RP(kRvP + i)];
    const int n0 = 2 * j * kRvP + i - p.delay, n1 = n0 + kRvP;
    if (n0 >= 0 && n0 < p.n_out)
      orow[rev_o ? p.n_out - 1 - n0 : n0] = fmaf(y.x, scale, dry ? arow[rev_a ? p.N - 1 - n0 : n0] : 0.0f);
    if (n1 >= 0 && n1 < p.n_out)
      orow[rev_o ? p.n_out - 1 - n1 : n1] = fmaf(y.y, scale, dry ? arow[rev_a ? p.N - 1 - n1 : n1] : 0.0f);
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
  if (B > 65535 || rv_parts(L) > kRvMaxParts) return DDSP_ERR_UNSUPPORTED;
  if (workspace_bytes < ddsp_fft_convolve_long_ex_workspace_bytes(B, Bir, N, L, n_out, delay) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  RvArgs p;
  p.N = N; p.L = L; p.n_out = n_out; p.nb = rv_blocks(n_out, delay); p.np = rv_parts(L); p.delay = delay;
  p.flags = flags; p.ir_batch = Bir;
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
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute((const void*)rv_fft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvStore * sizeof(float2)));
    (void)hipFuncSetAttribute((const void*)rv_ifft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kRvStore * sizeof(float2)));
    return true;
  }();
  (void)attr_set;
  {
    ProfileScope prof(kReverbFft, st);
    hipLaunchKernelGGL(rv_fft_kernel, dim3((unsigned)(p.nb > p.np ? p.nb : p.np), (unsigned)(rows_z + Bir)), dim3(kRvThreads), lds, st,
                       audio, impulse_response, xspec, hspec, p, rows_z);
  }
  {
    ProfileScope prof(kReverbMac, st);
    hipLaunchKernelGGL(rv_mac_kernel, dim3(kRvN / 2 / kRvMacThreads, (unsigned)rows_z), dim3(kRvMacThreads), 0, st,
                       (float4*)xspec, (const float4*)hspec, p);
  }
  {
    ProfileScope prof(kReverbIfft, st);
    hipLaunchKernelGGL(rv_ifft_kernel, dim3((unsigned)(pair_mode ? p.nb : p.nb / 2), (unsigned)rows_z), dim3(kRvThreads), lds, st,
                       (const float2*)xspec, audio, out, p);
  }
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

extern "C" int ddsp_fft_convolve_long_f32(const float* audio, const float* impulse_response,
                                          float* out, void* workspace, size_t workspace_bytes,
                                          int B, int Bir, int N, int L, int delay, unsigned flags,
                                          void* stream) {
  return ddsp_fft_convolve_long_ex_f32(audio, impulse_response, out, workspace, workspace_bytes, B,
                                       Bir, N, L, N, delay, flags, stream);
}
