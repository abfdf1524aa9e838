// FilteredNoise.__call__ (ddsp/synths.py:165-196) and core.fft_convolve (ddsp/core.py:1382-1473) on the gfx950 matrix
// cores for the shapes noise_mfma65_kernel (filtered_noise_mfma.hip) does not take: any number of bands, any window
// size, any frame size - the shapes the reference's own tests use (100 bands: synths_test.py:43-50; 256: a 510-tap
// filter; frames of 100 samples).  Rounds 1-3 ran these on two plain kernels (filtered_noise.hip: noise_ir_kernel, one
// LDS cosine per multiply; tv_fir_kernel, two LDS reads per multiply): 22 x and 106 x the canonical shape's time at
// batch 32 (profiles/r05c_generic_shapes_b32.txt).  Here: two kernels with the taps travelling through HBM between them (any
// shape up to 288 bands / ~1000 taps); for up to 128 bands and 256 taps the second of them designing its tiles' taps itself (one
// launch: 100 bands 32.7 us at batch 32); and, further down, the backward pass of the canonical filter on the same algebra
// (noise_bwd_mfma_kernel).
//
//   noise_ir_gemm_kernel   the IR design as ONE matrix product with a constant matrix.  frequency_impulse_response
//       (core.py:1534-1565) is a real inverse DFT of the magnitudes, apply_window_to_impulse_response (:1477-1531)
//       reorders and windows its samples - both linear, so
//           taps[row][kappa] = sum_m scaled_mag[row][m] C[m][kappa],
//           C[m][kappa] = window[kappa] c_m cos(2 pi m n(kappa) / L0) / L0,    c_m = 1 (m = 0, M - 1) or 2
//       with C made on the host once per (bands, window size) in double precision, split into fp16 hi / lo MFMA
//       B-fragments and kept on the device.  A block's eight wavefronts hold 16 rows each as A-fragments (exp_sigmoid and
//       the controls output happen on the way in) and walk the tap tiles, C's fragments passing through LDS.
//
//   tv_fir_mfma_kernel     the time-varying FIR  z[m] = sum_i x[i] h_{frame(i)}[m - i]  (what fft_convolve's framed FFT /
//       overlap-add computes) as Toeplitz products.  Every frame is cut into PIECES of at most 64 samples that share the
//       frame's taps.  A piece that starts at sample s is placed at offset s mod 16 of a zero-padded slot, so that its
//       outputs start at the 16-aligned index Z = s - s mod 16.  With an output index written Z + 16 a + b and a tap
//       index 16 q + d:
//           C[b][a] = sum_{p, d} X[b][(p, d)] H[(p, d)][a],    X[b][(p, d)] = slot[16 p + b - d],  H[(p, d)][a] = h[16 (a - p) + d]
//       p = 0 .. 5, d = 0 .. 15: K = 96, three v_mfma_f32_16x16x32_f16 steps, the same algebra as noise_mfma65_kernel's
//       pairs.  A wavefront takes a RUN of consecutive pieces and keeps the outputs of the whole run - 16 NT columns of
//       16 - in registers: piece r of the run adds into the columns from (Z_r - Z_run) / 16 on, so nothing is shifted
//       and nothing is carried; a lane's column decides which 8 taps it reads (or the row's group of zeros).  The runs'
//       results meet in an fp32 output buffer in LDS: runs are long enough that only NEIGHBOURS overlap, a run writes
//       the outputs up to the next run's first and adds the rest behind a barrier - every output is one write and at
//       most one add, in that order, whichever wavefront is faster.  A block = one tile: W runs, of which the first
//       few pieces only rebuild the history the taps reach back into.
//   fp32 results from fp16 matrix cores as everywhere in this library: both factors split x = hi + lo / 2048,
//   hi.hi and hi.lo + lo.hi accumulated in fp32 (generated noise is exact in fp16: common.h, no lo part).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "../../include/ddsp_amd.h"
#include "common.h"
#include "noise_ir_geom.h"
#include "profile.h"
#include "filtered_noise_general.h"

namespace ddsp {

typedef _Float16 gf_f16x8 __attribute__((ext_vector_type(8)));
typedef float gf_f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) GfU4 { uint32_t x, y, z, w; };       // 16 bytes from a 4-byte aligned address
constexpr float kGfLoScale = 2048.0f;

__device__ __forceinline__ void gf_split(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * kGfLoScale);
}
__device__ __forceinline__ uint32_t gf_pack(_Float16 a, _Float16 b) {
  return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}
__device__ __forceinline__ void gf_split8(const float (&v)[8], gf_f16x8& hi, gf_f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 h, l;
    gf_split(v[e], h, l);
    hi[e] = h;
    lo[e] = l;
  }
}

// =====================================================================================================================
// IR design
// =====================================================================================================================
constexpr int kGiWaves = 4;                    // 16 rows each (blocks of 64 rows: several per CU, each at its own barrier)
constexpr int kGiRows = 16 * kGiWaves;
constexpr int kGiMaxKSteps = 9;                // 288 bands

// tap tiles per LDS chunk: 32 KB, at least one tile
template <int KS> struct GiChunk { static constexpr int kNtg = (2048 / (KS * 128)) > 0 ? (2048 / (KS * 128)) : 1; };

// C's fragments: [tap tile nt][k-step ks][hi / lo][lane] 16 bytes: elements B[k = 32 ks + 8 (lane >> 4) + e][n = 16 nt + (lane & 15)]
template <int KS>
__global__ __launch_bounds__(64 * kGiWaves) void noise_ir_gemm_kernel(const float* __restrict__ mag, float* __restrict__ ctl_out,
                                                                      float* __restrict__ ir, const uint4* __restrict__ cm,
                                                                      long rows, int M, int L, int NT, float bias,
                                                                      int scale) {
  extern __shared__ __attribute__((aligned(16))) uint4 s_b[];              // two chunks of [kNtg][KS][2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const long row0 = (long)blockIdx.x * kGiRows + 16 * wave;
  const float kLog10 = 2.302585092994046f;

  // C's fragments pass through LDS a chunk of kNtg tap tiles at a time, two buffers: the loads of chunk c + 1 are in flight
  // (in registers) while chunk c is multiplied, and one barrier per chunk separates a buffer's readers from its next writer
  // (a block is a chain of L2 latencies, one per chunk, with little work in between: 25 of the kernel's 25 us at 100 bands
  // before the loads were issued a chunk ahead - profiles/r05f)
  constexpr int kTile = KS * 2 * 64;                                    // uint4 per tap tile
  constexpr int kNtg = GiChunk<KS>::kNtg, kChunk = kNtg * kTile, kPre = (kChunk + 64 * kGiWaves - 1) / (64 * kGiWaves);
  constexpr int kBuf = kPre * 64 * kGiWaves;                            // an LDS buffer: a chunk, rounded up to whole rounds of the copy
  // (the matrix is padded to whole chunks on the host: no bounds in the copies, so that `pre` stays in registers)
  typedef uint32_t gi_u32x4 __attribute__((ext_vector_type(4)));      // (an array of HIP's uint4 - a struct of unions - stays in scratch)
  gi_u32x4 pre[kPre];
#define DDSP_GI_FETCH(nt0)                                                       \
  {                                                                              \
    const gi_u32x4* src_ = reinterpret_cast<const gi_u32x4*>(cm) + (size_t)(nt0) * kTile + tid; \
    _Pragma("unroll") for (int j = 0; j < kPre; ++j) pre[j] = src_[64 * kGiWaves * j]; \
  }
#define DDSP_GI_STASH(buf)                                                       \
  {                                                                              \
    gi_u32x4* dst_ = reinterpret_cast<gi_u32x4*>(buf) + tid;                     \
    _Pragma("unroll") for (int j = 0; j < kPre; ++j) dst_[64 * kGiWaves * j] = pre[j]; \
  }
  DDSP_GI_FETCH(0);

  // The block's 64 rows of magnitudes are one contiguous stretch of HBM: read it as such (a lane fetching ITS fragment
  // elements - 16 bytes here, 16 bytes 32 further on, sixteen rows per instruction - ran at 1.1 TB/s: 45 of the kernel's
  // 57 us at batch 128, profiles/r05i), scale it, write the controls the same way, and leave it in LDS in rows of S
  // floats, S = 4 (mod 8) and >= the padded band count: the sixteen rows of a fragment read then start in sixteen different
  // groups of four banks.  The staging area is the chunk buffers' space; chunk 0 waits in registers meanwhile.
  float* const s_a = reinterpret_cast<float*>(s_b);
  const int S = 32 * KS + 4;
  const long block_row0 = (long)blockIdx.x * kGiRows;
  const int n_rows = (int)min((long)kGiRows, rows - block_row0);
#ifdef DDSP_GI_NO_SCALE
  scale = 0;
#endif
  {
    const float* src = mag + block_row0 * M;
    float* ctl = ctl_out ? ctl_out + block_row0 * M : nullptr;
    if ((M & 3) == 0 && ((((uintptr_t)mag) | ((uintptr_t)ctl_out)) & 15) == 0) {
      const int n4 = n_rows * (M >> 2), m4 = M >> 2;
      for (int k = tid; k < n4; k += 64 * kGiWaves) {
        float4 q = reinterpret_cast<const float4*>(src)[k];
        if (scale) {
          q.x = exp_sigmoid(q.x + bias, kLog10, 2.0f, 1e-7f);
          q.y = exp_sigmoid(q.y + bias, kLog10, 2.0f, 1e-7f);
          q.z = exp_sigmoid(q.z + bias, kLog10, 2.0f, 1e-7f);
          q.w = exp_sigmoid(q.w + bias, kLog10, 2.0f, 1e-7f);
        }
        if (ctl) reinterpret_cast<float4*>(ctl)[k] = q;
        const int r = k / m4, c4 = k - r * m4;
        *reinterpret_cast<float4*>(s_a + r * S + 4 * c4) = q;
      }
    } else {
      const int n = n_rows * M;
      for (int k = tid; k < n; k += 64 * kGiWaves) {
        float q = src[k];
        if (scale) q = exp_sigmoid(q + bias, kLog10, 2.0f, 1e-7f);
        if (ctl) ctl[k] = q;
        const int r = k / M;
        s_a[r * S + (k - r * M)] = q;
      }
    }
    // bands M .. 32 KS - 1 of every row, and the rows past the end: zero
    const int pad = 32 * KS - M;
    for (int k = tid; k < kGiRows * pad; k += 64 * kGiWaves) {
      const int r = k / pad;
      s_a[r * S + M + (k - r * pad)] = 0.0f;
    }
    for (int k = tid; k < (kGiRows - n_rows) * M; k += 64 * kGiWaves) {
      const int r = k / M;
      s_a[(n_rows + r) * S + (k - r * M)] = 0.0f;
    }
  }
  __syncthreads();
  // this wavefront's 16 rows as A-fragments: lane (i, g) holds bins 32 ks + 8 g .. + 7 of row i
  gf_f16x8 ah[KS], al[KS];
  // Magnitudes that are not squashed by exp_sigmoid (core.frequency_impulse_response on the caller's own: any scale) are brought
  // to [1/2, 1) by the power of two of the largest among this wavefront's sixteen rows before the fp16 hi / lo split; the
  // epilogue takes it back (the design is linear)
  int a_exp = 0;
  {
    const float* arow = s_a + (16 * wave + i) * S + 8 * g;
    if (!scale) {
      float mx = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 q0 = *reinterpret_cast<const float4*>(arow + 32 * ks), q1 = *reinterpret_cast<const float4*>(arow + 32 * ks + 4);
        mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(fabsf(q0.x), fabsf(q0.y))), fmaxf(fabsf(q0.z), fabsf(q0.w))),
                   fmaxf(fmaxf(fabsf(q1.x), fabsf(q1.y)), fmaxf(fabsf(q1.z), fabsf(q1.w))));
      }
      a_exp = pow2_exponent(wave_max_nonneg_dpp(mx));
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 q0 = *reinterpret_cast<const float4*>(arow + 32 * ks), q1 = *reinterpret_cast<const float4*>(arow + 32 * ks + 4);
      float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      if (a_exp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ldexpf(v[e], -a_exp);
      }
      gf_split8(v, ah[ks], al[ks]);
    }
  }
  __syncthreads();                          // (the fragments are in registers: the space is the chunk buffers' from here on)

  DDSP_GI_STASH(s_b);                       // (chunk 0 was fetched before the magnitudes: see the top of the kernel)
  __syncthreads();
#ifdef DDSP_GI_NO_CHUNKS
  NT = ah[0][0] == (_Float16)77.0f ? 1 : 0;
#endif
#pragma unroll 1
  for (int nt0 = 0, c = 0; nt0 < NT; nt0 += kNtg, ++c) {
    const int n_chunk = min(kNtg, NT - nt0);
    const uint4* buf = s_b + (c & 1) * kBuf;
#ifndef DDSP_GI_NO_FETCH
    if (nt0 + kNtg < NT) DDSP_GI_FETCH(nt0 + kNtg);
#endif
    // kG tap tiles at a time: 3 kG independent accumulation chains (a single tile's two chains of dependent MFMAs left
    // the matrix cores waiting for themselves: the loop was 11 of the kernel's 21 us at 100 bands, profiles/r05k)
    constexpr int kG = kNtg < 4 ? kNtg : 4;
#pragma unroll 1
    for (int t0 = 0; t0 < n_chunk; t0 += kG) {
      gf_f32x4 acc[kG], acc_hl[kG], acc_lh[kG];
#pragma unroll
      for (int u = 0; u < kG; ++u) {
        acc[u] = (gf_f32x4){0.f, 0.f, 0.f, 0.f};
        acc_hl[u] = (gf_f32x4){0.f, 0.f, 0.f, 0.f};
        acc_lh[u] = (gf_f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          // (tiles past the chunk's end multiply whatever the buffer holds: inside the buffer, never stored)
          const int t = t0 + u < kNtg ? t0 + u : kNtg - 1;
#ifndef DDSP_GI_NO_LDSREAD
          const gf_f16x8 bh = __builtin_bit_cast(gf_f16x8, buf[((t * KS + ks) * 2 + 0) * 64 + lane]);
          const gf_f16x8 bl = __builtin_bit_cast(gf_f16x8, buf[((t * KS + ks) * 2 + 1) * 64 + lane]);
#else
          const gf_f16x8 bh = ah[(ks + t) % KS], bl = al[(ks + t) % KS];
#endif
#ifndef DDSP_GI_NO_MFMA
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc[u], 0, 0, 0);
          acc_hl[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bl, acc_hl[u], 0, 0, 0);
          acc_lh[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bh, acc_lh[u], 0, 0, 0);
#else
          acc[u][0] += (float)bh[0] + (float)bl[1];
#endif
        }
      }
      // D[row 4 g + r][column i]
#pragma unroll
      for (int u = 0; u < kG; ++u) {
        const int col = 16 * (nt0 + t0 + u) + i;
        if (t0 + u < n_chunk && col < L) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long rr = row0 + 4 * g + r;
            const float v = ldexpf(acc[u][r] + (acc_hl[u][r] + acc_lh[u][r]) * (1.0f / kGfLoScale), a_exp);
#ifndef DDSP_GI_NO_STORE
            if (rr < rows) ir[rr * L + col] = v;
#else
            if (rr < rows && v == 1234.5f) ir[rr * L + col] = v;
#endif
          }
        }
      }
    }
    if (nt0 + kNtg < NT) DDSP_GI_STASH(s_b + ((c + 1) & 1) * kBuf);
    __syncthreads();
  }
#undef DDSP_GI_FETCH
#undef DDSP_GI_STASH
}

// ---- host: the constant matrix ----------------------------------------------------------------------------------------
static unsigned short gi_f16_bits(float x) {
  const _Float16 h = (_Float16)x;
  unsigned short b;
  memcpy(&b, &h, 2);
  return b;
}
static float gi_f16_value(unsigned short b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
struct GiMatrix {
  uint4* dev = nullptr; int NT = 0, KS = 0;
  // the DISTINCT columns only (a zero-phase response under a symmetric window is symmetric: tap kappa and its mirror are
  // the same column; some taps are zero), for the kernel that designs its own taps: NTu tiles of 16 columns, and where
  // each column's product goes: two tap indices per column, 0xFFFF = nowhere
  uint4* dev_u = nullptr; uint32_t* dest = nullptr; int NTu = 0;
  // the TRANSPOSE, for the backward pass (dL/d magnitudes = dL/d taps . C^T): B[k = tap][n = band] as fragments
  // [band tile nb][k-step of 32 taps][hi / lo][lane], KSt k-steps, NTb band tiles
  uint4* dev_t = nullptr; int KSt = 0, NTb = 0;
};

static const GiMatrix* gi_matrix(int M, int window_size) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int>, GiMatrix> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const IrGeom g = ir_geom(M, window_size);
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(dev, M, g.ws);
  auto it = cache.find(key);
  if (it != cache.end()) return &it->second;
  GiMatrix m;
  m.KS = (M + 31) / 32;
  m.NT = (g.L + 15) / 16;
  // C[m][kappa] in double, rounded once to fp32, split
  std::vector<float> c((size_t)m.KS * 32 * m.NT * 16, 0.0f);
  const double two_pi = 2.0 * 3.14159265358979323846264338327950288;
  for (int kappa = 0; kappa < g.L; ++kappa) {
    int n, widx;
    ir_tap_map(g, kappa, &n, &widx);
    // (angles folded into the first half turn, so that mirror columns come out bit for bit equal)
    // tf.signal.hann_window: the denominator is ws for even ws, ws - 1 for odd ws; one sample: [1.0] (noise_ir_geom.h)
    const int wden = hann_denominator(g.ws);
    const int wfold = widx < 0 ? 0 : (widx <= wden - widx ? widx : wden - widx);
    const double w = widx < 0 ? 0.0 : (g.ws == 1 ? 1.0 : 0.5 - 0.5 * cos(two_pi * (double)wfold / (double)wden));
    for (int b = 0; b < M; ++b) {
      long long ph = ((long long)b * n) % g.L0;
      if (ph > g.L0 - ph) ph = g.L0 - ph;
      const double cm = (b == 0 || b == M - 1) ? 1.0 : 2.0;                                         // irfft's weights
      c[(size_t)b * m.NT * 16 + kappa] = (float)(w * cm * cos(two_pi * (double)ph / (double)g.L0) / (double)g.L0);
    }
  }
  // padded with zeros to whole LDS chunks of the kernel (and to whole rounds of its copy loop)
  const int ntg = (2048 / (m.KS * 128)) > 0 ? (2048 / (m.KS * 128)) : 1;
  const size_t chunk_u4 = (size_t)ntg * m.KS * 128;
  const size_t round_u4 = (chunk_u4 + 64 * kGiWaves - 1) / (64 * kGiWaves) * (64 * kGiWaves);
  const size_t n_chunks = ((size_t)m.NT + ntg - 1) / ntg;
  std::vector<uint32_t> frags(((n_chunks - 1) * chunk_u4 + round_u4) * 4, 0u);
  for (int nt = 0; nt < m.NT; ++nt)
    for (int ks = 0; ks < m.KS; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int d = 0; d < 4; ++d) {
          uint32_t hi2 = 0, lo2 = 0;
          for (int h = 0; h < 2; ++h) {
            const int k = 32 * ks + 8 * (lane >> 4) + 2 * d + h, col = 16 * nt + (lane & 15);
            const float x = c[(size_t)k * m.NT * 16 + col];
            const unsigned short hb = gi_f16_bits(x);
            const unsigned short lb = gi_f16_bits((x - gi_f16_value(hb)) * kGfLoScale);
            hi2 |= (uint32_t)hb << (16 * h);
            lo2 |= (uint32_t)lb << (16 * h);
          }
          frags[((((size_t)nt * m.KS + ks) * 2 + 0) * 64 + lane) * 4 + d] = hi2;
          frags[((((size_t)nt * m.KS + ks) * 2 + 1) * 64 + lane) * 4 + d] = lo2;
        }
  if (hipMalloc((void**)&m.dev, frags.size() * 4) != hipSuccess) return nullptr;
  if (hipMemcpy(m.dev, frags.data(), frags.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(m.dev);
    return nullptr;
  }
  // the distinct columns
  {
    const int ld = m.NT * 16, K = m.KS * 32;
    std::vector<int> uniq;                                   // a representative tap per distinct non-zero column
    std::vector<uint32_t> dest;
    for (int kappa = 0; kappa < g.L; ++kappa) {
      bool zero = true;
      for (int b = 0; b < K && zero; ++b) zero = c[(size_t)b * ld + kappa] == 0.0f;
      if (zero) continue;
      int found = -1;
      for (size_t u = 0; u < uniq.size() && found < 0; ++u) {
        if ((dest[u] >> 16) != 0xFFFFu) continue;             // (already has its mirror)
        bool same = true;
        for (int b = 0; b < K && same; ++b) same = c[(size_t)b * ld + kappa] == c[(size_t)b * ld + uniq[u]];
        if (same) found = (int)u;
      }
      if (found >= 0) {
        dest[found] = (dest[found] & 0xFFFFu) | ((uint32_t)kappa << 16);
      } else {
        uniq.push_back(kappa);
        dest.push_back((uint32_t)kappa | 0xFFFF0000u);
      }
    }
    m.NTu = ((int)uniq.size() + 15) / 16;
    if (m.NTu < 1) m.NTu = 1;
    dest.resize((size_t)m.NTu * 16, 0xFFFFFFFFu);
    std::vector<uint32_t> fu((size_t)m.NTu * m.KS * 2 * 64 * 4, 0u);
    for (int nt = 0; nt < m.NTu; ++nt)
      for (int ks = 0; ks < m.KS; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int d = 0; d < 4; ++d) {
            uint32_t hi2 = 0, lo2 = 0;
            for (int h = 0; h < 2; ++h) {
              const int k = 32 * ks + 8 * (lane >> 4) + 2 * d + h, u = 16 * nt + (lane & 15);
              const float x = u < (int)uniq.size() ? c[(size_t)k * ld + uniq[u]] : 0.0f;
              const unsigned short hb = gi_f16_bits(x);
              const unsigned short lb = gi_f16_bits((x - gi_f16_value(hb)) * kGfLoScale);
              hi2 |= (uint32_t)hb << (16 * h);
              lo2 |= (uint32_t)lb << (16 * h);
            }
            fu[((((size_t)nt * m.KS + ks) * 2 + 0) * 64 + lane) * 4 + d] = hi2;
            fu[((((size_t)nt * m.KS + ks) * 2 + 1) * 64 + lane) * 4 + d] = lo2;
          }
    if (hipMalloc((void**)&m.dev_u, fu.size() * 4) != hipSuccess || hipMalloc((void**)&m.dest, dest.size() * 4) != hipSuccess ||
        hipMemcpy(m.dev_u, fu.data(), fu.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m.dest, dest.data(), dest.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(m.dev);
      if (m.dev_u) (void)hipFree(m.dev_u);
      if (m.dest) (void)hipFree(m.dest);
      return nullptr;
    }
  }
  {
    const int ld = m.NT * 16;
    m.KSt = (g.L + 31) / 32;
    m.NTb = (M + 15) / 16;
    std::vector<uint32_t> ft((size_t)m.NTb * m.KSt * 2 * 64 * 4, 0u);
    for (int nb = 0; nb < m.NTb; ++nb)
      for (int ks = 0; ks < m.KSt; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int d = 0; d < 4; ++d) {
            uint32_t hi2 = 0, lo2 = 0;
            for (int h = 0; h < 2; ++h) {
              const int t = 32 * ks + 8 * (lane >> 4) + 2 * d + h, band = 16 * nb + (lane & 15);
              const float x = (t < g.L && band < M) ? c[(size_t)band * ld + t] : 0.0f;
              const unsigned short hb = gi_f16_bits(x);
              const unsigned short lb = gi_f16_bits((x - gi_f16_value(hb)) * kGfLoScale);
              hi2 |= (uint32_t)hb << (16 * h);
              lo2 |= (uint32_t)lb << (16 * h);
            }
            ft[((((size_t)nb * m.KSt + ks) * 2 + 0) * 64 + lane) * 4 + d] = hi2;
            ft[((((size_t)nb * m.KSt + ks) * 2 + 1) * 64 + lane) * 4 + d] = lo2;
          }
    if (hipMalloc((void**)&m.dev_t, ft.size() * 4) != hipSuccess ||
        hipMemcpy(m.dev_t, ft.data(), ft.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(m.dev);                                    // (every buffer made so far: ADVICE r4)
      (void)hipFree(m.dev_u);
      (void)hipFree(m.dest);
      if (m.dev_t) (void)hipFree(m.dev_t);
      return nullptr;
    }
  }
  return &cache.emplace(key, m).first->second;
}

int noise_general_prepare(int M, int window_size) {
  if (M < 2 || !noise_ir_gemm_ok(M, window_size)) return 0;    // (shapes the plain kernels take need no constants)
  return gi_matrix(M, window_size) ? 0 : 1;
}

bool noise_ir_gemm_ok(int M, int window_size) {
  (void)window_size;
  return M >= 2 && (M + 31) / 32 <= kGiMaxKSteps;
}

template <int KS>
static void gi_launch(const float* mag, float* ctl_out, float* ir, const GiMatrix* m, long rows, int M, int L, float bias,
                      int scale, hipStream_t st) {
  const size_t chunk_u4 = (size_t)GiChunk<KS>::kNtg * KS * 128;
  size_t lds = 2 * ((chunk_u4 + 64 * kGiWaves - 1) / (64 * kGiWaves) * (64 * kGiWaves)) * 16;
  const size_t stage = (size_t)kGiRows * (32 * KS + 4) * sizeof(float);
  if (stage > lds) lds = stage;
  const unsigned grid = (unsigned)((rows + kGiRows - 1) / kGiRows);
  if (lds > 48 * 1024)      // (per launch: the attribute belongs to the current device's copy of the kernel)
    (void)hipFuncSetAttribute((const void*)noise_ir_gemm_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipLaunchKernelGGL((noise_ir_gemm_kernel<KS>), dim3(grid), dim3(64 * kGiWaves), lds, st, mag, ctl_out, ir,
                     m->dev, rows, M, L, m->NT, bias, scale);
}

int launch_noise_ir_gemm(const float* mag, float* ctl_out, float* ir, long rows, int M, int window_size, float bias,
                         int scale, hipStream_t st) {
  const GiMatrix* m = gi_matrix(M, window_size);
  if (!m) return DDSP_ERR_LAUNCH;
  const int L = ir_geom(M, window_size).L;
  ProfileScope prof(kNoiseIrGemm, st);
  switch (m->KS) {
    case 1: gi_launch<1>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 2: gi_launch<2>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 3: gi_launch<3>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 4: gi_launch<4>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 5: gi_launch<5>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 6: gi_launch<6>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 7: gi_launch<7>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 8: gi_launch<8>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    case 9: gi_launch<9>(mag, ctl_out, ir, m, rows, M, L, bias, scale, st); break;
    default: return DDSP_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

// =====================================================================================================================
// time-varying FIR
// =====================================================================================================================
constexpr int kGfPiece = 64;                   // samples per piece at most
constexpr int kGfSlot = 96;                    // elements per piece slot: 80 (15 of offset + 64 + 1) and a gap of 16 zeros
constexpr int kGfLdsBudget = 158 * 1024;

struct GfArgs {
  const float* x;              // [B, N] or null (generated noise)
  const float* ir;             // [Bir, F, L]
  float* out;                  // [B, N]
  int N, F, L, fs, start;
  int npf;                     // pieces per frame: ceil(fs / 64)
  FastDiv fs_div, npf_div;
  int Q;                       // blocks of 16 taps: ceil(L / 16)
  int R, W;                    // pieces per run (= per wavefront), runs per tile
  int hist, fresh;             // W R = hist + fresh pieces per tile: the first `hist` rebuild what the taps reach back into
  int max_rows;                // tap rows (frames) a tile stages at most
  int out_len;                 // floats of the LDS output buffer
  int x_part;                  // bytes of one part (hi or lo) of the noise: copy E, then copy O
  int x_o;                     // byte offset of copy O from copy E
  int tap_row_bytes, tap_plane;
  size_t ir_batch_stride;      // F L, or 0 (one filter for the whole batch: core.py:1433-1434)
  int ir_pairs;                // the tap rows may be read two floats at a time
  uint32_t k0, k1;
  uint64_t batch_offset;
  int taps_bounded;            // KS = 0: the taps in HBM are this library's own design of exp_sigmoid magnitudes (<= 2 in sum): no
                               // normalisation pass (it read every tap row a second time: 139 -> 223 us at 256 bands, r05_final)
  // (GEN = false with x == null: DDSP_NOISE_BITS_23 - the noise is made here with 23-bit samples, which take the hi / lo planes
  // that supplied noise takes; common.h)
  // taps designed in the kernel (template parameter KS > 0): FilteredNoise.__call__ in one launch
  const float* mag;            // [B, F, M]
  float* ctl;                  // [B, F, M] or null
  const uint4* cm;             // the constant matrix's fragments, distinct columns only (gi_matrix: dev_u)
  const uint32_t* dest;        // per column: the two taps it is (16 bits each, 0xFFFF = none)
  int NTu;                     // tiles of 16 distinct columns
  int M, scale, mag_vec;       // bands; exp_sigmoid on the way in; the rows may be read four floats at a time
  float bias;
  int row_groups;              // ceil(max_rows / 16)
  int taps_at;                 // byte offset of the tap table in LDS
};

// first sample of piece v
__device__ __forceinline__ int gf_piece_start(int v, const GfArgs& p, int* frame) {
  uint32_t c;
  const uint32_t f = fastdiv((uint32_t)v, p.npf_div, c);
  *frame = (int)f;
  return (int)f * p.fs + kGfPiece * (int)c;
}

// KS = 0: the taps come from HBM (p.ir).  KS > 0: the kernel designs them itself from the tile's rows of magnitudes - the
// constant matrix of noise_ir_gemm_kernel, KS k-steps, NPW (1 or 2) tiles of 16 columns per wavefront - and they never
// leave LDS.
template <bool GEN, int NT, int KS, int NPW>
__global__ __launch_bounds__(512, (NT == 4 && NPW == 1) ? 6 : 1) void tv_fir_mfma_kernel(GfArgs p) {      // (six wavefronts per SIMD = three blocks per CU: 80 registers)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float* const s_out = reinterpret_cast<float*>(smem);
  uint8_t* const s_x = smem + 4 * (size_t)p.out_len;                    // E hi, O hi [, E lo, O lo]
  const int x_lo = p.x_part;                                                  // byte offset of the lo parts (supplied noise)
  uint8_t* const s_taps = smem + p.taps_at;                             // hi plane, lo plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int b = blockIdx.y, T = blockIdx.x;
  const int WR = p.W * p.R;
  const int Vt0 = T * p.fresh - p.hist;                     // first staged piece (negative: before the clip, empty)
  const int V0 = Vt0 < 0 ? 0 : Vt0;
  int f_lo;
  const int i_lo = gf_piece_start(V0, p, &f_lo);            // first staged sample
  const int Z_t0 = i_lo & ~15;
  int f_dummy;
  const int i_end = min(p.N, gf_piece_start(Vt0 + WR, p, &f_dummy));

  // ---- the scale of what the caller supplies ----------------------------------------------------------------------
  // core.fft_convolve is fp32 and scale invariant (core.py:1382-1473: tf.signal's FFTs); an fp16 hi / lo split is exact to 22
  // bits only inside fp16's normal range - int16-range audio (x 32768) overflowed it, an impulse response of 1e-9 fell below
  // it.  Audio the caller supplies, taps that come from HBM and magnitudes that are not squashed by exp_sigmoid are brought to
  // [1/2, 1) by the power of two of the TILE's largest magnitude before they are split, and the outputs take the exponents back
  // as they leave the accumulators.  Generated noise (|x| < 1) and exp_sigmoid's values (<= 2) need none of it.  The maxima are
  // a pass of their own over the tile's inputs (L2 hits the second time round), reduced per wavefront, exchanged through 3 x 8
  // floats of LDS and read behind the first barrier that follows.
  __shared__ float s_wmax[3][8];
  {
    float xmx = 0.0f, hmx = 0.0f, mmx = 0.0f;
    if constexpr (!GEN) {
      if (p.x) {
        const float* xb = p.x + (size_t)b * p.N;
        for (int i = i_lo + tid; i < i_end; i += nthr) xmx = fmaxf(xmx, fabsf(xb[i]));
      }
    }
    const int rows_in = max(0, min(p.max_rows, p.F - f_lo));
    if constexpr (KS == 0) {
      if (!p.taps_bounded) {
        const float* irb = p.ir + (size_t)b * p.ir_batch_stride + (size_t)f_lo * p.L;
        for (int k = tid; k < rows_in * p.L; k += nthr) hmx = fmaxf(hmx, fabsf(irb[k]));
      }
    } else if (!p.scale) {
      const float* mb = p.mag + ((size_t)b * p.F + f_lo) * p.M;
      for (int k = tid; k < rows_in * p.M; k += nthr) mmx = fmaxf(mmx, fabsf(mb[k]));
    }
    xmx = wave_max_nonneg_dpp(xmx); hmx = wave_max_nonneg_dpp(hmx); mmx = wave_max_nonneg_dpp(mmx);
    if (lane == 0) { s_wmax[0][wave] = xmx; s_wmax[1][wave] = hmx; s_wmax[2][wave] = mmx; }
  }
  auto tile_exponent = [&](int which) {
    float m = 0.0f;
    for (int w = 0; w < p.W; ++w) m = fmaxf(m, s_wmax[which][w]);
    return pow2_exponent(m);
  };
  int x_exp = 0, h_exp = 0;                                  // audio; taps (KS > 0: the magnitudes' - the design is linear)

  // ---- the taps, designed here (KS > 0) -----------------------------------------------------------------------------------
  // The tile's frames f_lo .. are consecutive rows of the magnitudes: read as one stretch, scaled (the frames this tile owns
  // also go out as controls), split into fp16 hi / lo and laid down in LDS - in the space the noise and the output buffer
  // take later - as rows of SA halves, SA = 8 (mod 16): a lane's A-fragment is 16 aligned bytes and the sixteen rows of a
  // fragment read start in sixteen different groups of four banks.  Wavefront w then multiplies tap tiles w and w + 8 of
  // the constant matrix (its fragments straight from L2 into registers, requested before anything else) with the row
  // groups, and writes the products - split again - where the FIR's B-fragments are read from.
  if constexpr (KS > 0) {
    typedef uint32_t gi_u32x4 __attribute__((ext_vector_type(4)));
    const int NTt = p.NTu;                                    // tiles of 16 distinct columns
    gi_u32x4 bfr[NPW][KS][2];
    uint32_t col_dest[NPW];
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      col_dest[n] = 0xFFFFFFFFu;
      const int nt = wave + 8 * n;
      if (nt < NTt) {
        col_dest[n] = p.dest[16 * nt + (lane & 15)];
        const gi_u32x4* src = reinterpret_cast<const gi_u32x4*>(p.cm) + (size_t)nt * KS * 128 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bfr[n][ks][0] = src[(2 * ks + 0) * 64];
          bfr[n][ks][1] = src[(2 * ks + 1) * 64];
        }
      }
    }
    constexpr int SA = 32 * KS + 8;
    // (the last row group's rows past max_rows are not staged: what those lanes read - the lo plane, the tap table - only
    // reaches accumulator rows that are not written anywhere)
    const int rows_a = p.max_rows;
    const int a_plane = rows_a * SA * 2;                      // bytes of the hi plane; lo follows
    const float kLog10 = 2.302585092994046f;
    const int rows_valid = max(0, min(p.max_rows, p.F - f_lo));
    // the frames whose first piece lies in this tile's fresh range are this tile's to write out as controls
    const int own_lo = (T * p.fresh + p.npf - 1) / p.npf, own_hi = ((T + 1) * p.fresh + p.npf - 1) / p.npf;
    const float* src = p.mag + ((size_t)b * p.F + f_lo) * p.M;
    float* ctl = p.ctl ? p.ctl + ((size_t)b * p.F + f_lo) * p.M : nullptr;
    auto scaled = [&](float x) { return p.scale ? exp_sigmoid(x + p.bias, kLog10, 2.0f, 1e-7f) : x; };
    if (!p.scale) {                                           // (launch-uniform: raw magnitudes, scale_fn = None)
      __syncthreads();
      h_exp = tile_exponent(2);
    }
#ifdef DDSP_GF_NO_DESIGN_STAGE
    if (p.M == 12345)
#endif
    if (p.mag_vec) {
      const int m4 = p.M >> 2;
      for (int k = tid; k < rows_valid * m4; k += nthr) {
        float4 q = reinterpret_cast<const float4*>(src)[k];
        q.x = scaled(q.x); q.y = scaled(q.y); q.z = scaled(q.z); q.w = scaled(q.w);
        const int r = k / m4, c4 = k - r * m4;
        if (ctl && f_lo + r >= own_lo && f_lo + r < own_hi) reinterpret_cast<float4*>(ctl)[k] = q;
        _Float16 h[4], l[4];
        if (h_exp) { q.x = ldexpf(q.x, -h_exp); q.y = ldexpf(q.y, -h_exp); q.z = ldexpf(q.z, -h_exp); q.w = ldexpf(q.w, -h_exp); }
        gf_split(q.x, h[0], l[0]); gf_split(q.y, h[1], l[1]); gf_split(q.z, h[2], l[2]); gf_split(q.w, h[3], l[3]);
        uint8_t* dst = smem + (r * SA + 4 * c4) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(gf_pack(h[0], h[1]), gf_pack(h[2], h[3]));
        *reinterpret_cast<uint2*>(dst + a_plane) = make_uint2(gf_pack(l[0], l[1]), gf_pack(l[2], l[3]));
      }
    } else {
      for (int k = tid; k < rows_valid * p.M; k += nthr) {
        const float q = scaled(src[k]);
        const int r = k / p.M, c = k - r * p.M;
        if (ctl && f_lo + r >= own_lo && f_lo + r < own_hi) ctl[k] = q;
        _Float16 h, l;
        gf_split(ldexpf(q, -h_exp), h, l);
        *reinterpret_cast<_Float16*>(smem + (r * SA + c) * 2) = h;
        *reinterpret_cast<_Float16*>(smem + (r * SA + c) * 2 + a_plane) = l;
      }
    }
    // bands past M, rows past the last frame: zero (pairs of halves; M + pad and SA are even... M may be odd: halves)
    {
      const int pad = SA - p.M;
      for (int k = tid; k < rows_valid * pad; k += nthr) {
        const int r = k / pad, c = p.M + (k - r * pad);
        *reinterpret_cast<_Float16*>(smem + (r * SA + c) * 2) = (_Float16)0.0f;
        *reinterpret_cast<_Float16*>(smem + (r * SA + c) * 2 + a_plane) = (_Float16)0.0f;
      }
      const int rest = (rows_a - rows_valid) * SA / 2;         // dwords
      for (int k = tid; k < rest; k += nthr) {
        reinterpret_cast<uint32_t*>(smem + rows_valid * SA * 2)[k] = 0u;
        reinterpret_cast<uint32_t*>(smem + rows_valid * SA * 2 + a_plane)[k] = 0u;
      }
      // the tap table: the rows' groups of zeros, the taps that are zero whatever the magnitudes (no column writes them)
      for (int k = tid; k < (2 * p.tap_plane) >> 4; k += nthr) reinterpret_cast<uint4*>(s_taps)[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#ifdef DDSP_GF_NO_DESIGN_MFMA
    if (p.M == 12345)
#endif
    {
      const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int n = 0; n < NPW; ++n) {
        const int nt = wave + 8 * n;
        if (nt >= NTt) continue;                               // (wave-uniform)
#pragma unroll 1
        for (int rg = 0; rg < p.row_groups; ++rg) {
          gf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_hl = {0.f, 0.f, 0.f, 0.f}, acc_lh = {0.f, 0.f, 0.f, 0.f};
          const uint8_t* arow = smem + ((16 * rg + i16) * SA + 8 * g) * 2;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const gf_f16x8 ah = *reinterpret_cast<const gf_f16x8*>(arow + 64 * ks);
            const gf_f16x8 al = *reinterpret_cast<const gf_f16x8*>(arow + 64 * ks + a_plane);
            const gf_f16x8 bh = __builtin_bit_cast(gf_f16x8, bfr[n][ks][0]), bl = __builtin_bit_cast(gf_f16x8, bfr[n][ks][1]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            acc_hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_hl, 0, 0, 0);
            acc_lh = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc_lh, 0, 0, 0);
          }
          // D[row 4 g + r][column 16 nt + i16] -> tap k of the row: halfword k & 7 of group ((k >> 3) & 1, k >> 4); a
          // column is one tap or a tap and its mirror
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rg + 4 * g + r;
            if (row < p.max_rows) {
              _Float16 h, l;
              gf_split(acc[r] + (acc_hl[r] + acc_lh[r]) * (1.0f / kGfLoScale), h, l);
#pragma unroll
              for (int d = 0; d < 2; ++d) {
                const int k = d ? (int)(col_dest[n] >> 16) : (int)(col_dest[n] & 0xFFFFu);
                if (k != 0xFFFF) {
                  uint8_t* dst = s_taps + row * p.tap_row_bytes + (((k >> 3) & 1) * (p.Q + 1) + (k >> 4)) * 16 + 2 * (k & 7);
                  *reinterpret_cast<_Float16*>(dst) = h;
                  *reinterpret_cast<_Float16*>(dst + p.tap_plane) = l;
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();                                          // (the magnitudes' space is the noise's and the outputs' from here on)
  }

  // ---- zero everything a stale value could be read from --------------------------------------------------------------
  // (DDSP_GF_NO_*: parts of the kernel compiled out for tools/exp_noise_general.py's time accounting - wrong results)
#ifndef DDSP_GF_NO_ZERO
  {
    const int n16 = (4 * p.out_len + (GEN ? 1 : 2) * p.x_part) >> 4;
    uint4* z = reinterpret_cast<uint4*>(smem);
    for (int k = tid; k < n16; k += nthr) z[k] = make_uint4(0u, 0u, 0u, 0u);
  }
#endif
  __syncthreads();
  if constexpr (!GEN) x_exp = tile_exponent(0);
  if constexpr (KS == 0) h_exp = tile_exponent(1);
  const int o_exp = x_exp + h_exp;

  // ---- the taps of the tile's frames: fp32 rows from HBM, split, as groups of 8 ------------------------------------------
#ifndef DDSP_GF_NO_TAPS
  if constexpr (KS == 0) {
    const float* irb = p.ir + (size_t)b * p.ir_batch_stride;
    const int gpr = 2 * (p.Q + 1);                           // groups per row, the zero groups included
    for (int k = tid; k < p.max_rows * gpr; k += nthr) {
      const int row = k / gpr, gi = k - row * gpr;           // gi = 2 q + half: neighbouring lanes read neighbouring taps
      const int q = gi >> 1, half = gi & 1;
      const int f = f_lo + row;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.0f;
      if (f < p.F && q < p.Q) {
        const float* src = irb + (size_t)f * p.L;
        const int k0 = 16 * q + 8 * half;
        if (p.ir_pairs && k0 + 8 <= p.L) {                  // an even tap count on an 8-byte aligned base: four loads of two
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const float2 t = *reinterpret_cast<const float2*>(src + k0 + e);
            v[e] = t.x;
            v[e + 1] = t.y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + e < p.L) v[e] = src[k0 + e];
        }
      }
      gf_f16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ldexpf(v[e], -h_exp);
      gf_split8(v, hi, lo);
      unsigned char* dst = s_taps + row * p.tap_row_bytes + (half * (p.Q + 1) + q) * 16;
      *reinterpret_cast<gf_f16x8*>(dst) = hi;
      *reinterpret_cast<gf_f16x8*>(dst + p.tap_plane) = lo;
    }
  }
#endif

  // ---- the noise: reversed inside its piece's slot, two copies one element apart ---------------------------------------
  // sample i of piece v (first sample s, frame position rem) sits at slot element j' = s mod 16 + (i - s), stored at
  // element 16 + 96 (v - Vt0) + 79 - j' of copy E and one element further in copy O
  auto place = [&](int i, float val) {
    uint32_t rem;
    const uint32_t f = fastdiv((uint32_t)i, p.fs_div, rem);
    const int c = (int)(rem >> 6), within = (int)(rem & 63u);
    const int sl = (int)f * p.npf + c - Vt0;
    const int s = i - within;
    const int el = 16 + kGfSlot * sl + 79 - ((s & 15) + within);
    _Float16 hi, lo;
    if constexpr (!GEN) val = ldexpf(val, -x_exp);
    gf_split(val, hi, lo);
    *reinterpret_cast<_Float16*>(s_x + 2 * el) = hi;
    *reinterpret_cast<_Float16*>(s_x + p.x_o + 2 * (el + 1)) = hi;
    if constexpr (!GEN) {
      *reinterpret_cast<_Float16*>(s_x + x_lo + 2 * el) = lo;
      *reinterpret_cast<_Float16*>(s_x + x_lo + p.x_o + 2 * (el + 1)) = lo;
    }
  };
#ifndef DDSP_GF_NO_NOISE
  if constexpr (GEN) {
    for (int q = (i_lo >> 3) + tid; 8 * q < i_end; q += nthr) {
      const U4 r = noise_philox(U4{(uint32_t)q, (uint32_t)(p.batch_offset + b), 0u, 0u}, p.k0, p.k1);
      const float4 a = noise_quad(r, 0), c4 = noise_quad(r, 1);
      const float v[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = 8 * q + e;
        if (i >= i_lo && i < i_end) place(i, v[e]);
      }
    }
  } else if (p.x == nullptr) {                                // 23-bit samples made here (|x| < 1: x_exp = 0)
    for (int q = (i_lo >> 2) + tid; 4 * q < i_end; q += nthr) {
      const float4 a = noise_quad_at(4u * (uint32_t)q, p.batch_offset + b, p.k0, p.k1, true);
      const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        if (i >= i_lo && i < i_end) place(i, v[e]);
      }
    }
  } else {
    const float* xb = p.x + (size_t)b * p.N;
    for (int i = i_lo + tid; i < i_end; i += nthr) place(i, xb[i]);
  }
#endif
  __syncthreads();

  // ---- a wavefront's run of pieces -------------------------------------------------------------------------------------
  const int i16 = lane & 15, g = lane >> 4;
  gf_f32x4 acc[NT], acc_x[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = (gf_f32x4){0.f, 0.f, 0.f, 0.f};
    acc_x[t] = (gf_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  unsigned touched = 0;
  int Z_run = 0;
  bool have = false;
  const int v_first = Vt0 + wave * p.R;
#ifndef DDSP_GF_NO_MFMA
  if (wave < p.W) {
    // element 79 - (16 p + i - 8 (g & 1)) of the slot, p = 2 c + (g >> 1): its parity is that of 79 - i for every step
    const int x_copy = (i16 & 1) ? 0 : p.x_o + 2;           // even element -> copy E; odd -> copy O, one element further
    const int x_lane = 2 * (16 + 79 - (16 * (g >> 1) + i16 - 8 * (g & 1))) + x_copy;
#pragma unroll 1
    for (int r = 0; r < p.R; ++r) {
      const int v = v_first + r;
      if (v < 0) continue;
      int f;
      const int s = gf_piece_start(v, p, &f);
      if (f >= p.F || s >= p.N) break;
      if (!have) { Z_run = s & ~15; have = true; }
      const int colbase = ((s & ~15) - Z_run) >> 4;
      const uint8_t* xa = s_x + x_lane + 2 * kGfSlot * (v - Vt0);
      gf_f16x8 ah[3], al[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const GfU4 qh = *reinterpret_cast<const GfU4*>(xa - 64 * c);          // 32 elements down per step
        ah[c] = __builtin_bit_cast(gf_f16x8, qh);
        if constexpr (!GEN) {
          const GfU4 ql = *reinterpret_cast<const GfU4*>(xa - 64 * c + x_lo);
          al[c] = __builtin_bit_cast(gf_f16x8, ql);
        }
      }
      const uint8_t* trow = s_taps + (f - f_lo) * p.tap_row_bytes;
      const int half_off = (g & 1) * (p.Q + 1);
      const int q_lane = i16 - colbase - (g >> 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (16 * t + 16 <= colbase || 16 * t >= colbase + 6 + p.Q) continue;     // (wave-uniform)
        touched |= 1u << t;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int q = 16 * t + q_lane - 2 * c;
          const int grp = (q >= 0 && q < p.Q) ? half_off + q : p.Q;
          const gf_f16x8 bh = *reinterpret_cast<const gf_f16x8*>(trow + 16 * grp);
          const gf_f16x8 bl = *reinterpret_cast<const gf_f16x8*>(trow + 16 * grp + p.tap_plane);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bh, acc[t], 0, 0, 0);
          acc_x[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bl, acc_x[t], 0, 0, 0);
          if constexpr (!GEN) acc_x[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c], bh, acc_x[t], 0, 0, 0);
        }
      }
    }
  }
#endif
  // ---- the runs into the output buffer: a run OWNS the outputs up to where the next run's begin and writes them; what
  // it adds to the next run's (its taps reach that far, and no further: gf_plan) it adds once those are written, behind a
  // barrier.  No output is touched by two wavefronts at a time and every sum has one order.  (ds_add_f32 here was 33 of
  // the kernel's 48 us at 100 bands and batch 32: an LDS float atomic costs ~100 clocks of the CU's one LDS pipeline.)
  // D[row 4 g + r][column a]: outputs 16 a + 4 g + r - a lane's four values are 16 aligned bytes
  float* const o_run = s_out + (Z_run - Z_t0) + 16 * i16 + 4 * g;
  auto fin = [&](float a, float ax) { return ldexpf(a + ax * (1.0f / kGfLoScale), o_exp); };    // (the exponents of audio and taps back)
  int head = 0, limit = 0;
  if (have) {
    int fd;
    head = (gf_piece_start(max(v_first + p.R, 0), p, &fd) & ~15) - Z_run;
    limit = (gf_piece_start(max(v_first + 2 * p.R, 0), p, &fd) & ~15) - Z_run;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!((touched >> t) & 1u)) continue;
      if (256 * t + 16 * i16 + 4 * g < head)
        *reinterpret_cast<float4*>(o_run + 256 * t) = make_float4(fin(acc[t][0], acc_x[t][0]), fin(acc[t][1], acc_x[t][1]),
                                                                  fin(acc[t][2], acc_x[t][2]), fin(acc[t][3], acc_x[t][3]));
    }
  }
  __syncthreads();
  if (have) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!((touched >> t) & 1u)) continue;
      const int pos = 256 * t + 16 * i16 + 4 * g;
      if (pos >= head && pos < limit) {
        float4 cur = *reinterpret_cast<const float4*>(o_run + 256 * t);
        cur.x += fin(acc[t][0], acc_x[t][0]);
        cur.y += fin(acc[t][1], acc_x[t][1]);
        cur.z += fin(acc[t][2], acc_x[t][2]);
        cur.w += fin(acc[t][3], acc_x[t][3]);
        *reinterpret_cast<float4*>(o_run + 256 * t) = cur;
      }
    }
  }
  __syncthreads();

  // ---- the tile's own outputs: z in [first sample of piece T fresh, of piece (T + 1) fresh); out[n] = z[n + start] -----
  {
    int fd;
    const int z_lo = T == 0 ? 0 : gf_piece_start(T * p.fresh, p, &fd);
    const int z_hi = gf_piece_start((T + 1) * p.fresh, p, &fd);
    float* ob = p.out + (size_t)b * p.N;
    const int k_lo = max(max(z_lo, p.start) - Z_t0, 0);
    const int k_hi = min(min(z_hi, p.N + p.start) - Z_t0, p.out_len);
#ifndef DDSP_GF_NO_STORE
    for (int k = k_lo + tid; k < k_hi; k += nthr) ob[Z_t0 + k - p.start] = s_out[k];
#else
    if (k_lo + tid == k_hi + 12345) ob[0] = s_out[tid];
#endif
  }
}

// ---- host: how a tile is cut ---------------------------------------------------------------------------------------------
struct GfPlan { bool ok = false; int NT = 0; size_t lds = 0; GfArgs a; };

static GfPlan gf_plan(int F, int L, int N, bool gen, int design_ks = 0) {
  // What a tile costs in LDS is what decides how many wavefronts a CU holds (a piece is ~1.3 KB of noise, taps and
  // outputs): the kernel has no pipeline of its own - staging, products and the store are separated by barriers - so
  // it is the OTHER blocks of the CU that cover a block's latencies.  Short runs, many blocks.
  GfPlan best;
  double best_score = 0.0;
  const int fs = (N + F - 1) / F;
  const int npf = (fs + kGfPiece - 1) / kGfPiece;
  const int Q = (L + 15) / 16;
  const int hist = L > 1 ? npf * ((L - 1 + fs - 1) / fs) : 0;
  for (int NT : {4, 8}) {
    if (design_ks && NT != 4) continue;
    const int cols = 16 * NT - 6 - Q;
    if (cols < 0) continue;
    const int r_max = cols / 4 + 1;
    // only neighbouring runs may overlap: a run's outputs end before the 16-aligned first output of the run after the
    // next, i.e. R consecutive pieces span at least L - 1 + 15 samples
    int r_min = 1;
    while (r_min <= r_max && (long)(r_min / npf) * fs < L - 1 + 15) ++r_min;
    if (r_min > r_max) continue;
    // DDSP_EXP_GF_PLAN=W,R pins the cut (tools/exp_noise_general.py's sweep)
    static const char* plan_env = getenv("DDSP_EXP_GF_PLAN");
    int w_env = 0, r_env = 0;
    if (plan_env && sscanf(plan_env, "%d,%d", &w_env, &r_env) != 2) w_env = r_env = 0;
    for (int R : {r_max, (r_max + r_min) / 2, r_min, r_env}) {
      if (R < r_min || R > r_max || (r_env && R != r_env)) continue;
      for (int W : {8, 4, 2}) {
        if (w_env && W != w_env) continue;
        if (design_ks && W != 8) continue;                      // (a tap tile per wavefront and round: eight wavefronts)
        const int WR = W * R;
        const int fresh = WR - hist;
        if (fresh < 1) continue;
        GfArgs a;
        memset(&a, 0, sizeof(a));
        a.Q = Q; a.R = R; a.W = W; a.hist = hist; a.fresh = fresh; a.npf = npf; a.fs = fs;
        a.max_rows = (WR + npf - 1) / npf + 1;
        a.out_len = ((W - 1) * R * kGfPiece + 256 * NT + 3) & ~3;
        const int x_elems = 32 + kGfSlot * WR + 2;
        const int x_copy = (2 * x_elems + 15) & ~15;            // bytes of one copy
        a.x_o = x_copy;
        while (((a.x_o / 4) % 32) != 16) a.x_o += 16;          // copy O starts 16 banks away from copy E: the odd and the even
                                                                // rows of a fragment read then sit in different halves of the banks
        a.x_part = a.x_o + x_copy;
        a.tap_row_bytes = 32 * (Q + 1);
        a.tap_plane = a.max_rows * a.tap_row_bytes;
        size_t front = 4 * (size_t)a.out_len + (size_t)(gen ? 1 : 2) * a.x_part;       // outputs + noise
        a.row_groups = (a.max_rows + 15) / 16;
        if (design_ks) front = std::max(front, (size_t)2 * a.max_rows * (32 * design_ks + 8) * 2);   // ... or the magnitudes before them
        front = (front + 15) & ~(size_t)15;
        a.taps_at = (int)front;
        const size_t lds = front + 2 * (size_t)a.tap_plane;
        if (lds > (size_t)kGfLdsBudget) continue;
        const int per_cu = (int)std::min<size_t>(8, (size_t)(160 * 1024) / (lds + 512));
        const double score = (double)fresh / WR * std::min(24, W * per_cu);
        if (score > best_score) {
          best_score = score;
          best.ok = true; best.NT = NT; best.lds = lds; best.a = a;
        }
      }
    }
  }
  return best;
}

bool tv_fir_mfma_ok(int B, int Bir, int F, int L, int N) {
  if (B <= 0 || B > 65535 || F <= 0 || L <= 0 || N <= 0 || (Bir != B && Bir != 1)) return false;
  if ((long)N + L >= (1L << 30)) return false;
  const int fs = (N + F - 1) / F;
  if ((N + fs - 1) / fs != F) return false;
  return gf_plan(F, L, N, false).ok;
}

template <bool GEN, int NT, int KS, int NPW>
static void gf_launch_one(const GfPlan& pl, dim3 grid, dim3 block, hipStream_t st) {
  // (per launch, not once per process: the attribute belongs to the current device's copy of the kernel)
  if (pl.lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)tv_fir_mfma_kernel<GEN, NT, KS, NPW>, hipFuncAttributeMaxDynamicSharedMemorySize, kGfLdsBudget);
  hipLaunchKernelGGL((tv_fir_mfma_kernel<GEN, NT, KS, NPW>), grid, block, pl.lds, st, pl.a);
}

// the launch shared by the FIR alone (ks = 0) and the whole of FilteredNoise.__call__ (ks = the matrix's k-steps)
static int gf_launch(GfPlan& pl, int B, int Bir, int F, int L, int N, int start, const float* x, const float* ir, float* out,
                     uint64_t seed, uint64_t batch_offset, int ks, bool gen, hipStream_t st) {
  GfArgs& a = pl.a;
  a.x = x; a.ir = ir; a.out = out;
  a.N = N; a.F = F; a.L = L; a.start = start;
  a.fs_div = make_fastdiv((uint32_t)a.fs);
  a.npf_div = make_fastdiv((uint32_t)a.npf);
  a.ir_batch_stride = Bir == 1 ? 0 : (size_t)F * L;
  a.ir_pairs = ((L & 1) == 0 && (((uintptr_t)ir) & 7) == 0) ? 1 : 0;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
  a.batch_offset = batch_offset;
  // pieces whose outputs anyone wants: z < N + start
  const long frames_needed = ((long)N + start + a.fs - 1) / a.fs;
  const long pieces = std::max(frames_needed, (long)F) * a.npf;          // (every frame is some tile's to write out as controls)
  const long tiles = (pieces + a.fresh - 1) / a.fresh;
  if (tiles > 0x7fffffffL / 2) return DDSP_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)tiles, (unsigned)B), block((unsigned)(64 * a.W));
  // gen: the no-lo-plane instances (generated noise of 2048 levels); otherwise supplied noise or 23-bit samples made in the kernel
  ProfileScope prof(kTvFirMfma, st);
  const int npw = ks > 0 && a.NTu > 8 ? 2 : 1;
#define DDSP_GF_CASE(NT_, KS_, NPW_)                                        \
  if (pl.NT == NT_ && ks == KS_ && npw == NPW_) {                           \
    if (gen) gf_launch_one<true, NT_, KS_, NPW_>(pl, grid, block, st);      \
    else gf_launch_one<false, NT_, KS_, NPW_>(pl, grid, block, st);         \
  } else
  DDSP_GF_CASE(4, 0, 1) DDSP_GF_CASE(8, 0, 1)
  DDSP_GF_CASE(4, 1, 1) DDSP_GF_CASE(4, 2, 1) DDSP_GF_CASE(4, 3, 1) DDSP_GF_CASE(4, 4, 1)
  DDSP_GF_CASE(4, 1, 2) DDSP_GF_CASE(4, 2, 2) DDSP_GF_CASE(4, 3, 2) DDSP_GF_CASE(4, 4, 2)
  return DDSP_ERR_UNSUPPORTED;
#undef DDSP_GF_CASE
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

int launch_tv_fir_mfma(const float* x, const float* ir, float* out, int B, int Bir, int F, int L, int N, int start,
                       uint64_t seed, uint64_t batch_offset, int bits23, int taps_bounded, hipStream_t st) {
  const bool gen = x == nullptr && !bits23;
  GfPlan pl = gf_plan(F, L, N, gen);
  if (!pl.ok) return DDSP_ERR_UNSUPPORTED;
  pl.a.taps_bounded = taps_bounded;
  return gf_launch(pl, B, Bir, F, L, N, start, x, ir, out, seed, batch_offset, 0, gen, st);
}

// FilteredNoise.__call__ in one launch: up to 128 bands (four k-steps: the matrix's fragments of two tap tiles fit a
// wavefront's registers) and 256 taps (sixteen tap tiles: two per wavefront)
static bool gf_fused_plan(int B, int F, int M, int N, int window_size, bool gen, GfPlan* pl) {
  if (B <= 0 || B > 65535 || F <= 0 || N <= 0 || M < 2 || M > 128) return false;
  const IrGeom g = ir_geom(M, window_size);
  if (g.L > 256 || (long)N + g.L >= (1L << 30)) return false;
  const int fs = (N + F - 1) / F;
  if ((N + fs - 1) / fs != F) return false;
  *pl = gf_plan(F, g.L, N, gen, (M + 31) / 32);
  return pl->ok && pl->NT == 4 && pl->a.W == 8;
}
bool filtered_noise_general_fused_ok(int B, int F, int M, int N, int window_size) {
  GfPlan pl;
  return gf_fused_plan(B, F, M, N, window_size, false, &pl);
}
int launch_filtered_noise_general_fused(const float* mag, const float* x, float* out, float* ctl_out, int B, int F, int M, int N,
                                        int window_size, float bias, int scale, uint64_t seed, uint64_t batch_offset,
                                        int bits23, hipStream_t st) {
  GfPlan pl;
  const bool gen = x == nullptr && !bits23;
  if (!gf_fused_plan(B, F, M, N, window_size, gen, &pl)) return DDSP_ERR_UNSUPPORTED;
  const GiMatrix* m = gi_matrix(M, window_size);
  if (!m) return DDSP_ERR_LAUNCH;
  const IrGeom g = ir_geom(M, window_size);
  int start = (g.L - 1) / 2 - 1;                         // crop_and_compensate_delay's automatic start (core.py:1375-1376)
  if (start < 0) start = 0;
  GfArgs& a = pl.a;
  if (m->NTu > 16) return DDSP_ERR_UNSUPPORTED;
  a.mag = mag; a.ctl = ctl_out; a.cm = m->dev_u; a.dest = m->dest; a.NTu = m->NTu;
  a.M = M; a.scale = scale; a.bias = bias;
  a.mag_vec = ((M & 3) == 0 && ((((uintptr_t)mag) | ((uintptr_t)ctl_out)) & 15) == 0) ? 1 : 0;
  return gf_launch(pl, B, B, F, g.L, N, start, x, nullptr, out, seed, batch_offset, m->KS, gen, st);
}

// =====================================================================================================================
// backward of FilteredNoise.__call__ for the canonical filter (65 bands, full window: 128 taps; frames of 64 c samples)
// =====================================================================================================================
// dL/dh_f[t] = sum_{i in frame f} x[i] gz[i + t], gz[m] = dL/d audio[m - start]; then dL/d magnitudes = (dL/dh . C^T) exp_sigmoid'.
// (filtered_noise.hip, "Backward pass": noise_bwd_taps_kernel + noise_bwd_mags_kernel, 28 + 23 us at batch 32 against 11 forward.)
// The tap gradients are a correlation; with u = the frame's samples reversed it is the convolution (u * gz_window)[t + 63], and that is
// tv_fir_mfma_kernel's algebra with the roles swapped: the NOISE piece (reversed twice: in natural order, one element into its
// slot, so that output 64 + t is tap t) is the Toeplitz operand, the GRADIENT window is the dense one - and since a piece starts
// at a multiple of 64 samples, the 8 gradient samples of a B-fragment are an aligned 16-byte group of ONE linear array: no tap rows.
// Nothing is overlap-added: a frame keeps its own 8 columns of 16 taps; two frames share an MFMA tile (columns 0-7 / 8-15, each
// with its own half of the K range).  Then a second product per 16 frames with the transposed design matrix (window and irfft
// weights folded in), the derivative of exp_sigmoid in the epilogue.  One block = NF frames: no history, no carries.
struct NbArgs {
  const float* mag; const float* x; const float* g; float* grad_mag;
  const uint4* ct;             // the transposed design matrix's fragments (gi_matrix: dev_t)
  int N, F, M, fs, npf, start, NF;
  int scale;
  float bias;
  uint32_t k0, k1;
  uint64_t batch_offset;
  int gz_plane;                // bytes of one part (hi or lo) of the staged gradient
  int x_part, x_o;             // as GfArgs
  int x_at, dh_at;             // byte offsets of the noise planes and the tap-gradient planes in LDS
};
constexpr int kNbDhStride = 136;               // halves per frame row of the tap gradients (128 + 8)

template <bool GEN>
__global__ __launch_bounds__(512, 6) void noise_bwd_mfma_kernel(NbArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* const s_gz = smem;                                  // hi plane, lo plane: gz[m0 + n] at half n, n < NF fs + 128; then 16 bytes of zeros
  uint8_t* const s_x = smem + p.x_at;                          // E hi, O hi [, E lo, O lo]
  _Float16* const s_dh = reinterpret_cast<_Float16*>(smem + p.dh_at);   // [hi / lo][NF][kNbDhStride]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = 512;
  const int b = blockIdx.y, f0 = blockIdx.x * p.NF;
  const int m0 = f0 * p.fs;                                     // first sample of the tile
  const int span = p.NF * p.fs + 128;
  const int zero_at = 2 * span;                                 // byte offset (inside the hi plane's allocation) of a group of zeros
  const int P = p.NF * p.npf;                                   // pieces of 64 samples

  // the constant fragments of this wavefront's first task of the second product, requested now
  typedef uint32_t gi_u32x4 __attribute__((ext_vector_type(4)));
  const int RG = p.NF >> 4;                                     // row groups of 16 frames
  const int n_tasks = RG * 5;                                   // (row group, band tile): 65 bands = 5 tiles
  gi_u32x4 cfr[4][2];                                           // (one task's at a time: 32 registers, three blocks per CU)
  auto fetch_task = [&](int task) {
    const int nb = task / RG;
    const gi_u32x4* src = reinterpret_cast<const gi_u32x4*>(p.ct) + (size_t)nb * 4 * 128 + lane;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      cfr[ks][0] = src[(2 * ks + 0) * 64];
      cfr[ks][1] = src[(2 * ks + 1) * 64];
    }
  };
  if (wave < n_tasks) fetch_task(wave);

  // ---- stage the gradient window (split) and the noise pieces -------------------------------------------------------------
  // dL/d audio comes at any scale (loss scaling, sum-reduced losses, vanishing gradients) and so does noise the caller supplies;
  // the fp16 hi / lo split is exact to 22 bits only inside fp16's normal range.  Both are brought to [1/2, 1) by the power of two
  // of the TILE's largest magnitude before they are split, the tap gradients stay in those units through the second product
  // (|dh| <= frame size there), and the epilogue takes the two exponents back: scale invariant like tf.GradientTape's fp32
  // gradients (3e4 overflowed to NaN, 1e-10 lost 7-9 %).  A thread keeps its (at most two) groups of the gradient
  // window in registers across the barrier the maxima need - the one that was here already.
  __shared__ float s_wmax[2][8];
  constexpr int kNbGroups = 2;                                  // groups of 8 per thread: span <= 16 * 256 + 128 (launch_noise_bwd_mfma)
  float gv[kNbGroups][8];
  {
    const float* gb = p.g + (size_t)b * p.N;
    float gmx = 0.0f, xmx = 0.0f;
#pragma unroll
    for (int u = 0; u < kNbGroups; ++u) {                       // groups of 8; the group behind the span is the zeros
      const int k = tid + nthr * u;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = 8 * k + e;
        const long a = (long)m0 + n - p.start;                  // gz[m] = g[m - start]
        gv[u][e] = (n < span && a >= 0 && a < p.N) ? gb[a] : 0.0f;
        gmx = fmaxf(gmx, fabsf(gv[u][e]));
      }
    }
    if constexpr (!GEN) {
      if (p.x) {                                                // (null: 23-bit samples regenerated below, |x| < 1)
        const float* xb = p.x + (size_t)b * p.N;
        const int i_end = min(p.N, m0 + P * 64);
        for (int i = m0 + tid; i < i_end; i += nthr) xmx = fmaxf(xmx, fabsf(xb[i]));
      }
    }
    gmx = wave_max_nonneg_dpp(gmx);
    xmx = wave_max_nonneg_dpp(xmx);
    if (lane == 0) { s_wmax[0][wave] = gmx; s_wmax[1][wave] = xmx; }
    // the noise planes: zero, then the samples (a piece's sample i at element 31 + 96 slot + i)
    const int n16 = ((GEN ? 1 : 2) * p.x_part) >> 4;
    for (int k = tid; k < n16; k += nthr) reinterpret_cast<uint4*>(s_x)[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  int g_exp, x_exp = 0;
  {
    float gmx = 0.0f, xmx = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { gmx = fmaxf(gmx, s_wmax[0][w]); xmx = fmaxf(xmx, s_wmax[1][w]); }
    g_exp = pow2_exponent(gmx);
    if constexpr (!GEN) x_exp = pow2_exponent(xmx);
  }
#pragma unroll
  for (int u = 0; u < kNbGroups; ++u) {
    const int k = tid + nthr * u;
    if (k < (span + 8) / 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ldexpf(gv[u][e], -g_exp);
      gf_f16x8 hi, lo;
      gf_split8(v, hi, lo);
      *reinterpret_cast<gf_f16x8*>(s_gz + 16 * k) = hi;
      *reinterpret_cast<gf_f16x8*>(s_gz + p.gz_plane + 16 * k) = lo;
    }
  }
  {
    const int x_lo = p.x_part;
    auto place = [&](int i, float val) {                        // sample i of the row, inside the tile
      const int rel = i - m0;
      const int el = 31 + kGfSlot * (rel >> 6) + (rel & 63);
      _Float16 hi, lo;
      if constexpr (!GEN) val = ldexpf(val, -x_exp);
      gf_split(val, hi, lo);
      *reinterpret_cast<_Float16*>(s_x + 2 * el) = hi;
      *reinterpret_cast<_Float16*>(s_x + p.x_o + 2 * (el + 1)) = hi;
      if constexpr (!GEN) {
        *reinterpret_cast<_Float16*>(s_x + x_lo + 2 * el) = lo;
        *reinterpret_cast<_Float16*>(s_x + x_lo + p.x_o + 2 * (el + 1)) = lo;
      }
    };
    const int i_end = min(p.N, m0 + P * 64);
    if constexpr (GEN) {
      for (int q = (m0 >> 3) + tid; 8 * q < i_end; q += nthr) {
        const U4 r = noise_philox(U4{(uint32_t)q, (uint32_t)(p.batch_offset + b), 0u, 0u}, p.k0, p.k1);
        const float4 a = noise_quad(r, 0), c4 = noise_quad(r, 1);
        const float v[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (8 * q + e < i_end) place(8 * q + e, v[e]);
      }
    } else if (p.x == nullptr) {                                // DDSP_NOISE_BITS_23: the forward's 23-bit samples again
      for (int q = (m0 >> 2) + tid; 4 * q < i_end; q += nthr) {
        const float4 a = noise_quad_at(4u * (uint32_t)q, p.batch_offset + b, p.k0, p.k1, true);
        const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * q + e < i_end) place(4 * q + e, v[e]);
      }
    } else {
      const float* xb = p.x + (size_t)b * p.N;
      for (int i = m0 + tid; i < i_end; i += nthr) place(i, xb[i]);
    }
  }
  __syncthreads();

  // ---- the tap gradients of two frames per MFMA tile -------------------------------------------------------------------------
  const int i16 = lane & 15, g = lane >> 4;
  {
    const int x_copy = (i16 & 1) ? 0 : p.x_o + 2;
    const int x_lane = 2 * (16 + 79 - (16 * (g >> 1) + i16 - 8 * (g & 1))) + x_copy;
    const int half_col = i16 >> 3;                               // which frame of the pair this lane's column belongs to
#pragma unroll 1
    for (int pp = wave; pp < p.NF / 2; pp += 8) {
      gf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int c = 0; c < p.npf; ++c) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int fr = 2 * pp + half;
          const int slot = fr * p.npf + c;
          const uint8_t* xa = s_x + x_lane + 2 * kGfSlot * slot;
          const int base = fr * p.fs + 64 * c;                   // the piece's first sample relative to the tile: a multiple of 8
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) {
            const GfU4 qh = *reinterpret_cast<const GfU4*>(xa - 64 * ks);
            const gf_f16x8 ah = __builtin_bit_cast(gf_f16x8, qh);
            const int q = (i16 & 7) + 4 - 2 * ks - (g >> 1);     // block of 16 gradient samples this lane's column reads
            const int off = (half_col == half && q >= 0) ? 2 * (base + 16 * q + 8 * (g & 1)) : zero_at;
            const gf_f16x8 bh = *reinterpret_cast<const gf_f16x8*>(s_gz + off);
            const gf_f16x8 bl = *reinterpret_cast<const gf_f16x8*>(s_gz + p.gz_plane + off);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            acc_x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_x, 0, 0, 0);
            if constexpr (!GEN) {
              const GfU4 ql = *reinterpret_cast<const GfU4*>(xa - 64 * ks + p.x_part);
              acc_x = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gf_f16x8, ql), bh, acc_x, 0, 0, 0);
            }
          }
        }
      }
      // D[row 4 g + r][column]: tap 16 (column & 7) + 4 g + r of frame 2 pp + (column >> 3): four consecutive halves
      _Float16 hi[4], lo[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) gf_split(acc[r] + acc_x[r] * (1.0f / kGfLoScale), hi[r], lo[r]);
      _Float16* dst = s_dh + (2 * pp + half_col) * kNbDhStride + 16 * (i16 & 7) + 4 * g;
      *reinterpret_cast<uint2*>(dst) = make_uint2(gf_pack(hi[0], hi[1]), gf_pack(hi[2], hi[3]));
      *reinterpret_cast<uint2*>(dst + p.NF * kNbDhStride) = make_uint2(gf_pack(lo[0], lo[1]), gf_pack(lo[2], lo[3]));
    }
  }
  __syncthreads();

  // ---- dL/d magnitudes: 16 frames x 16 bands per task ----------------------------------------------------------------------------
  const float kLog10 = 2.302585092994046f;
#pragma unroll 1
  for (int task = wave; task < n_tasks; task += 8) {             // (wave-uniform)
    if (task != wave) fetch_task(task);                          // (the second task of wavefronts 0, 1: its fragments only now)
    const int nb = task / RG, rg = task - nb * RG;
    gf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_hl = {0.f, 0.f, 0.f, 0.f}, acc_lh = {0.f, 0.f, 0.f, 0.f};
    const _Float16* arow = s_dh + (16 * rg + i16) * kNbDhStride + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const gf_f16x8 ah = *reinterpret_cast<const gf_f16x8*>(arow + 32 * ks);
      const gf_f16x8 al = *reinterpret_cast<const gf_f16x8*>(arow + 32 * ks + p.NF * kNbDhStride);
      const gf_f16x8 bh = __builtin_bit_cast(gf_f16x8, cfr[ks][0]), bl = __builtin_bit_cast(gf_f16x8, cfr[ks][1]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
      acc_hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_hl, 0, 0, 0);
      acc_lh = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc_lh, 0, 0, 0);
    }
    const int band = 16 * nb + i16;
    if (band < p.M) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = f0 + 16 * rg + 4 * g + r;
        if (f < p.F) {
          const size_t at = ((size_t)b * p.F + f) * p.M + band;
          float v = ldexpf(acc[r] + (acc_hl[r] + acc_lh[r]) * (1.0f / kGfLoScale), g_exp + x_exp);
          if (p.scale) {
            const float xr = p.mag[at] + p.bias;
            const float y = exp_sigmoid_fast(xr, kLog10, 2.0f, 1e-7f);
            v *= kLog10 * (y - 1e-7f) * (1.0f - __builtin_amdgcn_rcpf(1.0f + __expf(-xr)));
          }
          p.grad_mag[at] = v;
        }
      }
    }
  }
}

bool noise_bwd_mfma_ok(int B, int F, int M, int N, int window_size) {
  static const bool off = [] { const char* e = getenv("DDSP_EXP_NOISE_BWD"); return e && e[0] == 'p'; }();      // "plain": the two kernels of rounds 1-3
  if (off || B <= 0 || B > 65535 || F <= 0 || N <= 0 || M != 65) return false;
  const IrGeom g = ir_geom(M, window_size);
  if (g.padding != 0) return false;
  const int fs = (N + F - 1) / F;
  return fs % 64 == 0 && fs <= 256 && (N + fs - 1) / fs == F && (long)N + 256 < (1L << 30);
}

int launch_noise_bwd_mfma(const float* magnitudes, const float* noise, const float* grad_audio, float* grad_magnitudes, int B, int F,
                          int M, int N, int window_size, float bias, int scale, uint64_t seed, uint64_t batch_offset, int bits23,
                          hipStream_t st) {
  const GiMatrix* m = gi_matrix(M, window_size);
  if (!m || m->KSt != 4 || m->NTb != 5) return DDSP_ERR_LAUNCH;
  const IrGeom g = ir_geom(M, window_size);
  NbArgs a;
  memset(&a, 0, sizeof(a));
  a.mag = magnitudes; a.x = noise; a.g = grad_audio; a.grad_mag = grad_magnitudes; a.ct = m->dev_t;
  a.N = N; a.F = F; a.M = M; a.fs = (N + F - 1) / F; a.npf = a.fs / 64;
  a.start = (g.L - 1) / 2 - 1;
  a.NF = a.npf <= 2 ? 32 : 16;                                   // at most 64 pieces per tile
  a.scale = scale; a.bias = bias;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.batch_offset = batch_offset;
  const int span = a.NF * a.fs + 128;
  if ((span + 8) / 8 > 2 * 512) return DDSP_ERR_UNSUPPORTED;     // (the kernel keeps two groups of 8 gradient samples per thread)
  a.gz_plane = (2 * span + 16 + 15) & ~15;
  const int P = a.NF * a.npf;
  const int x_elems = 32 + kGfSlot * P + 2;
  const int x_copy = (2 * x_elems + 15) & ~15;
  a.x_o = x_copy;
  while (((a.x_o / 4) % 32) != 16) a.x_o += 16;
  a.x_part = a.x_o + x_copy;
  a.x_at = 2 * a.gz_plane;
  const bool lo_planes = noise != nullptr || bits23 != 0;       // supplied noise, or 23-bit samples regenerated in the kernel
  a.dh_at = a.x_at + (lo_planes ? 2 : 1) * a.x_part;
  const size_t lds = (size_t)a.dh_at + 2 * (size_t)a.NF * kNbDhStride * 2;
  if (lds > (size_t)kGfLdsBudget) return DDSP_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((F + a.NF - 1) / a.NF), (unsigned)B);
  const void* fn = lo_planes ? (const void*)noise_bwd_mfma_kernel<false> : (const void*)noise_bwd_mfma_kernel<true>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kGfLdsBudget);
  ProfileScope prof(kNoiseBwdMfma, st);
  if (lo_planes) hipLaunchKernelGGL((noise_bwd_mfma_kernel<false>), grid, dim3(512), lds, st, a);
  else hipLaunchKernelGGL((noise_bwd_mfma_kernel<true>), grid, dim3(512), lds, st, a);
  return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH;
}

}  // namespace ddsp
