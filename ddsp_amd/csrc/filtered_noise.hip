// FilteredNoise kernels for gfx950 (MI355X).  Replaces ddsp/synths.py:165-196 ->
// ddsp/core.py:1534-1565 (frequency_impulse_response), :1477-1531
// (apply_window_to_impulse_response), :1382-1473 (fft_convolve) and :1338-1379
// (crop_and_compensate_delay).  See DESIGN.md "FilteredNoise".
//
// fft_convolve's framed FFT / overlap-add is algebraically the direct time-varying FIR
//     z[m] = sum_k x[m-k] * h_{frame(m-k)}[k],      out[n] = z[n + start],
// (tap set chosen by the frame of the INPUT sample; oracle test
// test_fft_convolve_equals_direct_time_varying_fir), which is what is evaluated here:
// no FFT, no [B,F,fft_size] complex intermediates in HBM.
#include <hip/hip_ext.h>
#include <cstdlib>
#include "common.h"
#include "noise_ir65.h"
#include "filtered_noise_mfma.h"
#include "filtered_noise_general.h"
#include "noise_ir_geom.h"
#include "profile.h"
#include "../../include/ddsp_amd.h"

namespace ddsp {

// ------------------------------------------------------------------------------------
// kernel: controls.  ctl = exp_sigmoid(mag + bias) (synths.py:176-177) or copy.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_controls_kernel(const float* __restrict__ mag,
                                                             float* __restrict__ ctl, size_t n,
                                                             float bias, int scale) {
  const float kLog10 = 2.302585092994046f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = mag[i];
    ctl[i] = scale ? exp_sigmoid(x + bias, kLog10, 2.0f, 1e-7f) : x;
  }
}

// ------------------------------------------------------------------------------------
// kernel: impulse-response design, general shapes.  One block per (batch*frame) row;
// lanes = causal taps; the row's (scaled) magnitudes and the cosine table live in LDS.
//   hz[n] = (1/L0) * ( mag[0] + mag[M-1]*cos(pi n) + 2*sum_{m=1}^{M-2} mag[m] cos(2 pi m n / L0) )
//   h[kappa] = window_zp[n(kappa)] * hz[n(kappa)]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_ir_kernel(const float* __restrict__ mag,
                                                       float* __restrict__ ctl_out,
                                                       float* __restrict__ ir, long rows, int M,
                                                       int window_size, float bias, int scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const IrGeom g = ir_geom(M, window_size);
  float* s_cos = smem;            // [L0]
  float* s_mag = smem + g.L0;     // [M]
  const float kLog10 = 2.302585092994046f;
  for (int i = threadIdx.x; i < g.L0; i += 256) s_cos[i] = cospif(2.0f * (float)i / (float)g.L0);
  const float inv_L0 = 1.0f / (float)g.L0;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += 256) {
      float x = mag[row * M + m];
      if (scale) x = exp_sigmoid(x + bias, kLog10, 2.0f, 1e-7f);
      if (ctl_out) ctl_out[row * M + m] = x;
      // fold the irfft weights in: DC and Nyquist bins count once, the others twice
      s_mag[m] = (m == 0 || m == M - 1) ? x : 2.0f * x;
    }
    __syncthreads();
    for (int kappa = threadIdx.x; kappa < g.L; kappa += 256) {
      int n, widx;
      ir_tap_map(g, kappa, &n, &widx);
      float acc = 0.0f;
      int idx = 0;                               // (m*n) mod L0, incrementally
      for (int m = 0; m < M; ++m) {
        acc = fmaf(s_mag[m], s_cos[idx], acc);
        idx += n;
        if (idx >= g.L0) idx -= g.L0;
      }
      // tf.signal.hann_window: 0.5 - 0.5 cos(2 pi i / n), n = ws for even ws, ws - 1 for odd ws; [1.0] for ws = 1 (noise_ir_geom.h)
      const float w = (widx < 0) ? 0.0f : (g.ws == 1 ? 1.0f : 0.5f - 0.5f * cospif(2.0f * (float)widx / (float)hann_denominator(g.ws)));
      ir[row * g.L + kappa] = w * (acc * inv_L0);
    }
  }
}

// ------------------------------------------------------------------------------------
// kernel: uniform noise (the on-chip stand-in for tf.random.uniform, synths.py:192-193)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uniform_noise_kernel(float* __restrict__ out, int N,
                                                            uint32_t k0, uint32_t k1,
                                                            uint64_t batch_offset, int bits23) {
  const int b = blockIdx.y;
  const int nq = (N + 7) / 8;                              // eight samples per octet: one Philox block (11 bits) or two (23: common.h)
  for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256) {
    const float4 lo = noise_quad_at(8u * (uint32_t)q, batch_offset + b, k0, k1, bits23 != 0);
    const float4 hi = noise_quad_at(8u * (uint32_t)q + 4u, batch_offset + b, k0, k1, bits23 != 0);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (8 * q + i < N) out[(size_t)b * N + 8 * q + i] = v[i];
  }
}

// ------------------------------------------------------------------------------------
// kernel: direct time-varying FIR, general shapes (v1).  One block per (tile of outputs,
// batch row).  LDS: the input samples the tile touches (zero outside [0,N)) and the tap
// sets of every frame those inputs belong to.  Each thread owns outputs tid + 256*r.
// ------------------------------------------------------------------------------------
struct FirArgs {
  const float* x;      // [B,N] or null (generate)
  const float* ir;     // [Bir,F,L]
  float* out;          // [B,N]
  int N, F, L, frame_size, start, tile;
  size_t ir_batch_stride;   // F*L, or 0 when Bir == 1 (broadcast, core.py:1433-1434)
  uint32_t k0, k1;
  uint64_t batch_offset;
  int bits23;               // generated noise: 23-bit samples (DDSP_NOISE_BITS_23) instead of the 2048 levels
};

template <bool GEN_NOISE>
__global__ __launch_bounds__(256) void tv_fir_kernel(FirArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * p.tile;
  const int n_out = min(p.tile, p.N - n0);
  // inputs needed: i in [n0 + start - (L-1), n0 + n_out - 1 + start]
  const int i_lo = n0 + p.start - (p.L - 1);
  const int n_in = n_out + p.L - 1;
  const int i_hi = i_lo + n_in - 1;
  const int f_lo = max(i_lo, 0) / p.frame_size;
  const int f_hi = min(min(i_hi, p.N - 1) / p.frame_size, p.F - 1);
  const int nfr = max(f_hi - f_lo + 1, 0);
  float* s_x = smem;                                   // [tile + L - 1]
  float* s_h = smem + ((p.tile + p.L - 1 + 3) & ~3);   // [nfr][L]

  for (int t = threadIdx.x; t < n_in; t += 256) {
    const int i = i_lo + t;
    float v = 0.0f;
    if (i >= 0 && i < p.N)
      v = GEN_NOISE ? philox_noise((uint32_t)i, p.batch_offset + b, p.k0, p.k1, p.bits23 != 0)
                    : p.x[(size_t)b * p.N + i];
    s_x[t] = v;
  }
  const float* irb = p.ir + (size_t)b * p.ir_batch_stride + (size_t)f_lo * p.L;
  for (int t = threadIdx.x; t < nfr * p.L; t += 256) s_h[t] = irb[t];
  __syncthreads();

  for (int o = threadIdx.x; o < n_out; o += 256) {
    const int m = n0 + o + p.start;              // index into the un-cropped convolution
    float acc = 0.0f;
    for (int f = f_lo; f <= f_hi; ++f) {         // uniform loop over the tile's frames
      const int ia = max(f * p.frame_size, m - p.L + 1);
      const int ib = min(min((f + 1) * p.frame_size - 1, m), p.N - 1);
      const float* h = s_h + (f - f_lo) * p.L;
      for (int i = ia; i <= ib; ++i) acc = fmaf(s_x[i - i_lo], h[m - i], acc);
    }
    p.out[(size_t)b * p.N + n0 + o] = acc;
  }
}

// ====================================================================================
// Fast path for the canonical shape (ae.gin): M = 65 bands -> 128-point zero-phase IR,
// full-length Hann window (window_size <= 0 or >= 128), frame_size = 64, L = 128.
// ====================================================================================

// (the compile-time cosine tables live in noise_ir65.h, shared with filtered_noise_mfma.hip)

// ---- IR design, lanes = frames --------------------------------------------------------
// One block = 64 consecutive (batch*frame) rows, 4 wavefronts.  Every lane keeps ITS row's 65
// magnitudes in registers; the cosine factor of (band m, tap n) is wave-uniform and comes from
// the constant table through scalar loads, so the inner product is pure v_fmac with an SGPR
// operand and no LDS traffic.  hz[-n] = hz[n] and cos(2 pi m (64-n)/128) = (-1)^m cos(2 pi m n/128)
// give taps n and 64-n from one even-m and one odd-m partial sum.  h[64 +- d] = win[d]*hz[d],
// h[0] = 0.  The four wavefronts split n = 0..32.
__global__ __launch_bounds__(256) void noise_ir65_kernel(const float* __restrict__ mag,
                                                         float* __restrict__ ctl_out,
                                                         float* __restrict__ ir, long rows,
                                                         float bias, int scale) {
  __shared__ float s_m[64 * 65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long r0 = (long)blockIdx.x * 64;
  const int nrows = (int)min((long)64, rows - r0);
  const float kLog10 = 2.302585092994046f;
  const float* __restrict__ src = mag + r0 * 65;
  for (int i = tid; i < nrows * 65; i += 256) {
    float x = src[i];
    if (scale) x = exp_sigmoid(x + bias, kLog10, 2.0f, 1e-7f);
    if (ctl_out) ctl_out[r0 * 65 + i] = x;
    s_m[i] = x;
  }
  for (int i = nrows * 65 + tid; i < 64 * 65; i += 256) s_m[i] = 0.0f;
  __syncthreads();
  float me[33], mo[32];
#pragma unroll
  for (int i = 0; i <= 32; ++i) me[i] = s_m[lane * 65 + 2 * i];
#pragma unroll
  for (int i = 0; i < 32; ++i) mo[i] = s_m[lane * 65 + 2 * i + 1];
  __syncthreads();                               // s_m is reused for the taps below
  for (int n = __builtin_amdgcn_readfirstlane(wave); n <= 32; n += 4) {
    const float* __restrict__ ce = kIr65.c + n * kIrRowStride;
    const float* __restrict__ co = ce + 40;
    float e = 0.0f, o = 0.0f;
#pragma unroll
    for (int i = 0; i <= 32; ++i) e = fmaf(me[i], ce[i], e);
#pragma unroll
    for (int i = 0; i < 32; ++i) o = fmaf(mo[i], co[i], o);
    s_m[lane * 65 + n] = kIr65.win[n] * (e + o);                 // g[n],    n in [0,32]
    if (n >= 1 && n < 32) s_m[lane * 65 + 64 - n] = kIr65.win[64 - n] * (e - o);   // g[64-n]
  }
  __syncthreads();
  // coalesced write of the 128 causal taps per row: h[kappa] = g[|kappa - 64|], h[0] = 0
  float4* __restrict__ dst = reinterpret_cast<float4*>(ir + r0 * 128);
  for (int i = tid; i < nrows * 32; i += 256) {
    const int row = i >> 5, k4 = (i & 31) * 4;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kappa = k4 + u;
      const int d = kappa >= 64 ? kappa - 64 : 64 - kappa;
      v[u] = (kappa == 0) ? 0.0f : s_m[row * 65 + d];
    }
    dst[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// ---- FIR, register tiled -------------------------------------------------------------------
// z[m] = sum_k x[m-k] h_{frame(m-k)}[k].  A lane owns kFirR = 16 consecutive outputs; the
// 128 taps are walked in 8 blocks of 16.  For tap block k0 the inputs are the 31 samples
// x[i0-16 .. i0+15], i0 = m0 - k0: the upper half belongs to frame(i0) and the lower half to
// frame(i0-16) (i0 is a multiple of 16 and frames are 64 long), so the 16x16 product splits
// into the triangle r >= c fed by frame(i0)'s taps and r < c fed by frame(i0-16)'s taps:
// 256 FMAs per 12 ds_read_b128, exactly 128 FMAs per output sample, no per-sample selects.
// LDS: x tile in four 16-byte-chunk planes (lane stride 64 B would be a 4-way conflict), tap
// rows padded to 132 dwords (the 16 frames a wavefront touches land on distinct bank quads).
constexpr int kFirR = 16;
constexpr int kFirWaves = 2;
constexpr int kFirTile = 64 * kFirR * kFirWaves;        // 2048 z-samples per block
constexpr int kFirFrames = kFirTile / 64 + 2;           // 34 tap rows
constexpr int kTapStride = 132;
constexpr int kFirXLen = kFirTile + 128;                // x[z0-128 .. z0+2047]
constexpr int kFirPlane = kFirXLen / 4;                 // dwords per x plane

__device__ __forceinline__ int fir_x_addr(int ip) {      // ip = i - (z0 - 128), dword address
  const int c = ip >> 2;
  return (c & 3) * kFirPlane + ((c >> 2) << 2) + (ip & 3);
}

__device__ __forceinline__ void fir_load16(const float* s_x, int ip, float (&v)[16]) {
  const int chunk = ip >> 2;                               // ip is a multiple of 16
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const int c = chunk + c4;
    const float4 t = *reinterpret_cast<const float4*>(&s_x[(c & 3) * kFirPlane + ((c >> 2) << 2)]);
    v[4 * c4] = t.x; v[4 * c4 + 1] = t.y; v[4 * c4 + 2] = t.z; v[4 * c4 + 3] = t.w;
  }
}

// one 16-tap block: hi = x[i0 .. i0+15] (already loaded), lo = x[i0-16 .. i0-1] (loaded here)
__device__ __forceinline__ void fir_tap_block(const float* s_x, const float* s_h, int irel, int k0,
                                              float (&acc)[16], const float (&hi)[16],
                                              float (&lo)[16]) {
  // frame(i0) and frame(i0-16) as tap-row indices (row 0 = frame J0-2); irel = i0 - z0 >= -112
  const int rowA = (irel + 128) >> 6;
  const int rowB = (irel + 112) >> 6;
  fir_load16(s_x, irel + 128 - 16, lo);
  float ta[16], tb[16];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const float4 va = *reinterpret_cast<const float4*>(&s_h[rowA * kTapStride + k0 + 4 * c4]);
    const float4 vb = *reinterpret_cast<const float4*>(&s_h[rowB * kTapStride + k0 + 4 * c4]);
    ta[4 * c4] = va.x; ta[4 * c4 + 1] = va.y; ta[4 * c4 + 2] = va.z; ta[4 * c4 + 3] = va.w;
    tb[4 * c4] = vb.x; tb[4 * c4 + 1] = vb.y; tb[4 * c4 + 2] = vb.z; tb[4 * c4 + 3] = vb.w;
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r >= c) acc[r] = fmaf(ta[c], hi[r - c], acc[r]);          // input i0 + (r-c)
      else        acc[r] = fmaf(tb[c], lo[16 + r - c], acc[r]);     // input i0 - (c-r)
    }
  }
}

struct Fir128Args {
  int N, F, start;
  uint32_t k0, k1;
  uint64_t batch_offset;
  int bits23;
};

template <bool GEN_NOISE>
__global__ __launch_bounds__(64 * kFirWaves, 4) void tv_fir128_kernel(
    const float* __restrict__ x /*[B,N] or null*/, const float* __restrict__ ir /*[B,F,128]*/,
    float* __restrict__ out /*[B,N]*/, Fir128Args p) {
  __shared__ __attribute__((aligned(16))) float s_x[kFirXLen];
  __shared__ __attribute__((aligned(16))) float s_h[kFirFrames * kTapStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int z0 = blockIdx.x * kFirTile;
  const int J0 = z0 / 64;

  // ---- stage x: quads, zero outside [0,N) ---------------------------------------------------
  for (int qd = tid; qd < kFirXLen / 4; qd += 64 * kFirWaves) {
    const int i = z0 - 128 + 4 * qd;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0 && i < p.N) {
      if (GEN_NOISE) {
        v = noise_quad_at((uint32_t)i, p.batch_offset + b, p.k0, p.k1, p.bits23 != 0);      // (i is a multiple of 4)
        if (i + 3 >= p.N) {
          if (i + 1 >= p.N) v.y = 0.f;
          if (i + 2 >= p.N) v.z = 0.f;
          v.w = 0.f;
        }
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
      }
    }
    *reinterpret_cast<float4*>(&s_x[fir_x_addr(4 * qd)]) = v;
  }
  // ---- stage tap rows J0-2 .. J0+31 (zeros outside [0,F)) -----------------------------------
  for (int t = tid; t < kFirFrames * 32; t += 64 * kFirWaves) {
    const int row = t >> 5, k4 = (t & 31) * 4;
    const int f = J0 - 2 + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f >= 0 && f < p.F)
      v = *reinterpret_cast<const float4*>(ir + ((size_t)b * p.F + f) * 128 + k4);
    *reinterpret_cast<float4*>(&s_h[row * kTapStride + k4]) = v;
  }
  __syncthreads();

  // ---- 16 outputs per lane ---------------------------------------------------------------------
  const int mrel = wave * (64 * kFirR) + lane * kFirR;    // m0 - z0
  float acc[kFirR];
#pragma unroll
  for (int r = 0; r < kFirR; ++r) acc[r] = 0.0f;

  float xa[16], xb[16];                   // x[i0 .. i0+15] and x[i0-16 .. i0-1], roles alternate
  fir_load16(s_x, mrel + 128, xa);                         // i0 - (z0 - 128) at k0 = 0
#pragma unroll 1
  for (int kb = 0; kb < 8; kb += 2) {                      // two tap blocks per trip: ping-pong
    fir_tap_block(s_x, s_h, mrel - 16 * kb, 16 * kb, acc, xa, xb);
    fir_tap_block(s_x, s_h, mrel - 16 * (kb + 1), 16 * (kb + 1), acc, xb, xa);
  }

  // ---- out[n] = z[n + start] ---------------------------------------------------------------------
  const long nbase = (long)z0 + mrel - p.start;
  float* __restrict__ o = out + (size_t)b * p.N;
#pragma unroll
  for (int r = 0; r < kFirR; ++r) {
    const long n = nbase + r;
    if (n >= 0 && n < p.N) o[n] = acc[r];
  }
}

// ---- fused FilteredNoise for the canonical shape: IR design + noise + FIR in ONE launch -------
// One block = kFnFrames = 62 output frames (3968 samples) of one batch row, 4 wavefronts.
//   1. the 64 magnitude rows of frames J0-2 .. J0+61 (contiguous in HBM) -> exp_sigmoid -> LDS
//   2. lanes = frames: every wavefront holds the 64 rows' magnitudes in registers and designs
//      its share of the taps n = 0..32 (cosines from the constant table through scalar loads);
//      the windowed taps go straight into the LDS tap table the FIR reads - they never touch HBM
//   3. the noise tile x[z0-128 .. z0+3967] is generated by Philox into LDS (or copied from HBM)
//   4. register-tiled FIR as tv_fir128_kernel (16 outputs per lane, 248 of 256 lanes busy)
//   5. outputs are transposed through LDS and stored as coalesced 8-byte pieces.
constexpr int kFnFrames = 62;
constexpr int kFnTile = kFnFrames * 64;                 // 3968 z-samples per block
constexpr int kFnXLen = kFnTile + 128;                  // 4096
constexpr int kFnPlane = kFnXLen / 4 + 16;              // 1024 dwords per plane, + 16 so that the four planes start
                                                        // 16 banks apart (the tile fill writes planes 0..3 from adjacent lanes)
constexpr int kFnUnion = 64 * 65;                       // floats: magnitude staging / x tile (4 planes of 1040) / out
static_assert(4 * kFnPlane <= kFnUnion, "x planes must fit the union buffer");

// Where output sample e of the tile lives in LDS between the FIR and the store phase: lane t holds
// outputs 16t..16t+15 and writes them as four float4; rotating the four slots by (t >> 2) & 3 spreads
// the 16 lanes of a pass over all 64 banks (a plain 64-byte lane stride is a 4-way conflict).
__device__ __forceinline__ int fn_out_addr(int e) {
  return (e & ~15) + ((((e >> 2) ^ (e >> 6)) & 3) << 2) + (e & 3);
}

__device__ __forceinline__ void fn_load16(const float* s_x, int ip, float (&v)[16]) {
  const int chunk = ip >> 2;
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const int c = chunk + c4;
    const float4 t = *reinterpret_cast<const float4*>(&s_x[(c & 3) * kFnPlane + ((c >> 2) << 2)]);
    v[4 * c4] = t.x; v[4 * c4 + 1] = t.y; v[4 * c4 + 2] = t.z; v[4 * c4 + 3] = t.w;
  }
}
// irel = i0 - z0 (multiple of 16, >= -112); rel0 = (z0 - 128) - f_first*fs >= 0; rows are frames
// relative to f_first.  (x + 0.5) * inv_fs floors exactly for the magnitudes involved (< 2^13).
__device__ __forceinline__ void fn_tap_block(const float* s_x, const float* s_h, int irel, int k0,
                                             float (&acc)[16], const float (&hi)[16],
                                             float (&lo)[16], int rel0, float inv_fs) {
  const int rowA = (int)(((float)(irel + 128 + rel0) + 0.5f) * inv_fs);   // frame(i0)    - f_first
  const int rowB = (int)(((float)(irel + 112 + rel0) + 0.5f) * inv_fs);   // frame(i0-16) - f_first
  fn_load16(s_x, irel + 128 - 16, lo);
  // two passes so that the taps of frame(i0) and of frame(i0-16) are never live together
  // (the 8-wavefront variant has 80 VGPRs: acc + hi + lo + one set of 16 taps = 64)
  {
    float ta[16];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 va = *reinterpret_cast<const float4*>(&s_h[rowA * kTapStride + k0 + 4 * c4]);
      ta[4 * c4] = va.x; ta[4 * c4 + 1] = va.y; ta[4 * c4 + 2] = va.z; ta[4 * c4 + 3] = va.w;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
      for (int r = c; r < 16; ++r) acc[r] = fmaf(ta[c], hi[r - c], acc[r]);
    }
  }
  {
    float tb[16];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 vb = *reinterpret_cast<const float4*>(&s_h[rowB * kTapStride + k0 + 4 * c4]);
      tb[4 * c4] = vb.x; tb[4 * c4 + 1] = vb.y; tb[4 * c4 + 2] = vb.z; tb[4 * c4 + 3] = vb.w;
    }
#pragma unroll
    for (int c = 1; c < 16; ++c) {
#pragma unroll
      for (int r = 0; r < c; ++r) acc[r] = fmaf(tb[c], lo[16 + r - c], acc[r]);
    }
  }
}

struct FusedNoiseArgs {
  int N, F, start, scale, fs;      // fs = frame size: a multiple of 16, >= 64
  float inv_fs;
  float bias;
  uint32_t k0, k1;
  uint64_t batch_offset;
  int bits23;
};

typedef _Float16 fn_f16x8 __attribute__((ext_vector_type(8)));
typedef float fn_f32x4 __attribute__((ext_vector_type(4)));
constexpr float kFnLoScale = 2048.0f;     // x = hi + lo / 2048 in two fp16 numbers (as harmonic_table.hip)

__device__ __forceinline__ void fn_split8(const float (&v)[8], fn_f16x8& hi, fn_f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 h = (_Float16)v[e];
    hi[e] = h;
    lo[e] = (_Float16)((v[e] - (float)h) * kFnLoScale);
  }
}

// NW wavefronts per block: 4, or 8 (the FIR's tap range split in two halves).
// The IR design's cosine transform - a [32 x 32] . [32 x 64 rows] product per parity with a constant left factor -
// runs on the fp16 matrix cores with both factors split hi + lo / 2048 (three products, fp32 accumulation), and the
// magnitudes go from HBM straight into the B-fragments: every lane loads the 16 bins of its fragments, the noise tile
// is generated while those loads are in flight, and exp_sigmoid runs on the registers.  (Round 2 timed three designs
// at batch 32 / 128: lanes = frames on the vector ALUs 23.6 / 56.6 us, matrix cores behind an LDS staging of the
// magnitudes 21.5 / 52.3, this one 19.6 / 47.3 - profiles/r02a_noise_ir_variants.json; the other two are gone.)
template <bool GEN_NOISE, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 6 : 3)) void noise_fused65_kernel(
    const float* __restrict__ mag /*[B,F,65]*/, const float* __restrict__ x /*[B,N] or null*/,
    float* __restrict__ ctl_out /*[B,F,65] or null*/, float* __restrict__ out /*[B,N]*/,
    FusedNoiseArgs p) {
  __shared__ __attribute__((aligned(16))) float s_u[kFnUnion];
  __shared__ __attribute__((aligned(16))) float s_h[64 * kTapStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int z0 = blockIdx.x * kFnTile;
  // frames whose taps this tile needs: those of the inputs x[z0-128 .. z0+3967]; the 64 staged rows
  // start at f_first = floor((z0-128)/fs) (negative for the first tile: zero rows)
  const int f_first = (z0 - 128 >= 0) ? (z0 - 128) / p.fs : -((128 - z0 + p.fs - 1) / p.fs);
  const int rel0 = (z0 - 128) - f_first * p.fs;
  // controls ownership: tile t writes frames [own_lo, own_hi) so that every frame is written once
  const int own_lo = (blockIdx.x == 0) ? 0 : f_first + 2;
  const int own_hi = (z0 + kFnTile - 128) / p.fs + 2;          // f_first of the next tile + 2
  const float kLog10 = 2.302585092994046f;
  // debug timeline (p.scale bit 30): ctl_out is reinterpreted as long long [blocks][8]
  const bool dbg_time = (p.scale & 0x40000000) != 0;
  long long* dbg = dbg_time ? reinterpret_cast<long long*>(ctl_out) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  int dbg_n = 0;
#define DDSP_STAMP() do { if (dbg_time && threadIdx.x == 0 && dbg_n < 8) dbg[dbg_n++] = wall_clock64(); } while (0)
  if (dbg_time) ctl_out = nullptr;
  const int do_scale = p.scale & 1;
  DDSP_STAMP();

  // this wavefront's share of the constant factor in MFMA A-operand layout, fetched before anything else.  Wavefront w designs the taps of rows 16 (w & 3) .. + 15; with 4
  // wavefronts it takes both tap tiles n = 0..15 and 16..31, with 8 the tile w >> 2.  Element e of lane
  // (i = lane & 15, g = lane >> 4): coefficient of tap n = 16 mt + i and bin 2 k' (+ 1), k' = 8 g + e.
  constexpr int kMt = (NW == 8) ? 1 : 2;
  const int mt0 = (NW == 8) ? (wave >> 2) : 0;
  fn_f16x8 ae_hi[kMt], ae_lo[kMt], ao_hi[kMt], ao_lo[kMt];
  {
#pragma unroll
    for (int q = 0; q < kMt; ++q) {
      const float* __restrict__ crow = kIr65.c + (16 * (mt0 + q) + (lane & 15)) * kIrRowStride + 8 * (lane >> 4);
      float ve[8], vo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { ve[e] = crow[e]; vo[e] = crow[40 + e]; }
      fn_split8(ve, ae_hi[q], ae_lo[q]);
      fn_split8(vo, ao_hi[q], ao_lo[q]);
    }
  }

  // the 16 bins of this lane's B-fragments (row = 16 (wave & 3) + (lane & 15), bins 16 (lane >> 4) .. + 15)
  // and bin 64 of that row, straight from HBM; rows outside [0, F) are fetched from frame 0 and masked afterwards
  // (unconditional loads: nothing waits on them until the noise tile below is done)
  struct __attribute__((packed, aligned(4))) U4f { float x, y, z, w; };      // a 16-byte load from a 4-byte aligned address
  U4f rq[4];
  float r_last = 0.0f;
  const int rrow = 16 * (wave & 3) + (lane & 15);
  const int rfr = f_first + rrow;
  const bool rvalid = rfr >= 0 && rfr < p.F;
  {
    const float* __restrict__ src = mag + ((size_t)b * p.F + (rvalid ? rfr : 0)) * 65;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) rq[c4] = *reinterpret_cast<const U4f*>(src + 16 * (lane >> 4) + 4 * c4);
    r_last = src[64];
  }

  DDSP_STAMP();    // 1: loads issued
  {
    // ---- the noise tile first: it depends on nothing that is in flight ---------------------------------
    for (int qd = tid; qd < kFnXLen / 4; qd += 64 * NW) {
      const int i = z0 - 128 + 4 * qd;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i >= 0 && i < p.N) {
        if (GEN_NOISE) {
          v = noise_quad_at((uint32_t)i, p.batch_offset + b, p.k0, p.k1, p.bits23 != 0);      // (i is a multiple of 4)
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
        }
      }
      *reinterpret_cast<float4*>(&s_u[(qd & 3) * kFnPlane + ((qd >> 2) << 2)]) = v;
    }
    // ---- IR design: controls in registers, fragments, products -----------------------------------------------------
    const int mg = lane >> 4;
    float y[16];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) { y[4 * c4] = rq[c4].x; y[4 * c4 + 1] = rq[c4].y; y[4 * c4 + 2] = rq[c4].z; y[4 * c4 + 3] = rq[c4].w; }
    float m_last = r_last;
    if (do_scale) {
#pragma unroll
      for (int c = 0; c < 16; ++c) y[c] = exp_sigmoid_fast(y[c] + p.bias, kLog10, 2.0f, 1e-7f);
      m_last = exp_sigmoid_fast(m_last + p.bias, kLog10, 2.0f, 1e-7f);
    }
    if (!rvalid) {
#pragma unroll
      for (int c = 0; c < 16; ++c) y[c] = 0.0f;
      m_last = 0.0f;
    }
    if (ctl_out && rvalid && rfr >= own_lo && rfr < own_hi && (NW == 4 || wave < 4)) {    // written by the owning tile only
      float* __restrict__ dst = ctl_out + ((size_t)b * p.F + rfr) * 65;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
        *reinterpret_cast<U4f*>(dst + 16 * mg + 4 * c4) = U4f{y[4 * c4], y[4 * c4 + 1], y[4 * c4 + 2], y[4 * c4 + 3]};
      if (mg == 0) dst[64] = m_last;
    }
    float ve[8], vo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ve[e] = y[2 * e]; vo[e] = y[2 * e + 1]; }
    fn_f16x8 be_hi, be_lo, bo_hi, bo_lo;
    fn_split8(ve, be_hi, be_lo);
    fn_split8(vo, bo_hi, bo_lo);
    float* __restrict__ hrow = s_h + rrow * kTapStride;
#pragma unroll
    for (int q = 0; q < kMt; ++q) {
      const fn_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      fn_f32x4 ea = __builtin_amdgcn_mfma_f32_16x16x32_f16(ae_hi[q], be_hi, zero, 0, 0, 0);
      fn_f32x4 oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(ao_hi[q], bo_hi, zero, 0, 0, 0);
      fn_f32x4 ex = __builtin_amdgcn_mfma_f32_16x16x32_f16(ae_hi[q], be_lo, zero, 0, 0, 0);
      fn_f32x4 ox = __builtin_amdgcn_mfma_f32_16x16x32_f16(ao_hi[q], bo_lo, zero, 0, 0, 0);
      ex = __builtin_amdgcn_mfma_f32_16x16x32_f16(ae_lo[q], be_hi, ex, 0, 0, 0);
      ox = __builtin_amdgcn_mfma_f32_16x16x32_f16(ao_lo[q], bo_hi, ox, 0, 0, 0);
      const fn_f32x4 ev = ea + ex * (1.0f / kFnLoScale), ov = oa + ox * (1.0f / kFnLoScale);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * (mt0 + q) + 4 * mg + r;                    // 0 .. 31
        const float e = fmaf(m_last, kIr65.c[n * kIrRowStride + 32], ev[r]);
        const float o = ov[r];
        const float g0 = kIr65.win[n] * (e + o);                       // g[n]:    taps 64+n and 64-n
        hrow[64 + n] = g0;
        if (n >= 1) {
          hrow[64 - n] = g0;
          const float g1 = kIr65.win[64 - n] * (e - o);                // g[64-n]: taps 128-n and n
          hrow[128 - n] = g1;
          hrow[n] = g1;
        }
      }
    }
    if (NW == 4 || wave < 4) {
      // tap 32: cos(pi m / 2) vanishes for odd bins; this lane's 8 even bins, then the row's four lanes together
      const float* __restrict__ c32 = kIr65.c + 32 * kIrRowStride + 8 * mg;
      float part = (mg == 0) ? m_last * kIr65.c[32 * kIrRowStride + 32] : 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(c32[e], ve[e], part);
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      if (mg == 0) {
        const float g0 = kIr65.win[32] * part;
        hrow[96] = g0;
        hrow[32] = g0;
        hrow[0] = 0.0f;                                                // h[0] = Hann(128)[0] * hz[-64] = 0
      }
    }
  }
  DDSP_STAMP();    // 2: IR designed
  __syncthreads();
  DDSP_STAMP();    // 3: noise tile staged
  // ---- 4. FIR: 16 outputs per lane; with 8 wavefronts, threads 256..511 take taps 64..127 of the
  //         same outputs (half the dependent chain per lane, twice the wavefronts to hide latency) ----
  const int half = (NW == 8) ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;
  const int mrel = (tid & 255) * kFirR;                  // m0 - z0; lanes 248..255 idle (>= 3968)
  float acc[kFirR];
#pragma unroll
  for (int r = 0; r < kFirR; ++r) acc[r] = 0.0f;
  if (mrel < kFnTile) {
    const int kb0 = (NW == 8) ? 4 * half : 0;
    float xa[16], xb[16];
    fn_load16(s_u, mrel + 128 - 16 * kb0, xa);
#pragma unroll 1
    for (int kb = kb0; kb < kb0 + (NW == 8 ? 4 : 8); kb += 2) {
      fn_tap_block(s_u, s_h, mrel - 16 * kb, 16 * kb, acc, xa, xb, rel0, p.inv_fs);
      fn_tap_block(s_u, s_h, mrel - 16 * (kb + 1), 16 * (kb + 1), acc, xb, xa, rel0, p.inv_fs);
    }
  }
  __syncthreads();                                      // everyone is done reading x
  DDSP_STAMP();    // 4: FIR done
  // ---- 5. (sum the two tap halves,) transpose through LDS, coalesced stores: out[n] = z[n + start] ----
  if (NW == 8) {
    if (half == 1 && mrel < kFnTile) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
        *reinterpret_cast<float4*>(&s_u[fn_out_addr(mrel + 4 * c4)]) =
            make_float4(acc[4 * c4], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]);
    }
    __syncthreads();
    if (half == 0 && mrel < kFnTile) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const float4 u = *reinterpret_cast<const float4*>(&s_u[fn_out_addr(mrel + 4 * c4)]);
        acc[4 * c4] += u.x; acc[4 * c4 + 1] += u.y; acc[4 * c4 + 2] += u.z; acc[4 * c4 + 3] += u.w;
      }
    }
  }
  if (half == 0 && mrel < kFnTile) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
      *reinterpret_cast<float4*>(&s_u[fn_out_addr(mrel + 4 * c4)]) =
          make_float4(acc[4 * c4], acc[4 * c4 + 1], acc[4 * c4 + 2], acc[4 * c4 + 3]);
  }
  __syncthreads();
  float* __restrict__ o = out + (size_t)b * p.N;
  const long nbase = (long)z0 - p.start;                // out index of tile element 0
  const int head = (int)((4 - (nbase & 3)) & 3);        // tile elements before the first 16-byte boundary
  if ((p.N & 3) == 0) {
    if (tid < head) { const long n = nbase + tid; if (n >= 0 && n < p.N) o[n] = s_u[fn_out_addr(tid)]; }
    for (int e = head + 4 * tid; e < kFnTile; e += 4 * 64 * NW) {
      const long n = nbase + e;
      if (n >= 0 && n + 3 < p.N && e + 3 < kFnTile) {
        *reinterpret_cast<float4*>(o + n) = make_float4(s_u[fn_out_addr(e)], s_u[fn_out_addr(e + 1)],
                                                        s_u[fn_out_addr(e + 2)], s_u[fn_out_addr(e + 3)]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e + u < kFnTile && n + u >= 0 && n + u < p.N) o[n + u] = s_u[fn_out_addr(e + u)];
      }
    }
  } else {
    for (int e = tid; e < kFnTile; e += 64 * NW) {
      const long n = nbase + e;
      if (n >= 0 && n < p.N) o[n] = s_u[fn_out_addr(e)];
    }
  }
  DDSP_STAMP();    // 5: stored
#undef DDSP_STAMP
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a,
                                                  const float* __restrict__ b,
                                                  float* __restrict__ out, size_t n) {
  const size_t n4 = n / 4;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* o4 = reinterpret_cast<float4*>(out);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 x = a4[i], y = b4[i];
    o4[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void exp_sigmoid_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out, size_t n,
                                                          float log_exponent, float max_value,
                                                          float threshold) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = exp_sigmoid(in[i], log_exponent, max_value, threshold);
}

}  // namespace ddsp

// =====================================================================================
// C ABI
// =====================================================================================
using namespace ddsp;

static inline int check_launch() { return hipGetLastError() == hipSuccess ? DDSP_OK : DDSP_ERR_LAUNCH; }
static inline unsigned grid_for(size_t n, unsigned cap = 256 * 8) {
  size_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
constexpr size_t kMaxDynLds = 64 * 1024;

extern "C" const char* ddsp_version(void) { return "ddsp_amd 0.1.0 gfx950"; }

extern "C" int ddsp_fir_size(int M, int window_size) {
  if (M < 2) return DDSP_ERR_BAD_SHAPE;
  return ir_geom(M, window_size).L;
}

extern "C" int ddsp_filtered_noise_controls_f32(const float* magnitudes, float* ctl, int B, int F,
                                                int M, float initial_bias, unsigned flags,
                                                void* stream) {
  if (!magnitudes || !ctl) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || M <= 0) return DDSP_ERR_BAD_SHAPE;
  const size_t n = (size_t)B * F * M;
  ProfileScope prof(kNoiseControls, (hipStream_t)stream);
  hipLaunchKernelGGL(noise_controls_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     magnitudes, ctl, n, initial_bias,
                     (flags & DDSP_NOISE_SCALE_EXP_SIGMOID) ? 1 : 0);
  return check_launch();
}

static bool general_plain_env() {
  static const bool v = [] { const char* e = getenv("DDSP_EXP_NOISE_GENERAL"); return e && e[0] == 'p'; }();
  return v;
}

static int launch_ir(const float* mag, float* ctl_out, float* ir, int B, int F, int M,
                     int window_size, float bias, int scale, hipStream_t st) {
  const IrGeom g = ir_geom(M, window_size);
  if (M == 65 && g.padding == 0) {                 // canonical shape: lanes = frames
    const long rows65 = (long)B * F;
    ProfileScope prof(kNoiseIr, st);
    hipLaunchKernelGGL(noise_ir65_kernel, dim3((unsigned)((rows65 + 63) / 64)), dim3(256), 0, st,
                       mag, ctl_out, ir, rows65, bias, scale);
    return check_launch();
  }
  const long rows = (long)B * F;
  // any other filter: one matrix product with a constant matrix (filtered_noise_general.hip); DDSP_EXP_NOISE_GENERAL=plain
  // keeps rounds 1-3's plain kernels (the A/B of tools/bench_generic.py)
  if (!general_plain_env() && noise_ir_gemm_ok(M, window_size))
    return launch_noise_ir_gemm(mag, ctl_out, ir, rows, M, window_size, bias, scale, st);
  const size_t lds = (size_t)(g.L0 + M) * sizeof(float);
  if (lds > kMaxDynLds) return DDSP_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)(rows < 256 * 16 ? rows : 256 * 16);
  ProfileScope prof(kNoiseIr, st);
  hipLaunchKernelGGL(noise_ir_kernel, dim3(grid), dim3(256), lds, st, mag, ctl_out, ir, rows, M,
                     window_size, bias, scale);
  return check_launch();
}

extern "C" int ddsp_frequency_impulse_response_f32(const float* ctl_magnitudes, float* ir, int B,
                                                   int F, int M, int window_size, void* stream) {
  if (!ctl_magnitudes || !ir) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || M < 2) return DDSP_ERR_BAD_SHAPE;
  return launch_ir(ctl_magnitudes, nullptr, ir, B, F, M, window_size, 0.0f, 0, (hipStream_t)stream);
}

// the plain tiled kernel's tile: as many outputs as keep x + taps under the LDS budget; < 64: none does
static int fir_plain_tile(int L, int frame_size, size_t& lds) {
  int tile = 1024;
  for (; tile >= 64; tile /= 2) {
    const int nfr = (tile + L - 1) / frame_size + 2;
    lds = ((size_t)((tile + L - 1 + 3) & ~3) + (size_t)nfr * L) * sizeof(float);
    if (lds <= kMaxDynLds) break;
  }
  return tile;
}
// ... in which case launch_fir needs the noise in memory (FilteredNoise with generated noise: a scratch row per clip)
static bool fir_needs_noise_in_memory(int B, int F, int L, int N) {
  if (F <= 0 || N <= 0 || L <= 0) return false;
  const int frame_size = (N + F - 1) / F;
  if (L == 128 && frame_size == 64) return false;
  if (!general_plain_env() && tv_fir_mfma_ok(B, B, F, L, N)) return false;
  size_t lds = 0;
  return fir_plain_tile(L, frame_size, lds) < 64;
}

static int launch_fir(const float* x, const float* ir, float* out, int B, int Bir, int F, int L,
                      int N, int delay_compensation, uint64_t seed, uint64_t batch_offset,
                      int bits23, int taps_bounded, hipStream_t st, float* x_scratch = nullptr) {
  if (B > 65535) return DDSP_ERR_UNSUPPORTED;
  FirArgs p;
  p.x = x; p.ir = ir; p.out = out;
  p.N = N; p.F = F; p.L = L;
  p.frame_size = (N + F - 1) / F;                               // core.py:1446
  if ((N + p.frame_size - 1) / p.frame_size != F) return DDSP_ERR_BAD_SHAPE;   // :1451-1457
  p.start = delay_compensation < 0 ? (L - 1) / 2 - 1 : delay_compensation;      // :1375-1376
  if (p.start < 0) p.start = 0;   // L <= 2 with automatic compensation: python slice start -1
  p.ir_batch_stride = (Bir == 1) ? 0 : (size_t)F * L;
  p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
  p.batch_offset = batch_offset;
  p.bits23 = bits23;
  if (L == 128 && p.frame_size == 64 && Bir == B && (((uintptr_t)ir) & 15) == 0 &&
      (x == nullptr || (((uintptr_t)x) & 15) == 0)) {
    Fir128Args q;
    q.N = N; q.F = F; q.start = p.start; q.k0 = p.k0; q.k1 = p.k1; q.batch_offset = batch_offset; q.bits23 = bits23;
    const dim3 grid((unsigned)((N + p.start + kFirTile - 1) / kFirTile), (unsigned)B);
    ProfileScope prof(kTvFir, st);
    if (x) hipLaunchKernelGGL((tv_fir128_kernel<false>), grid, dim3(64 * kFirWaves), 0, st, x, ir, out, q);
    else hipLaunchKernelGGL((tv_fir128_kernel<true>), grid, dim3(64 * kFirWaves), 0, st, x, ir, out, q);
    return check_launch();
  }
  // any other tap count and frame size: Toeplitz products on the matrix cores (filtered_noise_general.hip)
  if (!general_plain_env() && tv_fir_mfma_ok(B, Bir, F, L, N))
    return launch_tv_fir_mfma(x, ir, out, B, Bir, F, L, N, p.start, seed, batch_offset, bits23, taps_bounded, st);
  // tile: as many outputs as keep x + taps under the LDS budget
  size_t lds = 0;
  const int tile = fir_plain_tile(L, p.frame_size, lds);
  if (tile < 64) {
    // filters that reach across so many frames that not even 64 outputs' taps fit the LDS (510 taps on frames of 5 samples: 116
    // tap rows per tile) - found by tools/fuzz_parity.py, the reference takes any shape: the plain sum of csrc/general.hip, which
    // reads its taps from memory.  It wants the noise in memory: the caller's, or generated into `x_scratch` first.
    if (!x) {
      if (!x_scratch) return DDSP_ERR_UNSUPPORTED;
      const int rc = ddsp_uniform_noise_ex_f32(x_scratch, B, N, seed, batch_offset, bits23 ? 23 : 11, st);
      if (rc != DDSP_OK) return rc;
      x = x_scratch;
    }
    return ddsp_fft_convolve_f32(x, ir, out, B, Bir, F, L, N, N, p.start, st);
  }
  p.tile = tile;
  const dim3 grid((unsigned)((N + tile - 1) / tile), (unsigned)B), block(256);
  ProfileScope prof(kTvFir, st);
  if (x) hipLaunchKernelGGL((tv_fir_kernel<false>), grid, block, lds, st, p);
  else hipLaunchKernelGGL((tv_fir_kernel<true>), grid, block, lds, st, p);
  return check_launch();
}

extern "C" int ddsp_fft_convolve_same_f32(const float* audio, const float* impulse_response,
                                          float* out, int B, int Bir, int F, int L, int N,
                                          int delay_compensation, void* stream) {
  if (!audio || !impulse_response || !out) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || L <= 0 || N <= 0 || (Bir != B && Bir != 1)) return DDSP_ERR_BAD_SHAPE;
  return launch_fir(audio, impulse_response, out, B, Bir, F, L, N, delay_compensation, 0, 0, 0, 0,
                    (hipStream_t)stream);
}

extern "C" size_t ddsp_filtered_noise_workspace_bytes(int B, int F, int M, int N, int window_size) {
  if (B <= 0 || F <= 0 || M < 2) return 0;
  const int L = ir_geom(M, window_size).L;
  size_t bytes = (size_t)B * F * (size_t)L * sizeof(float);
  if (N > 0 && fir_needs_noise_in_memory(B, F, L, N))                   // (launch_fir: generated noise for the plain sum)
    bytes = ((bytes + 15) & ~(size_t)15) + (size_t)B * N * sizeof(float);
  return bytes;
}

extern "C" int ddsp_filtered_noise_f32(const float* magnitudes, const float* noise, float* audio,
                                       float* ctl_magnitudes, void* workspace,
                                       size_t workspace_bytes, int B, int F, int M, int N,
                                       int window_size, float initial_bias, unsigned flags,
                                       uint64_t seed, uint64_t batch_offset, void* stream) {
  if (!magnitudes || !audio || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || M < 2 || N <= 0) return DDSP_ERR_BAD_SHAPE;
  if (workspace_bytes < ddsp_filtered_noise_workspace_bytes(B, F, M, N, window_size) ||
      ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int scale = (flags & DDSP_NOISE_SCALE_EXP_SIGMOID) ? 1 : 0;
  // bit 30 (the in-kernel timeline of tools/exp_noise_fir.py: the controls pointer then carries a stamp buffer) only
  // under DDSP_NOISE_DEBUG_TIMELINE=1; any other unknown bit is an error, not a silent reinterpretation
  static const bool dbg_allowed = getenv("DDSP_NOISE_DEBUG_TIMELINE") != nullptr;
  const unsigned known = DDSP_NOISE_SCALE_EXP_SIGMOID | DDSP_NOISE_FIR_VECTOR_ALU | DDSP_NOISE_BITS_23 | (dbg_allowed ? 0x40000000u : 0u);
  if (flags & ~known) return DDSP_ERR_UNSUPPORTED;
  const int bits23 = (flags & DDSP_NOISE_BITS_23) ? 1 : 0;             // (generated noise only: supplied noise is what it is)
  {
    const IrGeom g = ir_geom(M, window_size);
    const int frame_size = (N + F - 1) / F;
    // default for the canonical filter: IR design AND the FIR on the matrix cores (filtered_noise_mfma.hip);
    // DDSP_NOISE_FIR_VECTOR_ALU (or DDSP_EXP_NOISE_FIR=vector) keeps the FIR on the vector ALUs (noise_fused65_kernel)
    static const bool fir_vector_env = [] { const char* e = getenv("DDSP_EXP_NOISE_FIR"); return e && e[0] == 'v'; }();
    if (!(flags & DDSP_NOISE_FIR_VECTOR_ALU) && !fir_vector_env && B <= 65535 &&
        noise_mfma65_ok(F, M, N, g.padding, noise, scale)) {
      long long* dbg = (flags & 0x40000000u) ? reinterpret_cast<long long*>(ctl_magnitudes) : nullptr;   // debug timeline
      return launch_noise_mfma65(magnitudes, noise, audio, dbg ? nullptr : ctl_magnitudes, B, F, N, (g.L - 1) / 2 - 1,
                                 initial_bias, scale, seed, batch_offset, dbg, bits23, (float*)workspace, st);
    }
    // (the vector-ALU kernel designs its taps on the matrix cores from magnitudes it splits as they come: squashed ones only)
    if (scale && M == 65 && g.padding == 0 && frame_size >= 64 && (frame_size % 16) == 0 && frame_size <= 4096 &&
        (N + frame_size - 1) / frame_size == F && B <= 65535 &&
        (noise == nullptr || (((uintptr_t)noise) & 15) == 0)) {
      FusedNoiseArgs q;
      q.N = N; q.F = F; q.start = (g.L - 1) / 2 - 1; q.bias = initial_bias;
      q.fs = frame_size; q.inv_fs = 1.0f / (float)frame_size;
      q.scale = scale | ((flags & 0x40000000u) ? 0x40000000 : 0);      // bit 30: debug timeline
      q.k0 = (uint32_t)seed; q.k1 = (uint32_t)(seed >> 32); q.batch_offset = batch_offset; q.bits23 = bits23;
      const dim3 grid((unsigned)((N + q.start + kFnTile - 1) / kFnTile), (unsigned)B);
      hipEvent_t ev0, ev1;
      profile_kernel_events(kNoiseFused, &ev0, &ev1);
      // 8-wavefront blocks (FIR tap range split in two) hide more latency once there is more than one
      // round of blocks (3 resident per CU); a single partial round next to the harmonic kernel on the
      // other stream does better with 4-wavefront blocks (measured: profiles/README.md)
      static const int nw_env = [] { const char* e = getenv("DDSP_EXP_NOISE_WAVES"); return e ? atoi(e) : 0; }();
      const int nw = nw_env ? nw_env : ((size_t)grid.x * grid.y > 768 ? 8 : 4);
      if (nw == 8) {
        if (noise) hipExtLaunchKernelGGL((noise_fused65_kernel<false, 8>), grid, dim3(512), 0, st, ev0, ev1, 0,
                                         magnitudes, noise, ctl_magnitudes, audio, q);
        else hipExtLaunchKernelGGL((noise_fused65_kernel<true, 8>), grid, dim3(512), 0, st, ev0, ev1, 0,
                                   magnitudes, noise, ctl_magnitudes, audio, q);
      } else {
        if (noise) hipExtLaunchKernelGGL((noise_fused65_kernel<false, 4>), grid, dim3(256), 0, st, ev0, ev1, 0,
                                         magnitudes, noise, ctl_magnitudes, audio, q);
        else hipExtLaunchKernelGGL((noise_fused65_kernel<true, 4>), grid, dim3(256), 0, st, ev0, ev1, 0,
                                   magnitudes, noise, ctl_magnitudes, audio, q);
      }
      return check_launch();
    }
  }
  // any other filter of up to 128 bands and 256 taps, any frame size: one launch, the taps designed tile by tile in LDS
  // (filtered_noise_general.hip); DDSP_EXP_NOISE_GENERAL=two keeps the design and the FIR apart (the larger filters' path)
  static const bool two_env = [] { const char* e = getenv("DDSP_EXP_NOISE_GENERAL"); return e && e[0] == 't'; }();
  if (!general_plain_env() && !two_env && filtered_noise_general_fused_ok(B, F, M, N, window_size))
    return launch_filtered_noise_general_fused(magnitudes, noise, audio, ctl_magnitudes, B, F, M, N, window_size, initial_bias,
                                               scale, seed, batch_offset, bits23, st);
  float* ir = (float*)workspace;
  int rc = launch_ir(magnitudes, ctl_magnitudes, ir, B, F, M, window_size, initial_bias, scale, st);
  if (rc != DDSP_OK) return rc;
  const int L = ir_geom(M, window_size).L;
  float* x_scratch = nullptr;
  if (!noise && fir_needs_noise_in_memory(B, F, L, N))
    x_scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((((size_t)B * F * L * sizeof(float)) + 15) & ~(size_t)15));
  return launch_fir(noise, ir, audio, B, B, F, L, N, -1, seed,
                    batch_offset, bits23, scale, st, x_scratch);  // (scale: the taps were designed from squashed magnitudes)
}

// =====================================================================================
// Backward pass of FilteredNoise.__call__ for the fused shape (M = 65, full window, L = 128):
// dL/d(magnitudes) from dL/d(audio).  The output is linear in every frame's taps,
//     dL/dh_f[t] = sum_{i in frame f} x[i] gz[i + t],   gz[m] = dL/d audio[m - start],
// the taps are linear in the scaled magnitudes (the transpose of the cosine transform of the
// forward IR design), and only exp_sigmoid is not linear.  The noise x is regenerated by the same
// Philox counter/key as in the forward call (or read from the tensor the caller supplied).
//   noise_bwd_taps_kernel  one wavefront per frame, lanes = taps t and t+64; the frame's samples are
//                          held one per lane and broadcast with v_readlane, the gradient window
//                          slides through an LDS tile.
//   noise_bwd_mags_kernel  one wavefront per frame: folds the 128 tap gradients onto the 33 even /
//                          32 odd partial sums, applies the transposed cosine table and exp_sigmoid'.
// =====================================================================================
namespace ddsp {

struct NoiseBwdArgs {
  int N, F, fs, start, tile_frames, scale;
  int L, Lpad, M, window_size;   // taps per frame, rounded up to 128; magnitudes per frame; Hann size argument
  float bias;
  uint32_t k0, k1;
  uint64_t batch_offset;
  int bits23;
};

__global__ __launch_bounds__(256) void noise_bwd_taps_kernel(const float* __restrict__ x /*[B,N] or null*/,
                                                             const float* __restrict__ grad_audio,
                                                             float* __restrict__ dh /*[B,F,L]*/,
                                                             NoiseBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float s_g[];        // gz[f0*fs .. (f0+tile)*fs + Lpad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * p.tile_frames;
  const int nfr = min(p.tile_frames, p.F - f0);
  const int span = nfr * p.fs + p.Lpad;
  const float* __restrict__ g = grad_audio + (size_t)b * p.N;
  for (int e = tid; e < span + 64; e += 256) {             // + 64 zeros: the last 64-sample chunk of a frame
    const int n = f0 * p.fs + e - p.start;                 // whose size is not a multiple of 64 reads past the span
    s_g[e] = (e < span && n >= 0 && n < p.N) ? g[n] : 0.0f;   // gz[m] = g[m - start]
  }
  __syncthreads();
  for (int q = wave; q < nfr; q += 4) {
    const int f = f0 + q;
    for (int kc = 0; kc < p.L; kc += 128) {                // taps kc + lane and kc + 64 + lane (L = 128: one trip)
      // two samples at a time on packed FMAs: (x_l, x_l+1) from two readlanes into a scalar pair, (gz[i_l + t], gz[i_l+1 + t])
      // as the read pair brings them - the even and the odd samples' sums of a tap in the two halves of one register pair,
      // added at the end (3 instructions per sample where 5 went)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 acc_lo = {0.0f, 0.0f}, acc_hi = {0.0f, 0.0f};
      for (int c = 0; c < p.fs; c += 64) {                 // 64 samples of the frame per chunk
        const int i = f * p.fs + c + lane;                 // this lane's sample
        float xv = 0.0f;
        if (c + lane < p.fs && i < p.N)
          xv = x ? x[(size_t)b * p.N + i] : philox_noise((uint32_t)i, p.batch_offset + b, p.k0, p.k1, p.bits23 != 0);
        const float* __restrict__ win = s_g + q * p.fs + c + kc + lane;    // + l: gz[i_l + t], t = kc + lane
#pragma unroll
        for (int l = 0; l < 64; l += 2) {
          const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), l));
          const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), l + 1));
          const f32x2 xx = {x0, x1};
          acc_lo = __builtin_elementwise_fma(xx, (f32x2){win[l], win[l + 1]}, acc_lo);
          acc_hi = __builtin_elementwise_fma(xx, (f32x2){win[l + 64], win[l + 65]}, acc_hi);
        }
      }
      const float acc0 = acc_lo[0] + acc_lo[1], acc1 = acc_hi[0] + acc_hi[1];
      float* __restrict__ o = dh + ((size_t)b * p.F + f) * p.L + kc;
      if (kc + lane < p.L) o[lane] = acc0;
      if (kc + 64 + lane < p.L) o[lane + 64] = acc1;
    }
  }
}

// Any band count / window: dL/d mag[m] = (c_m / L0) sum_kappa dh[kappa] w(kappa) cos(2 pi m n(kappa) / L0),
// the transpose of noise_ir_kernel (c_m = 1 for the DC and Nyquist bins, 2 otherwise), times exp_sigmoid'.
__global__ __launch_bounds__(256) void noise_bwd_mags_generic_kernel(const float* __restrict__ mag,
                                                                     const float* __restrict__ dh,
                                                                     float* __restrict__ grad_mag,
                                                                     long rows, NoiseBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const IrGeom g = ir_geom(p.M, p.window_size);
  float* s_cos = smem;                 // [L0]
  float* s_dw = smem + g.L0;           // [L]  dh * window
  int* s_n = reinterpret_cast<int*>(smem + g.L0 + g.L);    // [L] zero-phase sample index of each tap
  for (int i = threadIdx.x; i < g.L0; i += 256) s_cos[i] = cospif(2.0f * (float)i / (float)g.L0);
  for (int kappa = threadIdx.x; kappa < g.L; kappa += 256) {
    int n, widx;
    ir_tap_map(g, kappa, &n, &widx);
    s_n[kappa] = n;
  }
  const float inv_L0 = 1.0f / (float)g.L0;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
    for (int kappa = threadIdx.x; kappa < g.L; kappa += 256) {
      int n, widx;
      ir_tap_map(g, kappa, &n, &widx);
      // (tf.signal.hann_window's denominator: noise_ir_geom.h - the forward kernel's window, above)
      const float w = (widx < 0) ? 0.0f : (g.ws == 1 ? 1.0f : 0.5f - 0.5f * cospif(2.0f * (float)widx / (float)hann_denominator(g.ws)));
      s_dw[kappa] = w * dh[(size_t)row * g.L + kappa];
    }
    __syncthreads();
    for (int m = threadIdx.x; m < p.M; m += 256) {
      float acc = 0.0f;
      for (int kappa = 0; kappa < g.L; ++kappa) {
        const int idx = (int)(((long)m * s_n[kappa]) % g.L0);
        acc = fmaf(s_dw[kappa], s_cos[idx], acc);
      }
      acc *= ((m == 0 || m == p.M - 1) ? 1.0f : 2.0f) * inv_L0;
      const size_t at = (size_t)row * p.M + m;
      if (p.scale) {
        const float xr = mag[at] + p.bias;
        const float y = exp_sigmoid(xr, 2.302585092994046f, 2.0f, 1e-7f);
        acc *= 2.302585092994046f * (y - 1e-7f) * (1.0f - 1.0f / (1.0f + __expf(-xr)));
      }
      grad_mag[at] = acc;
    }
  }
}

constexpr int kBwdMagRows = 16;        // rows (frames) per block: the cosine table is staged in LDS once per block

__global__ __launch_bounds__(256) void noise_bwd_mags_kernel(const float* __restrict__ mag /*[B,F,65]*/,
                                                             const float* __restrict__ dh /*[B*F,128]*/,
                                                             float* __restrict__ grad_mag, long rows,
                                                             NoiseBwdArgs p) {
  __shared__ float s_tab[33 * kIrRowStride];                 // kIr65.c: per-lane column reads are LDS reads, not 33 scattered loads
  __shared__ float s_eo[4][80];                              // per wavefront: d_e[0..32] at 0, d_o[0..32] at 40
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 33 * kIrRowStride; e += 256) s_tab[e] = kIr65.c[e];
  __syncthreads();
  const long row_end = min(rows, ((long)blockIdx.x + 1) * kBwdMagRows);
  for (long row = (long)blockIdx.x * kBwdMagRows + wave; row < row_end; row += 4) {
    const float* __restrict__ d = dh + (size_t)row * 128;
    if (lane <= 32) {
      const int n = lane;
      // forward: g0[n] = win[n] (e+o) -> taps 64+n and 64-n;  g1[n] = win[64-n] (e-o) -> taps 128-n and n
      // (1 <= n < 32);  tap 0 is the constant 0
      const float dg0 = d[64 + n] + (n >= 1 ? d[64 - n] : 0.0f);
      const float dg1 = (n >= 1 && n < 32) ? d[128 - n] + d[n] : 0.0f;
      const float w0 = kIr65.win[n] * dg0;
      const float w1 = (n >= 1 && n < 32) ? kIr65.win[64 - n] * dg1 : 0.0f;
      s_eo[wave][n] = w0 + w1;
      s_eo[wave][40 + n] = w0 - w1;
    }
    __builtin_amdgcn_s_waitcnt(0);                           // this wavefront's LDS writes (in order per wave)
    __builtin_amdgcn_wave_barrier();
    // bins 0 .. 63: one per lane, 33 terms each; bin 64 (even, i = 32): its 33 terms one per lane and a wavefront sum -
    // as a second trip through the loop above it was 33 serial steps on lane 0 with 63 lanes waiting, half the kernel
    float acc64 = (lane <= 32) ? s_tab[32 + lane * kIrRowStride] * s_eo[wave][lane] : 0.0f;
    acc64 = wave_sum(acc64);
    {
      const int m = lane;
      const int i = m >> 1, odd = m & 1;
      const float* __restrict__ tab = s_tab + (odd ? 40 + i : i);
      const float* __restrict__ src = &s_eo[wave][odd ? 40 : 0];
      float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
      for (int n = 0; n < 32; n += 2) {
        acc0 = fmaf(tab[n * kIrRowStride], src[n], acc0);
        acc1 = fmaf(tab[(n + 1) * kIrRowStride], src[n + 1], acc1);
      }
      float acc = fmaf(tab[32 * kIrRowStride], src[32], acc0 + acc1);
      const size_t at = (size_t)row * 65 + m;
      if (p.scale) {
        const float xr = mag[at] + p.bias;
        const float y = exp_sigmoid_fast(xr, 2.302585092994046f, 2.0f, 1e-7f);
        acc *= 2.302585092994046f * (y - 1e-7f) * (1.0f - __builtin_amdgcn_rcpf(1.0f + __expf(-xr)));
      }
      grad_mag[at] = acc;
      if (lane == 0) {
        const size_t at64 = (size_t)row * 65 + 64;
        float a64 = acc64;
        if (p.scale) {
          const float xr = mag[at64] + p.bias;
          const float y = exp_sigmoid_fast(xr, 2.302585092994046f, 2.0f, 1e-7f);
          a64 *= 2.302585092994046f * (y - 1e-7f) * (1.0f - __builtin_amdgcn_rcpf(1.0f + __expf(-xr)));
        }
        grad_mag[at64] = a64;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);                           // s_eo is rewritten by the next row of this wavefront
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace ddsp

extern "C" size_t ddsp_filtered_noise_backward_workspace_bytes(int B, int F, int M, int N) {
  (void)N;
  if (B <= 0 || F <= 0 || M < 2) return 0;
  return (size_t)B * F * (size_t)(2 * (M - 1)) * sizeof(float);     // tap gradients, L <= 2 (M - 1)
}

extern "C" int ddsp_filtered_noise_backward_f32(const float* magnitudes, const float* noise,
                                                const float* grad_audio, float* grad_magnitudes,
                                                void* workspace, size_t workspace_bytes, int B,
                                                int F, int M, int N, int window_size,
                                                float initial_bias, unsigned flags, uint64_t seed,
                                                uint64_t batch_offset, void* stream) {
  if (!magnitudes || !grad_audio || !grad_magnitudes || !workspace) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || M < 2 || N <= 0) return DDSP_ERR_BAD_SHAPE;
  const IrGeom g = ir_geom(M, window_size);
  const int fs = (N + F - 1) / F;
  const int start = (g.L - 1) / 2 - 1;
  if ((N + fs - 1) / fs != F || B > 65535 || start < 0 || fs > 8192 || g.L0 > 8192) return DDSP_ERR_UNSUPPORTED;
  if (workspace_bytes < ddsp_filtered_noise_backward_workspace_bytes(B, F, M, N) ||
      ((uintptr_t)workspace & 15))
    return DDSP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (flags & ~(DDSP_NOISE_SCALE_EXP_SIGMOID | DDSP_NOISE_FIR_VECTOR_ALU | DDSP_NOISE_BITS_23)) return DDSP_ERR_UNSUPPORTED;
  const int bits23 = (flags & DDSP_NOISE_BITS_23) ? 1 : 0;             // as in the forward call whose noise is regenerated
  // the canonical filter on frames of 64 c samples: one launch on the matrix cores (filtered_noise_general.hip)
  if (noise_bwd_mfma_ok(B, F, M, N, window_size))
    return launch_noise_bwd_mfma(magnitudes, noise, grad_audio, grad_magnitudes, B, F, M, N, window_size, initial_bias,
                                 (flags & DDSP_NOISE_SCALE_EXP_SIGMOID) ? 1 : 0, seed, batch_offset, bits23, st);
  NoiseBwdArgs p;
  p.N = N; p.F = F; p.fs = fs; p.start = start;
  p.L = g.L; p.Lpad = (g.L + 127) & ~127; p.M = M; p.window_size = window_size;
  p.tile_frames = 8192 / fs;                                // <= 8192 + Lpad gradient samples staged per block
  if (p.tile_frames > 32) p.tile_frames = 32;
  if (p.tile_frames < 1) p.tile_frames = 1;
  p.scale = (flags & DDSP_NOISE_SCALE_EXP_SIGMOID) ? 1 : 0;
  p.bias = initial_bias;
  p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32); p.batch_offset = batch_offset; p.bits23 = bits23;
  float* dh = (float*)workspace;
  {
    ProfileScope prof(kNoiseBwdTaps, st);
    const dim3 grid((unsigned)((F + p.tile_frames - 1) / p.tile_frames), (unsigned)B);
    const size_t lds = ((size_t)p.tile_frames * fs + p.Lpad + 64) * sizeof(float);
    static const bool attr_set = [] {
      (void)hipFuncSetAttribute((const void*)noise_bwd_taps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
      return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL(noise_bwd_taps_kernel, grid, dim3(256), lds, st, noise, grad_audio, dh, p);
  }
  {
    ProfileScope prof(kNoiseBwdMags, st);
    const long rows = (long)B * F;
    if (M == 65 && g.padding == 0) {                        // the fused forward kernel's shape: even / odd folding
      hipLaunchKernelGGL(noise_bwd_mags_kernel, dim3((unsigned)((rows + kBwdMagRows - 1) / kBwdMagRows)), dim3(256), 0, st,
                         magnitudes, (const float*)dh, grad_magnitudes, rows, p);
    } else {
      const size_t lds = ((size_t)g.L0 + 2 * (size_t)g.L) * sizeof(float);
      const unsigned blocks = (unsigned)(rows < 8192 ? rows : 8192);
      hipLaunchKernelGGL(noise_bwd_mags_generic_kernel, dim3(blocks), dim3(256), lds, st, magnitudes,
                         (const float*)dh, grad_magnitudes, rows, p);
    }
  }
  return check_launch();
}

extern "C" int ddsp_uniform_noise_ex_f32(float* out, int B, int N, uint64_t seed,
                                         uint64_t batch_offset, int noise_bits, void* stream) {
  if (!out) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || N <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (noise_bits != 11 && noise_bits != 23) return DDSP_ERR_UNSUPPORTED;
  const dim3 grid(grid_for((size_t)(N + 7) / 8, 64), (unsigned)B);
  ProfileScope prof(kUniformNoise, (hipStream_t)stream);
  hipLaunchKernelGGL(uniform_noise_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, N,
                     (uint32_t)seed, (uint32_t)(seed >> 32), batch_offset, noise_bits == 23 ? 1 : 0);
  return check_launch();
}

extern "C" int ddsp_uniform_noise_f32(float* out, int B, int N, uint64_t seed,
                                      uint64_t batch_offset, void* stream) {
  return ddsp_uniform_noise_ex_f32(out, B, N, seed, batch_offset, 11, stream);
}

extern "C" int ddsp_add_f32(const float* a, const float* b, float* out, size_t n, void* stream) {
  if (!a || !b || !out) return DDSP_ERR_NULL_POINTER;
  if (n == 0) return DDSP_OK;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return DDSP_ERR_UNSUPPORTED;
  ProfileScope prof(kAdd, (hipStream_t)stream);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n);
  return check_launch();
}

extern "C" int ddsp_exp_sigmoid_f32(const float* in, float* out, size_t n, float exponent,
                                    float max_value, float threshold, void* stream) {
  if (!in || !out) return DDSP_ERR_NULL_POINTER;
  if (n == 0) return DDSP_OK;
  ProfileScope prof(kExpSigmoid, (hipStream_t)stream);
  hipLaunchKernelGGL(exp_sigmoid_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in,
                     out, n, logf(exponent), max_value, threshold);
  return check_launch();
}

// =====================================================================================
// Stand-alone frame-rate -> audio-rate resampling (core.resample 'linear' / 'window',
// ddsp/core.py:573-714) for callers that want the envelopes themselves.  The synth kernels
// never materialise these [B,N,C] tensors; this is an HBM-write-bound elementwise kernel.
// =====================================================================================
namespace ddsp {
// out[b,t,c] = x[b,lo,c]*(1-w) + x[b,hi,c]*w with
//   'linear' (legacy bilinear, align_corners=False): pos = t*fl32(F/N), lo=floor, hi=min(ceil,F-1),
//             w = pos-lo, evaluated as top + (bottom-top)*w exactly like TF;
//   'window' (upsample_with_windows, add_endpoint=True): j=t/hop, r=t%hop, hi=min(j+1,F-1),
//             w = Hann(2*hop)[r] = 0.5-0.5*cos(pi*r/hop), out = x[j]*(1-w) + x[hi]*w.
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x,
                                                       float* __restrict__ out, int F, int N, int C,
                                                       int window, float scale, int hop) {
  const int b = blockIdx.y;
  const size_t total = (size_t)N * C;
  const float* __restrict__ xb = x + (size_t)b * F * C;
  float* __restrict__ ob = out + (size_t)b * total;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i / C), c = (int)(i - (size_t)t * C);
    if (window) {
      const int j = t / hop, r = t - j * hop;
      const int hi = min(j + 1, F - 1);
      const float w = 0.5f - 0.5f * cospif((float)r / (float)hop);
      ob[i] = xb[(size_t)j * C + c] * (1.0f - w) + xb[(size_t)hi * C + c] * w;
    } else {
      // every step individually rounded (no FMA contraction of t*scale - lo: found on the MI355X in round 1)
      const float pos = rn_mul((float)t, scale);
      const float lo = floorf(pos);
      const int lo_i = (int)lo, hi_i = min((int)ceilf(pos), F - 1);
      const float top = xb[(size_t)lo_i * C + c], bottom = xb[(size_t)hi_i * C + c];
      ob[i] = rn_add(top, rn_mul(rn_sub(bottom, top), rn_sub(pos, lo)));
    }
  }
}
}  // namespace ddsp

extern "C" int ddsp_resample_f32(const float* x, float* out, int B, int F, int N, int C, int window,
                                 void* stream) {
  if (!x || !out) return DDSP_ERR_NULL_POINTER;
  if (B <= 0 || F <= 0 || N <= 0 || C <= 0 || B > 65535) return DDSP_ERR_BAD_SHAPE;
  if (window && (N % F != 0)) return DDSP_ERR_BAD_SHAPE;
  const float scale = (float)F / (float)N;            // fp32, as TF computes it
  const size_t total = (size_t)N * C;
  const dim3 grid(grid_for(total, 2048), (unsigned)B);
  hipLaunchKernelGGL(ddsp::resample_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, F, N, C,
                     window, scale, window ? N / F : 1);
  return check_launch();
}
